#!/bin/bash
# round 4 call u: SQ counters + HBM counters of k_resample_chain (4096 utterances)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/exp_chain.py | tee gpurun_out/r4u_time.log
FUSED=0 python tools/exp_chain.py | tee -a gpurun_out/r4u_time.log
REPS=2 bash tools/pmc_cmd.sh chain k_resample_chain -- python tools/exp_chain.py 2>&1 | tail -30 | tee gpurun_out/r4u_sq.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  REPS=2 timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/r4u_$P -o p -- python $R/tools/exp_chain.py > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/r4u_$P/*counter_collection.csv")[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_resample_chain" in r["Kernel_Name"] and r["Counter_Name"]=="$P"]
v.sort(); print("$P median KiB per launch (4096 utt):", v[len(v)//2], "n=",len(v))
PY
done | tee $R/gpurun_out/r4u_hbm.log
find $R/gpurun_out -name "*counter_collection.csv" -delete; find $R/gpurun_out -name "*kernel_trace.csv" -delete
