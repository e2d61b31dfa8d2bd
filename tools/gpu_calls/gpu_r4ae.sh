#!/bin/bash
# round 4 call ae: matrix-pipe counters of k_tl_gemm (tools/exp_tlconv.py, 256 utterances)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
R=$PWD; OUT=$R/gpurun_out/pmc_gemm; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o c -- python tools/exp_tlconv.py) > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_gemm/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_tl_gemm" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        v.sort(); print("   %-32s n=%3d max=%.4e median=%.4e" % (c, len(v), v[-1], v[len(v)//2]))
PY
find gpurun_out/pmc_gemm -name "*.csv" -size +1M -delete
