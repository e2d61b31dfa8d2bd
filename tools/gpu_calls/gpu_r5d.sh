#!/bin/bash
# round 5 call d: mid-chunk barrier pipeline (three stages, transfers between the matrix instructions) in both products; SQ counters
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
R=$PWD
SSR_DEV_LIB=tools/_build/libssrhip_knobs.so VARIANTS="128:3,64:3" UTT=1024 timeout 900 python tools/exp_tlconv.py > gpurun_out/r5d_tlconv.log 2>&1
echo "rc=$?"; grep -v "^per-item\|^uniform\|^stft\|^multi cut" gpurun_out/r5d_tlconv.log | tail -30
OUT=$R/gpurun_out/r5d_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd $R && PARITY=0 UTT=256 CUTS_TIMED=683 MULTI=0 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o c -- python tools/exp_tlconv.py) > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r5d_pmc/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_tl_" in r["Kernel_Name"] and ("inv" in r["Kernel_Name"] or "fwd" in r["Kernel_Name"]):
            agg[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        v.sort(); print("   %-32s n=%3d max=%.4e median=%.4e" % (c, len(v), v[-1], v[len(v)//2]))
PY
find gpurun_out/r5d_pmc -name "*.csv" -size +1M -delete
