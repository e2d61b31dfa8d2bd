#!/bin/bash
# round 5 call x: SQ counters of k_resample_sinc (44.1 -> 48 kHz, 128 files): occupancy, VALU / LDS / VMEM activity, waits
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
R=$PWD; OUT=$R/gpurun_out/r5x_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  (cd $R && FILES=128 PAIRS=44100:48000 timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o c -- python tools/exp_sinc.py) > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python - <<'PY' | tee gpurun_out/r5x_sinc_counters.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r5x_pmc/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_resample_sinc" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for key in ("LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size", "Grid_Size", "Scratch_Size"):
                if key in r: agg[r["Kernel_Name"][:40]]["~" + key].append(float(r[key]))
for f in glob.glob("gpurun_out/r5x_pmc/p*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_resample_sinc" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, d in agg.items():
    v = sorted(dur[k]); print(k, "duration median %.3f ms (n=%d)" % (v[len(v)//2], len(v)))
    for c, v in sorted(d.items()):
        v.sort(); print("   %-32s n=%3d median=%.4e" % (c, len(v), v[len(v)//2]))
PY
find gpurun_out/r5x_pmc -name "*.csv" -size +1M -delete
