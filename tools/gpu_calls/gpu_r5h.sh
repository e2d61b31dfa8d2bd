#!/bin/bash
# round 5 call h: the whole GPU suite (per-test timeout), the FP64-mix ceiling micro-benchmark, per-kernel times of one
# ssr_fft_lowpass_multi call, bench cfg3 on the product default engine
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -x > gpurun_out/r5h_tests.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r5h_tests.log
timeout 120 tools/_build/fp64_mix > gpurun_out/r5h_fp64_mix.log 2>&1; echo "ubench rc=$?"; cat gpurun_out/r5h_fp64_mix.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r5h_trace
PARITY=0 UTT=1024 CUTS_TIMED=683 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5h_trace -o t -- python $R/tools/exp_tlconv.py > $R/gpurun_out/r5h_trace.log 2>&1
echo "trace rc=$?"
cd $R
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/r5h_trace/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_tl_" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"][:24], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
print("last 34 k_tl_* launches (two multi calls: pad, fwd, 7 x (inv, fold)):")
for t, k, ms in rows[-34:]:
    print("   %-26s %8.3f ms" % (k, ms))
PY
find gpurun_out/r5h_trace -name "*.csv" -size +2M -delete
timeout 400 python bench.py --config cfg3 --steps 3 --warmup 1 > gpurun_out/r5h_bench_cfg3.log 2>&1
echo "bench rc=$?"; tail -2 gpurun_out/r5h_bench_cfg3.log | cut -c1-2500
