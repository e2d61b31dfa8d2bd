#!/bin/bash
# round 6 call s: kernel trace of an evaluate() pass with the 36 IIR keys after the packed IIR wavefront and ssr_pair_metrics_multi_est64
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r6s_prof
PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6s_prof -o s -- python $R/tools/exp_e2e.py > $R/gpurun_out/r6s_e2e.log 2>&1
cd $R; grep "evaluate()" gpurun_out/r6s_e2e.log | cut -c1-200
T=$(find gpurun_out/r6s_prof -name "*kernel_trace.csv" | head -1)
WINDOW_MS=${WIN:-400} python tools/trace_gaps.py $T | tee gpurun_out/r6s_gaps_last_pass.txt
python - "$T" <<'PYEOF' | tee gpurun_out/r6s_last_pass_kernels.txt
import csv, sys, collections, os
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(); t_end = rows[-1][1]; rows = [r for r in rows if r[0] >= t_end - float(os.environ.get("WIN", 400)) * 1e6]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, k in rows: agg[k][0] += 1; agg[k][1] += (e - s) / 1e6
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]: print("%-62s %4d launches %9.2f ms" % (k, n, ms))
PYEOF
find gpurun_out/r6s_prof -name "*kernel_trace.csv" -delete
