#!/bin/bash
# round 4 call aj: timeline of one ssr_pair_metrics_multi call (7 keys x 1024) - gaps between its kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4aj -o t -- python $R/tools/exp_multi.py > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4aj/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: find the last k_rows_from_len
idx = [i for i, r in enumerate(rows) if "k_rows_from_len" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s start %8.1f us  dur %8.1f us  gap %6.1f us" % (r["Kernel_Name"][:44], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
print("total %.1f us" % ((prev_end - t0) / 1e3))
PY
rm -rf gpurun_out/r4aj
