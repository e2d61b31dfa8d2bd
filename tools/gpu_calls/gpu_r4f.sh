#!/bin/bash
# round 4, call 6: kernel breakdown of the cfg-3 step under rocprofv3
R=$PWD; OUT=$R/gpurun_out/prof_r04q; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg3 -o s -- python $R/bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-side > $OUT/stats_cfg3.log 2>&1; echo rc=$?
python - <<PY
import csv,glob
f=glob.glob("$OUT/stats_cfg3/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
