#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats of the default bench command plus the two
# TCC counter passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950).  Outputs under gpurun_out/prof_<tag>/.
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/stats.log 2>&1; echo "stats rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1; echo "$C rc=$?"
done
grep -h -o '{"metric.*' $OUT/stats.log | tail -1 > $OUT/bench_under_rocprof.json
ls -R $OUT | head -30
