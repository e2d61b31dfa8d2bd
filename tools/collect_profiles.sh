#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats of the bench command of every config, and (PMC=1) the two
# TCC counter passes of the default config (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950; counters are
# collected in their own runs, with --kernel-trace only).  Outputs under gpurun_out/prof_<tag>/.
#   tools/collect_profiles.sh <tag> [configs...]        e.g.  tools/collect_profiles.sh r02 cfg2 cfg3 cfg5
TAG=${1:-r02}; shift
CFGS=${@:-cfg2 cfg3 cfg5}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in $CFGS; do
  rm -rf $OUT/stats_$C
  CMD="python $R/bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline --no-side"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$C -o s -- $CMD > $OUT/stats_$C.log 2>&1; echo "stats $C rc=$?"
  grep -h -o '{"metric.*' $OUT/stats_$C.log | tail -1 > $OUT/bench_under_rocprof_$C.json
done
if [ "${PMC:-0}" = "1" ]; then
  for P in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_$P
    timeout 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$P -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side > $OUT/pmc_$P.log 2>&1; echo "$P rc=$?"
    for C in ${PMC_CFGS:-}; do          # the same two passes for further configs (their own kernels: low-pass, resampler)
      rm -rf $OUT/pmc_${P}_$C
      timeout 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_${P}_$C -o p -- python $R/bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline --no-side > $OUT/pmc_${P}_$C.log 2>&1; echo "$P $C rc=$?"
    done
  done
fi
find $OUT -name "*kernel_stats.csv" | head
