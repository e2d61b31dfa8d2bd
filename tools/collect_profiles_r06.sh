#!/bin/bash
# Run ON THE GPU BOX (through gpurun): round-6 profile set - ONE bench command per summary, so that every `roofline.frac` of a bench
# line can be recomputed from one CSV row:
#  * rocprofv3 --kernel-trace --stats of `bench.py --config <cfg> --no-side` for cfg2 (-> cfg2only: k_stft_wave<double, false, true,
#    true>, k_ssim<4, true>, k_finalize and nothing else of ours), apitrue (AudioMetrics(48000): 2229 / 480), cfg3 (product default
#    low-pass engine), cfg3f64 (--lowpass-engine segments), cfg4, cfg5;
#  * PMC=1: the two TCC counter passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950; counters are collected in their own
#    runs, with --kernel-trace only) for cfg2 / apitrue / cfg3 / cfg5.
# Every step under its own timeout.  Outputs under gpurun_out/prof_<tag>/; tools/pmc_to_json.py <tag> turns them into profiles/<tag>_*.
TAG=${1:-r06}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in cfg2only apitrue cfg3 cfg4 cfg5 cfg3f64; do
  rm -rf $OUT/stats_$C
  CFG=$C; ENG=""
  [ "$C" = "cfg2only" ] && CFG=cfg2
  [ "$C" = "cfg3f64" ] && { CFG=cfg3; ENG="--lowpass-engine segments"; }
  CMD="python $R/bench.py --config $CFG $ENG --steps 5 --warmup 2 --no-cpu-baseline --no-side"
  timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$C -o s -- $CMD > $OUT/stats_$C.log 2>&1; echo "stats $C rc=$?"
  grep -h -o '{"metric.*' $OUT/stats_$C.log | tail -1 > $OUT/bench_under_rocprof_$C.json
done
if [ "${PMC:-0}" = "1" ]; then
  for P in FETCH_SIZE WRITE_SIZE; do
    for C in cfg2 api cfg3 cfg5; do
      rm -rf $OUT/pmc_${P}_$C
      CFG=$C; [ "$C" = "api" ] && CFG=apitrue
      timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_${P}_$C -o p -- python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-side > $OUT/pmc_${P}_$C.log 2>&1; echo "$P $C rc=$?"
    done
  done
fi
# summarise ON the box and drop the raw traces (gpurun copies at most 64 MiB back)
(cd $R && python tools/pmc_to_json.py $TAG > $OUT/summary.log 2>&1; mkdir -p $OUT/summary && cp profiles/${TAG}_* $OUT/summary/ 2>/dev/null)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; ls $OUT/summary; tail -70 $OUT/summary.log
