#!/bin/bash
# round 5 call am: kernel stats of the default bench command on the shipped build (k_ssim and the sinc resampler changed since call k)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=$PWD; OUT=$R/gpurun_out/r5am_stats; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT.log 2>&1; echo "rc=$?"
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
F=$(find $OUT -name '*kernel_stats.csv' | head -1); head -8 "$F" | cut -c1-180
grep -h -o '{"metric.*' $OUT.log | tail -1 | cut -c1-300
