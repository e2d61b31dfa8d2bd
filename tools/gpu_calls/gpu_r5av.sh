#!/bin/bash
# round 5 call av: float64 estimates at 2229 / 480 on the rotating four-wave engine (IN64 variant): parity tests, evaluate() with 36 IIR keys, kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "float64 or est64 or iir or resident or sosfilt" 2>&1 | grep -E "passed|failed|rror" | tee gpurun_out/r5av_tests.log
export IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8"
PASSES=3 BATCH_FILES=64 timeout 600 python tools/exp_e2e.py 2>&1 | tail -2 | cut -c1-200 | tee gpurun_out/r5av_e2e.log
rm -rf gpurun_out/r5av_trace
PASSES=1 BATCH_FILES=64 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5av_trace -o e2e --output-format csv -- python tools/exp_e2e.py > /dev/null 2>&1
F=$(find gpurun_out/r5av_trace -name '*kernel_stats.csv' | head -1); head -7 "$F" | cut -c1-170 | tee -a gpurun_out/r5av_e2e.log
find gpurun_out/r5av_trace -name '*kernel_trace.csv' -delete
