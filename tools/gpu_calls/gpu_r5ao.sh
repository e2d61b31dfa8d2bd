#!/bin/bash
# round 5 call ao: ssr_sosfiltfilt_multi (every design of setting_lowpass_filtering in one launch): parity tests, then evaluate() with 36 IIR keys + the FFT key
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "sosfilt or iir or resident or pipelined or helper or eval" 2>&1 | grep -E "passed|failed|rror" | tee gpurun_out/r5ao_tests.log
export IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8"
PASSES=2 HOSTPROF=1 timeout 600 python tools/exp_e2e.py 2>&1 | tail -19 | cut -c1-200 | tee gpurun_out/r5ao_e2e.log
rm -rf gpurun_out/r5ao_trace
PASSES=1 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5ao_trace -o e2e --output-format csv -- python tools/exp_e2e.py > /dev/null 2>&1
F=$(find gpurun_out/r5ao_trace -name '*kernel_stats.csv' | head -1); head -8 "$F" | cut -c1-170 | tee -a gpurun_out/r5ao_e2e.log
find gpurun_out/r5ao_trace -name '*kernel_trace.csv' -delete
