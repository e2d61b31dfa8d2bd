#!/bin/bash
# round 5 call w: the windowed-sinc resampler on the phase-major table (vector loads four taps ahead): parity tests, kernel time, evaluate()
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "sinc or resamp or rate or eval or helper or io or load" 2>&1 | tail -4 | tee gpurun_out/r5w_tests.log
timeout 300 python tools/exp_sinc.py 2>&1 | tail -12 | tee gpurun_out/r5w_sinc.log
timeout 200 python tools/exp_e2e.py 2>&1 | tail -1 | tee gpurun_out/r5w_e2e.log
rm -rf gpurun_out/r5w_trace
PASSES=5 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r5w_trace -o e2e --output-format csv -- python tools/exp_e2e.py 2>&1 | tail -1 | tee -a gpurun_out/r5w_e2e.log
F=$(find gpurun_out/r5w_trace -name '*kernel_stats.csv' | head -1); head -12 "$F" | cut -c1-160 | tee -a gpurun_out/r5w_e2e.log
find gpurun_out/r5w_trace -name '*kernel_trace.csv' -delete
