#!/bin/bash
# round 5 call an: evaluate() with the zero-phase IIR degradations next to the FFT low-pass (4 filters x 3 cutoffs x 3 orders = 36 keys + 1): where the time goes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8"
PASSES=2 HOSTPROF=1 timeout 600 python tools/exp_e2e.py 2>&1 | tail -19 | cut -c1-200 | tee gpurun_out/r5an_e2e.log
rm -rf gpurun_out/r5an_trace
PASSES=1 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5an_trace -o e2e --output-format csv -- python tools/exp_e2e.py > /dev/null 2>&1
F=$(find gpurun_out/r5an_trace -name '*kernel_stats.csv' | head -1); head -10 "$F" | cut -c1-170 | tee -a gpurun_out/r5an_e2e.log
find gpurun_out/r5an_trace -name '*kernel_trace.csv' -delete
