#!/bin/bash
# round 5 call u: evaluate() with the descriptor uploads staged through one page-locked ring; idle gaps; the GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for BF in 128 64 256; do BATCH_FILES=$BF timeout 200 python tools/exp_e2e.py 2>&1 | tail -1; done | tee gpurun_out/r5u_e2e.log
rm -rf gpurun_out/r5u_trace
PASSES=3 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r5u_trace -o e2e --output-format csv -- python tools/exp_e2e.py 2>&1 | tail -1 | tee -a gpurun_out/r5u_e2e.log
F=$(find gpurun_out/r5u_trace -name '*kernel_trace.csv' | head -1)
WINDOW_MS=100 python tools/trace_gaps.py "$F" | tee -a gpurun_out/r5u_e2e.log
find gpurun_out/r5u_trace -name '*kernel_trace.csv' -delete
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -3 | tee gpurun_out/r5u_tests.log
