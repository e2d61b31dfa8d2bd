#!/bin/bash
# round 5 call ad: idle gaps of the GPU inside evaluate() passes after the host-side changes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/r5ad_trace
BATCH_FILES=64 PASSES=3 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r5ad_trace -o e2e --output-format csv -- python tools/exp_e2e.py 2>&1 | tail -1 | tee gpurun_out/r5ad_e2e.log
F=$(find gpurun_out/r5ad_trace -name '*kernel_trace.csv' | head -1)
WINDOW_MS=95 python tools/trace_gaps.py "$F" | tee -a gpurun_out/r5ad_e2e.log
find gpurun_out/r5ad_trace -name '*kernel_trace.csv' -delete
