#!/bin/bash
# round 5 call o: where an evaluate() pass spends its GPU time (kernel trace), and its sensitivity to the files-per-launch batch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/exp_e2e.py 2>&1 | tail -2 | tee gpurun_out/r5o_e2e.log
BATCH_FILES=512 timeout 200 python tools/exp_e2e.py 2>&1 | tail -1 | tee -a gpurun_out/r5o_e2e.log
BATCH_FILES=64 timeout 200 python tools/exp_e2e.py 2>&1 | tail -1 | tee -a gpurun_out/r5o_e2e.log
rm -rf gpurun_out/r5o_trace
PASSES=5 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r5o_trace -o e2e --output-format csv -- python tools/exp_e2e.py 2>&1 | tail -1 | tee -a gpurun_out/r5o_e2e.log
F=$(find gpurun_out/r5o_trace -name '*kernel_stats.csv' | head -1); head -25 "$F" | cut -c1-200 | tee -a gpurun_out/r5o_e2e.log
find gpurun_out/r5o_trace -name '*kernel_trace.csv' -delete
