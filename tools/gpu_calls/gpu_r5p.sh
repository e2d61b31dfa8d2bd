#!/bin/bash
# round 5 call p: evaluate() after the descriptor uploads stopped draining the stream (pinned staging) and the metric values of a batch
# are collected after the next batch has been queued; the e2e-relevant GPU tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/exp_e2e.py 2>&1 | tail -2 | tee gpurun_out/r5p_e2e.log
BATCH_FILES=64 timeout 200 python tools/exp_e2e.py 2>&1 | tail -1 | tee -a gpurun_out/r5p_e2e.log
BATCH_FILES=256 timeout 200 python tools/exp_e2e.py 2>&1 | tail -1 | tee -a gpurun_out/r5p_e2e.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "eval or helper or e2e or metric or io or flac or dist" 2>&1 | tail -5 | tee gpurun_out/r5p_tests.log
