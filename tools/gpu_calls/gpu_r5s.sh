#!/bin/bash
# round 5 call s: evaluate() with the bus transfer on its own stream and the pipeline-fill batches; the e2e-relevant GPU tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for BF in 128 64 32 256; do BATCH_FILES=$BF timeout 200 python tools/exp_e2e.py 2>&1 | tail -1; done | tee gpurun_out/r5s_e2e.log
HOSTPROF=1 BATCH_FILES=128 PASSES=3 timeout 200 python tools/exp_e2e.py 2>&1 | tail -16 | tee -a gpurun_out/r5s_e2e.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "eval or helper or e2e or metric or io or flac or dist" 2>&1 | tail -3 | tee gpurun_out/r5s_tests.log
