#!/bin/bash
# round 5 call ab: persistent device twins of the PCM arenas (no torch.empty on the upload stream per batch): tests, evaluate(), host seconds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "pipelined or eval or helper or io or flac or load or upload or decode or packed" 2>&1 | grep -E "passed|failed|rror" | tee gpurun_out/r5ab_tests.log
for BF in 64 64 128 32; do BATCH_FILES=$BF timeout 200 python tools/exp_e2e.py 2>&1 | tail -1; done | tee gpurun_out/r5ab_e2e.log
HOSTPROF=1 BATCH_FILES=64 PASSES=3 timeout 200 python tools/exp_e2e.py 2>&1 | tail -16 | tee -a gpurun_out/r5ab_e2e.log
