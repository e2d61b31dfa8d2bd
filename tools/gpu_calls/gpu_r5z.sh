#!/bin/bash
# round 5 call z: evaluate() with the file reads dealt to 16 tasks (not one future per file); io / flac / helper GPU tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for BF in 64 48 96 64; do BATCH_FILES=$BF timeout 200 python tools/exp_e2e.py 2>&1 | tail -1; done | tee gpurun_out/r5z_e2e.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "eval or helper or io or flac or load or pipelined or decode or upload" 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/r5z_tests.log
