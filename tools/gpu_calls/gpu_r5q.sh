#!/bin/bash
# round 5 call q: host seconds per stage of an evaluate() pass (no synchronisation added)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
HOSTPROF=1 BATCH_FILES=64 PASSES=3 timeout 200 python tools/exp_e2e.py 2>&1 | tail -16 | tee gpurun_out/r5q_e2e.log
HOSTPROF=1 BATCH_FILES=32 PASSES=3 timeout 200 python tools/exp_e2e.py 2>&1 | tail -16 | tee -a gpurun_out/r5q_e2e.log
