#!/bin/bash
# round 5 call ac: the reads of batch k+1 started after batch k is queued (reader threads off the launching thread's back)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for BF in 64 64 128 32; do BATCH_FILES=$BF timeout 200 python tools/exp_e2e.py 2>&1 | tail -1; done | tee gpurun_out/r5ac_e2e.log
HOSTPROF=1 BATCH_FILES=64 PASSES=3 timeout 200 python tools/exp_e2e.py 2>&1 | tail -16 | tee -a gpurun_out/r5ac_e2e.log
