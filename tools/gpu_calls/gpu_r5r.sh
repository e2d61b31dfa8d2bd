#!/bin/bash
# round 5 call r: cProfile of one evaluate() pass (host hot spots)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
CPROFILE=45 BATCH_FILES=64 PASSES=2 timeout 200 python tools/exp_e2e.py 2>&1 | cut -c1-170 | tail -62 | tee gpurun_out/r5r_e2e.log
