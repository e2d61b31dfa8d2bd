#!/bin/bash
# round 5 call al: evaluate() files/s against batch_files after the kernel changes (two rounds, alternating)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for R in 1 2; do for BF in 64 96 128 192; do BATCH_FILES=$BF timeout 200 python tools/exp_e2e.py 2>&1 | tail -1 | cut -c1-150; done; done | tee gpurun_out/r5al_e2e.log
