#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in 7 8; do SEED=$s timeout 900 python tools/stress_r04.py 2>&1 | tail -2; done | tee gpurun_out/r4ai_stress.log
