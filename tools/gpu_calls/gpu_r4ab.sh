#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
BATCHES=128 STAGES=1 NO_PROFILE=1 python tools/exp_e2e_profile.py 2>&1 | tail -4 | tee gpurun_out/r4ab_e2e.log
BATCHES=128 python tools/exp_e2e_profile.py 2>&1 | tail -30 | tee -a gpurun_out/r4ab_e2e.log
