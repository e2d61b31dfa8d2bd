#!/bin/bash
# round 4 call z: full GPU suite + the profile set (stats + PMC) after the fused chain / packed specred
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r4z_gpu_suite.log
tail -5 gpurun_out/r4z_gpu_suite.log
PMC=1 bash tools/collect_profiles_r04.sh r04 2>&1 | tail -40
