#!/bin/bash
# round 6 call k: host-side profile of an evaluate() pass with the 36 IIR keys at the new defaults
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
HOSTPROF=1 PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee gpurun_out/r6k_hostprof.log
