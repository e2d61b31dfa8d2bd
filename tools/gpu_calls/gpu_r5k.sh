#!/bin/bash
# round 5 call k: parity + timing after the fold change, then the round-5 profile set (kernel stats of every config + PMC traffic)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
UTT=1024 timeout 400 python tools/exp_tlconv.py > gpurun_out/r5k_tlconv.log 2>&1
echo "rc=$?"; grep "mismatching [1-9]\|^cut\|^multi (" gpurun_out/r5k_tlconv.log | tail -14
PMC=1 timeout 1500 bash tools/collect_profiles_r05.sh r05 > gpurun_out/r5k_profiles.log 2>&1
echo "profiles rc=$?"; tail -100 gpurun_out/r5k_profiles.log | cut -c1-260
