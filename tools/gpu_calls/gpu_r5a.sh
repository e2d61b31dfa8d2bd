#!/bin/bash
# round 5 call a: the rebuilt conv engine (k_tl_fwd / k_tl_inv: torch's accumulation order, LDS-DMA operands, shared forward) -
# parity against oracle/tl_chain.c and torch-CPU conv1d, then the time of the tile variants at cfg-3's size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SSR_DEV_LIB=tools/_build/libssrhip_knobs.so VARIANTS="128:2,128:3,64:2" UTT=1024 timeout 900 python tools/exp_tlconv.py > gpurun_out/r5a_tlconv.log 2>&1
echo "rc=$?"; tail -80 gpurun_out/r5a_tlconv.log
