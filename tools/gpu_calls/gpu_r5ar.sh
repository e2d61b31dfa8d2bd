#!/bin/bash
# round 5 call ar: the multi-design IIR launch with 16-lane groups (one update_dpp with the sample as the out-of-row value: no select on the step's chain) against 8-lane groups
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for R in 1 2; do for L in "" tools/_build/libssrhip_iir16.so; do SSR_DEV_LIB=$L timeout 200 python tools/exp_iir.py 2>&1 | tail -1; done; done | tee gpurun_out/r5ar_iir.log
SSR_DEV_LIB=tools/_build/libssrhip_iir16.so FILES=37 timeout 200 python tools/exp_iir.py 2>&1 | tail -1 | tee -a gpurun_out/r5ar_iir.log
