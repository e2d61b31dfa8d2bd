#!/bin/bash
# round 5 call ah: k_ssim with the column-sum hand-off stores as single ds_write_b64 (-DSSR_SSIM_SPLIT_STORES) against the default
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for R in 1 2 3; do
for L in "" tools/_build/libssrhip_ssimst.so; do
  SSR_DEV_LIB=$L NO_CHECK=$([ $R != 1 ] && echo 1) timeout 200 python tools/exp_stage.py 2>&1 | tail -1
done; done | tee gpurun_out/r5ah_ssimst.log
