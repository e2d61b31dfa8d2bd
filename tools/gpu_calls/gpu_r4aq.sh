#!/bin/bash
# round 4 call aq: streaming (nt) stores for the magnitude rows of k_stft_wave and the output of k_resample_chain - A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for L in "" tools/_build/libssrhip_nt.so; do
  echo -n "${L:-product} "; SSR_DEV_LIB=$L _ONE=1 python tools/exp_ssim.py 2>/dev/null | tail -1
  echo -n "${L:-product} "; SSR_DEV_LIB=$L N_ITEMS=12500 python tools/exp_chain.py 2>/dev/null | tail -1
done; done | tee gpurun_out/r4aq_nt.log
