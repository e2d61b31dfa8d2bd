#!/bin/bash
# round 4 call al: k_ssim four against eight columns per lane, alone and inside the cfg-2 step, alternating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
  for v in "4:tools/_build/libssrhip_knobs.so" "8:tools/_build/libssrhip_knobs.so" "8:tools/_build/libssrhip_knobs8w1.so"; do
    c=${v%%:*}; L=${v##*:}
    echo -n "cpt=$c $(basename $L) "; SSR_SSIM_CPT=$c _ONE=1 SSR_DEV_LIB=$L python tools/exp_ssim.py 2>/dev/null | tail -1
  done
done | tee gpurun_out/r4al_ssim8.log
