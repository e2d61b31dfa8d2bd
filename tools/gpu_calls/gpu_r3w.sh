#!/bin/bash
# k_ssim on the 401 x 1115 images of 2229/480: the strided CPT = 6 kernel (3 strips) against the CONTIG CPT = 4 kernel (5 strips)
mkdir -p gpurun_out/r3w
for i in 1 2 3; do
  echo -n "cpt 6 strided: "; SSR_DEV_LIB=tools/_build/libssrhip_knobs.so python tools/exp_api_true.py 2>&1 | grep '^{'
  echo -n "cpt 4 contig : "; SSR_DEV_LIB=tools/_build/libssrhip_knobs.so SSR_SSIM_CPT=4 python tools/exp_api_true.py 2>&1 | grep '^{'
done | tee gpurun_out/r3w/ssim_cpt.txt
