#!/bin/bash
# k_ssim launch geometry: row tiles per item (SSR_TARGET_WGS of the knobs build -> rows_per_tile), kernel alone
mkdir -p gpurun_out/r3s
for rep in 1 2; do
for t in 1024 2048 3072 4096 6144 8192 12288; do
  echo -n "target $t: "; SSR_TARGET_WGS=$t SSR_DEV_LIB=tools/_build/libssrhip_knobs.so _ONE=1 python tools/exp_ssim.py 2>&1 | grep '^{'
done; done | tee gpurun_out/r3s/ssim_geom.txt
