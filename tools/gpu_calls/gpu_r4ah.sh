#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for L in "" tools/_build/libssrhip_tl_nob.so tools/_build/libssrhip_tl_noa.so; do
  echo "== ${L:-product}"; SSR_DEV_LIB=$L timeout 300 python tools/exp_tlconv.py 2>&1 | grep "^cut" | sed -n '4p;8p'
done | tee gpurun_out/r4ah_gemm_exp.log
