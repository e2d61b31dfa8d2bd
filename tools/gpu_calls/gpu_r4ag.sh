#!/bin/bash
# round 4 call ag: where k_tl_gemm's matrix pipe idles - timing-only builds (wrong results): no chunk barrier / one chain / tiles staged once
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for L in "" tools/_build/libssrhip_tl_nobarrier.so tools/_build/libssrhip_tl_nochain.so tools/_build/libssrhip_tl_noload.so; do
  echo "== ${L:-product}"; SSR_DEV_LIB=$L timeout 300 python tools/exp_tlconv.py 2>&1 | grep "^cut" | sed -n '4p;8p'
done | tee gpurun_out/r4ag_gemm_exp.log
