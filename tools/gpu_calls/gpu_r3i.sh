#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3i; mkdir -p $O
for i in 1 2; do
  python tools/exp_resample.py | sed 's/^/exact (SciPy bits): /'
  SSR_DEV_LIB=tools/_build/libssrhip_rsfma.so python tools/exp_resample.py | sed 's/^/fused multiply-add: /'
done 2>&1 | grep -v amdgpu.ids | tee $O/resample_fma.log
PMC=1 bash tools/collect_profiles_r03.sh r03 2>&1 | tail -25
