#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3c; mkdir -p $O
rocm-smi --showpower --showmaxpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" | head -12
for i in 1 2; do
for v in oor oorx2 ldx4 ldx4oorx2; do
  SSR_DEV_LIB=tools/_build/libssrhip_$v.so NO_CHECK=1 python tools/exp_stage.py 2>&1 | tail -1 | sed "s/^/$v: /"
done
NO_CHECK=1 python tools/exp_stage.py 2>&1 | tail -1 | sed 's/^/new: /'
done | tee $O/ab.log
