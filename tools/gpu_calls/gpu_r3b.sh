#!/bin/bash
# round-3 GPU call B: where do the magnitude stores cost their 0.55 ms?  phase clocks per variant, store ablations
R=$PWD; O=$R/gpurun_out/r3b; mkdir -p $O
SSR_DEV_LIB=tools/_build/libssrhip_clk.so NO_CHECK=1 python tools/exp_stage.py > $O/clk.log 2>&1; grep -h "k_stft_wave\|lib" $O/clk.log | cut -c1-400 | head -40
for i in 1 2; do
for v in oor half; do
  SSR_DEV_LIB=tools/_build/libssrhip_$v.so NO_CHECK=1 python tools/exp_stage.py 2>&1 | tail -1 | sed "s/^/$v: /"
done
NO_CHECK=1 python tools/exp_stage.py 2>&1 | tail -1 | sed 's/^/new: /'
done | tee $O/ab.log
