#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3o; mkdir -p $O
for i in 1 2; do
SSR_DEV_LIB=tools/_build/libssrhip_nopk.so python tools/exp_api_true.py 2>&1 | tail -1 | sed 's/^/bin by bin:  /'
python tools/exp_api_true.py 2>&1 | tail -1 | sed 's/^/packed pairs: /'
done | tee $O/api_true3.log
CASES=40 SEED=31 timeout 600 python tools/stress_parity.py 2>&1 | grep "MISS\|cases" | cut -c1-600 | tee $O/stress31.log
SSR_DEV_LIB=tools/_build/libssrhip_r02.so CASES=40 SEED=31 timeout 600 python tools/stress_parity.py 2>&1 | grep "MISS\|cases" | cut -c1-600 | sed 's/^/r02: /' | tee -a $O/stress31.log
