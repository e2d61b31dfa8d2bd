#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3o; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $O/pytest.log
for i in 1 2; do
SSR_DEV_LIB=tools/_build/libssrhip_r02.so python tools/exp_api_true.py 2>&1 | tail -1 | sed 's/^/r02 (M=2048): /'
python tools/exp_api_true.py 2>&1 | tail -1 | sed 's/^/M=1536:       /'
done | tee $O/api_true.log
python tools/exp_rates.py 2>&1 | tail -8 | tee $O/rates.log
