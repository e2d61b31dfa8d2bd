#!/bin/bash
mkdir -p gpurun_out/r3x
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r3x/pytest.txt
python tools/exp_api_true.py 2>&1 | grep '^{' | tee gpurun_out/r3x/api.txt
python tools/exp_rates.py 2>&1 | tail -8 | tee gpurun_out/r3x/rates.txt
SEED=401 CASES=80 timeout 600 python tools/stress_parity.py 2>&1 | tail -1 | tee gpurun_out/r3x/stress.txt
