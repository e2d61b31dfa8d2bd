#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "random_rate_pairs or resampler_bit_exact or cfg5" 2>&1 | tail -6
for i in 1 2; do NO_MFMA=1 python tools/exp_resample.py 2>&1 | tail -1; done
NO_MFMA=1 UTT=12500 ITERS=3 python tools/exp_resample.py 2>&1 | tail -1
