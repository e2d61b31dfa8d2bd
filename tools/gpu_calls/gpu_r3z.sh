#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "resampler" 2>&1 | tail -12
for i in 1 2 3; do echo -n "default       : "; python tools/exp_resample.py 2>&1 | tail -1; done
echo -n "12500         : "; UTT=12500 ITERS=3 python tools/exp_resample.py 2>&1 | tail -1
