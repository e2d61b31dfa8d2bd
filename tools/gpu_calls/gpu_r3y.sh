#!/bin/bash
# matrix-core polyphase resampler: parity tests + stage timings against the bit-exact kernel
mkdir -p gpurun_out/r3y
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "resampler" 2>&1 | tail -6 | tee gpurun_out/r3y/pytest.txt
for i in 1 2 3; do timeout 300 python tools/exp_resample.py 2>&1 | tail -1; done | tee gpurun_out/r3y/resample.txt
UTT=12500 ITERS=3 timeout 300 python tools/exp_resample.py 2>&1 | tail -1 | tee -a gpurun_out/r3y/resample.txt
