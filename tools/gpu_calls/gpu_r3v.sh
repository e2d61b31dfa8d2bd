#!/bin/bash
# API-true 2229/480: rotating four-wave kernel (default) against the three-wave workgroups (knobs build + SSR_R3_THREE_WAVES=1)
mkdir -p gpurun_out/r3v
for i in 1 2 3; do
  echo -n "three waves : "; SSR_DEV_LIB=tools/_build/libssrhip_knobs.so SSR_R3_THREE_WAVES=1 python tools/exp_api_true.py 2>&1 | grep '^{'
  echo -n "rotating    : "; python tools/exp_api_true.py 2>&1 | grep '^{'
done | tee gpurun_out/r3v/api_ab.txt
python -m pytest tests -m gpu -x -q -k "2229 or radix or rn_wave or api or rates or audio_metrics or silent or ragged" 2>&1 | tail -4 | tee gpurun_out/r3v/pytest.txt
SEED=301 CASES=60 timeout 600 python tools/stress_parity.py 2>&1 | tail -1 | tee gpurun_out/r3v/stress.txt
