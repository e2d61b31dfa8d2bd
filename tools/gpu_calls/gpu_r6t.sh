#!/bin/bash
# round 6 call t: k_stft_wave with the magnitude rows staged through the exchange array and stored as 16-byte vectors (34 -> 10 stores per
# frame pair) - A/B against the direct-store build (alternating), parity tests of the wave engine
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2 3; do
  for L in tools/_build/libssrhip_iir_r5.so ""; do SSR_DEV_LIB=$L NO_CHECK=1 timeout 200 python tools/exp_stage.py 2>&1 | tail -1 | cut -c1-330; done
done | tee gpurun_out/r6t_rows_via_lds_ab.log
timeout 200 python tools/exp_stage.py 2>&1 | tail -1 | cut -c1-400
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -m gpu -k "pair or stft or cfg2 or multi or metrics or golden" 2>&1 | tail -5 | tee gpurun_out/r6t_tests.log
