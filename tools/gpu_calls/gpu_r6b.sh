#!/bin/bash
# round 6 call b: PAIRED third pass of k_stft_wave (partners in the lane's own registers, no half exchange) against the ascending
# order (-DSSR_WAVE_PAIRED=0), alternating on one box; parity tests of the pair path
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2 3; do
  for L in "" tools/_build/libssrhip_asc.so; do SSR_DEV_LIB=$L NO_CHECK=1 timeout 200 python tools/exp_stage.py 2>&1 | tail -1; done
done | tee gpurun_out/r6b_paired_ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "pair or stft or metrics or golden" 2>&1 | tail -5 | tee gpurun_out/r6b_tests.log
