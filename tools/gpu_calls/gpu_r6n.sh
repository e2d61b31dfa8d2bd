#!/bin/bash
# round 6 call n: the packed IIR wavefront (G = sections lanes per utterance, no LDS, 16-sample register chunks) - bit-exact tests, ms per
# launch against the round-5 build, the evaluate() pass with the 36 IIR keys
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "sosfiltfilt or iir" 2>&1 | tail -5 | tee gpurun_out/r6n_tests.log
for F in 64 128 256; do
  for L in tools/_build/libssrhip_iir_r5.so ""; do SSR_DEV_LIB=$L FILES=$F timeout 300 python tools/exp_iir.py 2>&1 | tail -1; done
done | tee gpurun_out/r6n_iir_ab.log
PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-200 | tee gpurun_out/r6n_e2e.log
