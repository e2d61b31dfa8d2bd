#!/bin/bash
# round 6 call o: packed IIR wavefront with the pipelined step order - bit-exact tests, ms per launch and ns per step per group width
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "sosfiltfilt or iir" 2>&1 | tail -3 | tee gpurun_out/r6o_tests.log
for F in 64 256; do
  for L in tools/_build/libssrhip_iir_r5.so ""; do SSR_DEV_LIB=$L PER_G=1 FILES=$F timeout 300 python tools/exp_iir.py 2>&1 | tail -1; done
done | tee gpurun_out/r6o_iir_ab.log
PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-200 | tee gpurun_out/r6o_e2e.log
