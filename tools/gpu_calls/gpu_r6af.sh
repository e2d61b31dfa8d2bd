#!/bin/bash
# round 6 call af: eight autonomous IIR waves per workgroup (the launch's waves on an eighth of the CUs) - bit-exact tests, launch time,
# the evaluate() pass with the 36 IIR keys on one and on two alternating streams
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "sosfiltfilt or iir" 2>&1 | grep -E "passed|failed" | tail -2
SEED=11 ROUNDS=40 timeout 400 python tools/stress_iir.py 2>&1 | tail -1
for L in tools/_build/libssrhip_iir_r5.so ""; do SSR_DEV_LIB=$L PER_G=1 FILES=64 timeout 300 python tools/exp_iir.py 2>&1 | tail -1 | cut -c1-600; done
for rep in 1 2; do for S in 1 2; do
  echo "streams $S, 36 IIR keys: $(SSR_EVAL_STREAMS=$S PASSES=4 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-170)"
done; done | tee gpurun_out/r6af_iir_wpg_streams.log
