#!/bin/bash
# round 6 call u: evaluate() with consecutive batches on two alternating GPU streams - A/B (SSR_EVAL_STREAMS=1 / 2), with and without IIR keys, tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
  for S in 1 2; do
    echo "streams $S, 36 IIR keys: $(SSR_EVAL_STREAMS=$S PASSES=4 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-170)"
    echo "streams $S, FFT key only: $(SSR_EVAL_STREAMS=$S PASSES=7 timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-200)"
  done
done | tee gpurun_out/r6u_streams_ab.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -m gpu -k "evaluate or helper or tree or flac or process or iir_degradation" 2>&1 | tail -5 | tee gpurun_out/r6u_tests.log
