#!/bin/bash
# round 6 call g: the float64-estimate epilogue with float32 logarithms (SSR_EST64_FAST) against the previous commit's library:
# evaluate() with the 36 IIR keys, and every float64 / IIR parity test
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
  for L in "" tools/_build/libssrhip_pre_est64fast.so; do
    echo "lib=${L:-shipped}"; SSR_DEV_LIB=$L BATCH_FILES=64 PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-200
  done
done | tee gpurun_out/r6g_e2e_iir_ab.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -k "float64 or est64 or f64 or iir or sos or lowpass_filter or helper" 2>&1 | tail -6 | cut -c1-300 | tee gpurun_out/r6g_tests.log
