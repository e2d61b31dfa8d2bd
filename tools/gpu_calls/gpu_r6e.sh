#!/bin/bash
# round 6 call e: the new GPU tests (SISpec members, cutoff-index differential); evaluate() with the 36 IIR keys at several batch sizes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -k "member_spread or differential or basic_testee" 2>&1 | tail -8 | cut -c1-400 | tee gpurun_out/r6e_tests.log
for BF in 64 128 256 367; do
  BATCH_FILES=$BF PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-200
done | tee gpurun_out/r6e_e2e_iir_batches.log
