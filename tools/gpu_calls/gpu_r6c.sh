#!/bin/bash
# round 6 call c: PAIRED with the in-place lane-0 rotation and later request points in the running-sums variant, against the
# ascending order; the whole GPU suite on the new library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2 3; do
  for L in "" tools/_build/libssrhip_asc.so; do SSR_DEV_LIB=$L NO_CHECK=1 timeout 200 python tools/exp_stage.py 2>&1 | tail -1; done
done | tee gpurun_out/r6c_paired_ab.log
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -8 | tee gpurun_out/r6c_tests.log
