#!/bin/bash
# round 5 call n: the whole library under other LLVM scheduling strategies (max-ilp, iterative-ilp) - cfg-2's stages, alternating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for R in 1 2; do
for L in "" tools/_build/libssrhip_maxilp.so tools/_build/libssrhip_iterilp.so; do
  SSR_DEV_LIB=$L NO_CHECK=$([ $R = 2 ] && echo 1) timeout 200 python tools/exp_stage.py 2>&1 | tail -1
done; done | tee gpurun_out/r5n_sched.log
