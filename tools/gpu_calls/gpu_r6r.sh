#!/bin/bash
# round 6 call r: ssr_pair_metrics_multi_est64 (two float64 estimates per complex transform on the rotating engine) - parity tests, evaluate() with the 36 IIR keys
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -q -x -m gpu -k "multi or est64 or float64_estimate or iir" 2>&1 | tail -8 | tee gpurun_out/r6r_tests.log
PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-200 | tee gpurun_out/r6r_e2e.log
HOSTPROF=1 PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -24 | tee gpurun_out/r6r_hostprof.log
