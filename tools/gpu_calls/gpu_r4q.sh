#!/bin/bash
# round 4 call q: ssr_plan_create_ex (FDomainHelper options) parity + conv regression
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "helper_options or conv" 2>&1 | tail -25 > gpurun_out/r4q_tests.log
cat gpurun_out/r4q_tests.log
