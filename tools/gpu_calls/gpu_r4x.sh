#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "conv or tl_chain or helper_options" 2>&1 | tail -4
timeout 600 python tools/exp_tlconv.py 2>&1 | tail -15 | tee gpurun_out/r4x_tlconv.log
