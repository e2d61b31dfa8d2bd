#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -k "resample or cfg5" 2>&1 | tail -5
python tools/exp_chain.py; N_ITEMS=12500 python tools/exp_chain.py; FUSED=0 python tools/exp_chain.py; FUSED=0 N_ITEMS=12500 python tools/exp_chain.py
