#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resample_chain" 2>&1 | tail -5
python tools/exp_chain.py; python tools/exp_chain.py; N_ITEMS=12500 python tools/exp_chain.py
