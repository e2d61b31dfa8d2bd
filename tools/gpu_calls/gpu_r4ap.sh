#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -m gpu -x -k "cfg5" 2>&1 | tail -4
