#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x -k "resample_chain or sharded_two_processes" 2>&1 | tail -6
