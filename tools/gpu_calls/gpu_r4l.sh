#!/bin/bash
# round 4: new GPU tests since the last full run + the round's profile set (stats + PMC)
timeout 1200 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "cfg5 or multi or nccl" 2>&1 | tail -5
PMC=1 bash tools/collect_profiles_r04.sh r04 2>&1 | tail -90
