#!/bin/bash
# round 4, call 4: ssr_pair_metrics_multi
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "multi" 2>&1 | tail -30
