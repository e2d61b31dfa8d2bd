#!/bin/bash
python -c "from oracle import tl_chain; tl_chain.build()"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_rate_pairs or other_transform_sizes" 2>&1 | tail -25
