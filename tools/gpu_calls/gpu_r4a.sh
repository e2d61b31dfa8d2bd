#!/bin/bash
# round 4, call 1: the conv (reference-arithmetic) low-pass engine: bit-exactness against oracle/tl_chain.c + first timing
python -c "from oracle import tl_chain; tl_chain.build()"
timeout 900 python tools/exp_tlconv.py 2>&1 | tail -40
