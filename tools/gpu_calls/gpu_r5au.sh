#!/bin/bash
# round 5 call au: tools/stress_degrade.py (its low-pass line now against oracle/tl_chain.c, bit for bit) on the shipped build, two seeds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 400 python tools/stress_degrade.py 2>&1 | tail -3 | cut -c1-300 | tee gpurun_out/r5au_stress.log
SEED=9 CASES=40 timeout 400 python tools/stress_degrade.py 2>&1 | tail -3 | cut -c1-300 | tee -a gpurun_out/r5au_stress.log
