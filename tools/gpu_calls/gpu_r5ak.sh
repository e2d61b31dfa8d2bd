#!/bin/bash
# round 5 call ak: randomized sinc-resampler parity sweep (random rate pairs, ragged batches, kaiser_fast, special signals)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 500 python tools/stress_sinc.py 2>&1 | tail -2 | tee gpurun_out/r5ak_stress_sinc.log
SEED=11 TRIALS=16 timeout 400 python tools/stress_sinc.py 2>&1 | tail -1 | tee -a gpurun_out/r5ak_stress_sinc.log
