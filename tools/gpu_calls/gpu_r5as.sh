#!/bin/bash
# round 5 call as: the shipped build with 16-lane groups in the multi-design IIR launch: parity tests, launch time, evaluate() with 36 IIR keys
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "sosfilt or iir or resident or lowpass" 2>&1 | grep -E "passed|failed|rror" | tee gpurun_out/r5as_tests.log
timeout 200 python tools/exp_iir.py 2>&1 | tail -1 | tee gpurun_out/r5as_iir.log
export IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8"
PASSES=3 BATCH_FILES=64 timeout 600 python tools/exp_e2e.py 2>&1 | tail -2 | cut -c1-200 | tee -a gpurun_out/r5as_iir.log
PASSES=3 BATCH_FILES=128 timeout 600 python tools/exp_e2e.py 2>&1 | tail -2 | cut -c1-200 | tee -a gpurun_out/r5as_iir.log
