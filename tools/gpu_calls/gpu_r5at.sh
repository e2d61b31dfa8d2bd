#!/bin/bash
# round 5 call at: single-design IIR launches on 16-lane groups up to 4096 utterances: parity tests (both group widths), stress_degrade, the suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "sosfilt or iir or lowpass or resident" 2>&1 | grep -E "passed|failed|rror" | tee gpurun_out/r5at_tests.log
timeout 400 python tools/stress_degrade.py 2>&1 | tail -2 | cut -c1-400 | tee -a gpurun_out/r5at_tests.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | grep -E "passed|failed|rror" | tee -a gpurun_out/r5at_tests.log
