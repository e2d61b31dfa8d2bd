#!/bin/bash
# round 5 call v: the GPU suite with the fault handler on (call u's run ended without a summary line)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r5v_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5v_tests.log
tail -60 gpurun_out/r5v_tests.log | cut -c1-220
