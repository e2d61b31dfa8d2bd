#!/bin/bash
# round 5 call ap: the GPU suite and smoke() on the shipped build at the head of the round
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r5ap_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5ap_tests.log
grep -E "passed|failed|rror|rc=" gpurun_out/r5ap_tests.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
