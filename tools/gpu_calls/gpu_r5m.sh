#!/bin/bash
# round 5 call m: randomized differential run of the conv engine (tools/stress_r05.py, two seeds), then the whole GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for S in 5 6; do SEED=$S TRIALS=9 timeout 500 python tools/stress_r05.py 2>&1 | tail -1; done | tee gpurun_out/r5m_stress.log
timeout 900 python -m pytest tests -m gpu -q --timeout 240 > gpurun_out/r5m_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5m_tests.log
