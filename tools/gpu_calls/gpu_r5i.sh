#!/bin/bash
# round 5 call i: the whole GPU suite (per-test timeout), bench cfg3 on the product default engine
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 240 > gpurun_out/r5i_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r5i_tests.log
timeout 400 python bench.py --config cfg3 --steps 3 --warmup 1 > gpurun_out/r5i_bench_cfg3.log 2>&1
echo "bench rc=$?"; tail -2 gpurun_out/r5i_bench_cfg3.log | cut -c1-4500
