#!/bin/bash
# round 5 call g: GPU tests touching the low-pass engines, then bench cfg3 on the product default engine
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "lowpass or conv or fdomain or cfg3 or multi or cfg1 or reference_waveform" > gpurun_out/r5g_tests.log 2>&1
echo "tests rc=$?"; tail -25 gpurun_out/r5g_tests.log
timeout 600 python bench.py --config cfg3 --steps 3 --warmup 1 > gpurun_out/r5g_bench_cfg3.log 2>&1
echo "bench rc=$?"; tail -3 gpurun_out/r5g_bench_cfg3.log | cut -c1-3000
