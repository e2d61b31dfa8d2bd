#!/bin/bash
# round 5 call l: clean bench lines (no profiler): the default run (cfg2 + side figures + CPU baseline), cfg3, cfg4, cfg5
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r5l_bench.log 2>&1; echo "cfg2 rc=$?"; tail -1 gpurun_out/r5l_bench.log | cut -c1-600
for C in cfg3 cfg4 cfg5; do
  timeout 400 python bench.py --config $C --no-side > gpurun_out/r5l_bench_$C.log 2>&1; echo "$C rc=$?"; tail -1 gpurun_out/r5l_bench_$C.log | cut -c1-400
done
