#!/bin/bash
# round 6 call v (after the packed IIR wavefront, ssr_pair_metrics_multi_est64, two-stream evaluate()): the whole GPU suite on the head of the round, smoke, the default bench line, apitrue / cfg3 / cfg4 / cfg5 bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -12 | cut -c1-400 | tee gpurun_out/r6v_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
timeout 900 python bench.py > gpurun_out/r6v_bench.json 2> gpurun_out/r6v_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/r6v_bench.json | cut -c1-900
for C in apitrue cfg3 cfg4 cfg5; do timeout 600 python bench.py --config $C --no-side > gpurun_out/r6v_bench_$C.json 2>/dev/null; echo "$C rc=$?"; tail -1 gpurun_out/r6v_bench_$C.json | cut -c1-400; done
