#!/bin/bash
# round 4 call t: fused resample chain - parity test, then cfg5 bench fused vs two calls
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resample_chain or resampler_bit_exact" 2>&1 | tail -25 > gpurun_out/r4t_tests.log
cat gpurun_out/r4t_tests.log
if grep -q "failed\|error" gpurun_out/r4t_tests.log; then exit 0; fi
for mode in fused two-calls; do
timeout 600 python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --resample-chain $mode 2>gpurun_out/r4t_bench_$mode.err | tee gpurun_out/r4t_bench_cfg5_$mode.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms']); print(json.dumps(d.get('extra',{}).get('stage_ms')))"
done
