#!/bin/bash
# round 4 call ak: final check - full GPU suite, smoke, bench lines of every config
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/bench_r04
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r4ak_gpu_suite.log; tail -3 gpurun_out/r4ak_gpu_suite.log
python bench.py > gpurun_out/bench_r04/cfg2.json 2> gpurun_out/bench_r04/cfg2.err
for c in cfg3 cfg4 cfg5; do python bench.py --config $c > gpurun_out/bench_r04/$c.json 2> gpurun_out/bench_r04/$c.err; done
python bench.py --config cfg3 --lowpass-engine conv --steps 3 --warmup 1 --no-side > gpurun_out/bench_r04/cfg3conv.json 2> gpurun_out/bench_r04/cfg3conv.err
for c in cfg2 cfg3 cfg4 cfg5 cfg3conv; do python - <<PY
import json
d=json.loads(open("gpurun_out/bench_r04/$c.json").read().strip().splitlines()[-1])
print("$c", d["value"], d["ms_per_step"], d["roofline"]["bound"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value"))
if "$c" == "cfg3": print(json.dumps(d["extra"].get("float32_fft_engine(SSR_F32 plan)")), d["extra"].get("conv_engine_error"))
PY
done
