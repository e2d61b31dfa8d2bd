#!/bin/bash
# round 4 call aa: the bench lines of every config (plain runs, defaults) for profiles/r04_bench*.json
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/bench_r04
( time python bench.py ) > gpurun_out/bench_r04/cfg2.json 2> gpurun_out/bench_r04/cfg2.err
for c in cfg3 cfg4 cfg5; do python bench.py --config $c > gpurun_out/bench_r04/$c.json 2> gpurun_out/bench_r04/$c.err; done
python bench.py --config cfg3 --lowpass-engine conv --steps 3 --warmup 1 --no-cpu-baseline --no-side > gpurun_out/bench_r04/cfg3conv.json 2> gpurun_out/bench_r04/cfg3conv.err
for c in cfg2 cfg3 cfg4 cfg5 cfg3conv; do python - <<PY
import json
d=json.loads(open("gpurun_out/bench_r04/$c.json").read().strip().splitlines()[-1])
print("$c", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("cpu_baseline",{}).get("value"))
PY
done
tail -4 gpurun_out/bench_r04/cfg2.err
