#!/bin/bash
# round 4: the bench lines (default = cfg2 with side figures, then cfg3 / cfg4 / cfg5)
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; echo "cfg2 rc=$?"; tail -c 600 gpurun_out/r04_bench.err
for C in cfg3 cfg4 cfg5; do timeout 900 python bench.py --config $C --no-side > gpurun_out/r04_bench_$C.json 2> gpurun_out/r04_bench_$C.err; echo "$C rc=$?"; done
python - <<PY
import json
for c in ("", "_cfg3", "_cfg4", "_cfg5"):
    d = json.loads(open("gpurun_out/r04_bench%s.json" % c).read().strip().splitlines()[-1])
    print(c or "cfg2", d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"][:50], d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
    if not c: print(json.dumps(d["extra"])[:3000])
PY
