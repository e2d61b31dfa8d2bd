#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3d; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
for c in cfg4 cfg3 cfg5; do python bench.py --config $c --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$c.json")); print("$c", d["value"], d["unit"], d["ms_per_step"], d["scaling"], d["extra"].get("parity_vs_oracle_max_rel_err"), d["roofline"]["frac"], d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
except Exception as e: print("ERR", e)
PY
done
for st in lsd stft ssim; do STAGE=$st python tools/exp_clocks.py 2>&1 | tail -2 | cut -c1-700; done | tee $O/clocks.log
