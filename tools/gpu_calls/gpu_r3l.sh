#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3l; mkdir -p $O
for seed in 11 12; do CASES=40 SEED=$seed timeout 600 python tools/stress_parity.py 2>&1 | tail -2; done | tee $O/stress_parity.log
for seed in 21 22; do CASES=25 SEED=$seed timeout 600 python tools/stress_degrade.py 2>&1 | tail -2; done | tee $O/stress_degrade.log
# same-box A/B of cfg-3 with and without the fused overlap-add (knobs build)
for v in 1 0 1 0; do
SSR_NO_FUSED_OLA=$v SSR_DEV_LIB=tools/_build/libssrhip_knobs.so python - <<PY 2>&1 | tail -1
import sys, runpy, json, io, contextlib
sys.path.insert(0, "tools"); import devlib; devlib.select()
sys.argv = ["bench.py", "--config", "cfg3", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print("SSR_NO_FUSED_OLA=$v", d["value"], d["ms_per_step"], d["extra"]["stage_ms_per_cutoff"])
PY
done | tee $O/cfg3_ab.log
