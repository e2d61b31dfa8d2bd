#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3h; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
for i in 1 2; do
for hop in 441 512; do
  SSR_NO_FUSED_OLA=1 HOP=$hop _ONE=1 SSR_DEV_LIB=tools/_build/libssrhip_knobs.so python tools/exp_lowpass.py | sed 's/^/paired segments + k_ola_paired: /'
  HOP=$hop _ONE=1 SSR_DEV_LIB=tools/_build/libssrhip_knobs.so python tools/exp_lowpass.py | sed 's/^/fused group kernel:            /'
done; done 2>&1 | tee $O/lowpass_ab.log
python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; python - <<PY
import json
d=json.load(open("$O/bench_cfg3.json")); print("cfg3", d["value"], d["ms_per_step"], d["extra"]["stage_ms_per_cutoff"])
PY
