#!/bin/bash
# randomized parity sweeps after the k_ssim rewrite (+ a default bench line with the single reduction kernel per step)
mkdir -p gpurun_out/r3u
for s in 101 102 103 104; do SEED=$s CASES=60 timeout 600 python tools/stress_parity.py 2>&1 | tail -3; done | tee gpurun_out/r3u/stress_parity.txt
for s in 201 202; do SEED=$s CASES=30 timeout 600 python tools/stress_degrade.py 2>&1 | tail -3; done | tee gpurun_out/r3u/stress_degrade.txt
python bench.py --no-side 2>/dev/null | tail -1 > gpurun_out/r3u/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3u/bench.json').read())
print(d['value'], d['ms_per_step'], d['extra']['stage_ms'], d['extra'].get('job_means'))
PY
