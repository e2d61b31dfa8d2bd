#!/bin/bash
mkdir -p gpurun_out/r3aa
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tail -12
python bench.py --config cfg5 2>/dev/null | tail -1 > gpurun_out/r3aa/bench_cfg5.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3aa/bench_cfg5.json').read())
print(d['value'], d['ms_per_step'], d['extra']['stage_ms'], d['extra'].get('matrix_core_mode'), d['extra'].get('parity_vs_oracle_max_rel_err'))
PY
