#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "multi" 2>&1 | tail -5
bash tools/gpu_calls/gpu_r4f.sh
timeout 600 python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['extra']['stage_ms'])"
