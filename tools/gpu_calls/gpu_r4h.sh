#!/bin/bash
for KG in 1 2 3; do
  echo "KG=$KG"; SSR_SPEC_KG=$KG timeout 600 python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['extra']['stage_ms'])"
done
