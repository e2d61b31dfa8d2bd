#!/bin/bash
# round 4 call y: cfg3 with the segments / fused float64 low-pass engines, alternating, same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for eng in segments fused; do
python bench.py --config cfg3 --steps 8 --warmup 3 --no-cpu-baseline --no-side --lowpass-engine $eng 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$eng', d['value'], d['ms_per_step'], json.dumps(d.get('extra',{}).get('stage_ms')))"
done; done | tee gpurun_out/r4y_cfg3_engines.log
