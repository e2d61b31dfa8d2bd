#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3n; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $O/pytest.log
for e in segments fused segments fused; do python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --lowpass-engine $e 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$e', d['value'], d['ms_per_step'], d['extra']['stage_ms_per_cutoff'], d['roofline']['traffic'])"; done | tee $O/cfg3_engines.log
