#!/bin/bash
# round 4 call r: packed k_specred_wave - multi tests + cfg3 bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu -x -k "multi" 2>&1 | tail -5 > gpurun_out/r4r_tests.log
cat gpurun_out/r4r_tests.log
python bench.py --config cfg3 --steps 5 --warmup 2 2>gpurun_out/r4r_bench.err | tee gpurun_out/r4r_bench_cfg3.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d.get('extra',{}).get('stage_ms'), indent=0))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4r_prof -o cfg3 -- python $GRAFT_REPO_ROOT/bench.py --config cfg3 --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r4r_prof/*kernel_stats.csv | head -1); head -12 "$f" | cut -c1-150
