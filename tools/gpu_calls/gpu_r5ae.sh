#!/bin/bash
# round 5 call ae: the GPU suite and the default bench line after the last evaluate() changes (gapped views, device twins, read order)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r5ae_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5ae_tests.log
grep -E "passed|failed|rror|rc=" gpurun_out/r5ae_tests.log | tail -4
timeout 600 python bench.py > gpurun_out/r5ae_bench.log 2> gpurun_out/r5ae_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r5ae_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["metric"], d["value"], d["ms_per_step"])
        print(json.dumps(d["extra"]["evaluate_end_to_end"])[:1000])
PY
