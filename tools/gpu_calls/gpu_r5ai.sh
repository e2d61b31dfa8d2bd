#!/bin/bash
# round 5 call ai: the GPU suite with k_ssim's hand-off stores as single ds_write_b64 (the shipped build) and the default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r5ai_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5ai_tests.log
grep -E "passed|failed|rror|rc=" gpurun_out/r5ai_tests.log | tail -4
timeout 600 python bench.py --no-side > gpurun_out/r5ai_bench.log 2> gpurun_out/r5ai_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r5ai_bench.log"):
    if l.startswith("{"):
        d = json.loads(l); print(d["metric"], d["value"], d["ms_per_step"], d["extra"].get("stage_ms"))
PY
