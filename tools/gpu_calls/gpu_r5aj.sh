#!/bin/bash
# round 5 call aj: the GPU suite and the bench lines at the head of the round (after the sinc resampler and k_ssim changes)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r5aj_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5aj_tests.log
grep -E "passed|failed|error|rc=" gpurun_out/r5aj_tests.log | tail -4
timeout 600 python bench.py > gpurun_out/r5aj_bench.log 2> gpurun_out/r5aj_bench.err; echo "bench rc=$?"
timeout 400 python bench.py --config cfg3 > gpurun_out/r5aj_bench_cfg3.log 2>> gpurun_out/r5aj_bench.err; echo "cfg3 rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r5aj_bench.log", "gpurun_out/r5aj_bench_cfg3.log"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f, d["metric"], d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:300])
            e = d.get("extra", {}).get("evaluate_end_to_end")
            if e: print(json.dumps(e)[:900])
            print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
PY
tail -3 gpurun_out/r5aj_bench.err
