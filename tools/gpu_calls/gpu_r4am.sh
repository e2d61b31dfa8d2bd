#!/bin/bash
# round 4 call am: band-specialised first inverse pass of k_lowpass_wave - parity tests, time per cutoff, cfg3
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x -k "lowpass or cfg3 or fdomain or evaluate" 2>&1 | tail -4
python - <<'PY'
import torch, bench, json
from ssr_eval_amd import backend as B
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
x = (0.1 * torch.randn((1024, bench.N_SAMPLES), generator=g, device=dev)).contiguous()
plan = B.get_plan(2048, 441, "f64", dev)
r = B.Ragged.from_uniform(x)
out = {}
for c in bench.CUT_BINS + [1025]:
    lb = B.LowpassBatch(plan, r, [c] * 1024)
    out[c] = round(bench.event_time_ms(lambda: lb.run(), 6), 4)
print(json.dumps(out))
PY
python bench.py --config cfg3 --steps 8 --warmup 3 --no-cpu-baseline --no-side 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('extra',{}).get('stage_ms')))"
