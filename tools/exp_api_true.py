"""Developer experiment: the API-true AudioMetrics(48000) sizes (n_fft 2229 / hop 480), stage timings + oracle check."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B
from oracle import metrics as om

n = int(os.environ.get("PAIRS", "1024"))
dev = torch.device("cuda", 0)
est, tgt = bench.make_inputs(n, dev, 1)
plan = B.get_plan(2229, 480, "f64", dev)
b = B.PairBatch(plan, B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
res = {}
for name, mask, st in [("stft lsd+mag", B.M_LSD | B.M_SSIM, 1), ("stft lsd only", B.M_LSD, 1), ("stft all4+mag", B.M_ALL, 1),
                       ("full lsd+ssim", B.M_LSD | B.M_SSIM, 7)]:
    res[name] = round(bench.event_time_ms(lambda: b.run(mask, stages=st), 5), 3)
got = b.run(B.M_ALL).cpu().numpy()
worst = 0.0
for i in (0, n - 1):
    w = om.evaluation(est[i].cpu().numpy(), tgt[i].cpu().numpy(), n_fft=2229, hop=480)
    w = np.array([w[k] for k in ("lsd", "log_sispec", "sispec", "ssim")])
    worst = max(worst, float(np.abs((got[i] - w) / w).max()))
res["max_rel_err_vs_oracle"] = worst
res["pairs_per_s"] = round(n / res["full lsd+ssim"] * 1e3, 1)
print(json.dumps(res))
