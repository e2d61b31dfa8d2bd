"""Developer experiment: time the two polyphase stages of BASELINE cfg-5 (16 k -> 44.1 k -> 48 k)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B

n = int(os.environ.get("UTT", "4096"))
dev = torch.device("cuda", 0)
x = (0.1 * torch.randn((n, 64000), device=dev)).contiguous()
s1 = B.ResampleBatch(B.Ragged.from_uniform(x), 44100, 16000)
s2 = B.ResampleBatch(s1.out_ragged(), 48000, 44100)
it = int(os.environ.get("ITERS", "5"))
res = {"utt": n, "ms_441_160": round(bench.event_time_ms(s1.run, it), 4), "ms_160_147": round(bench.event_time_ms(s2.run, it), 4)}
if hasattr(B._lib.load(), "ssr_resample_poly_mfma") and not os.environ.get("NO_MFMA"):
    e1, e2 = s1.run().clone(), s2.run().clone()
    m1 = B.ResampleBatch(B.Ragged.from_uniform(x), 44100, 16000, exact=False)
    m2 = B.ResampleBatch(s1.out_ragged(), 48000, 44100, exact=False)          # (same input as the exact second stage)
    res["mfma_ms_441_160"] = round(bench.event_time_ms(m1.run, it), 4)
    res["mfma_ms_160_147"] = round(bench.event_time_ms(m2.run, it), 4)
    res["mfma_max_abs_diff"] = [float((m1.run() - e1).abs().max()), float((m2.run() - e2).abs().max())]
    res["signal_max_abs"] = [float(e1.abs().max()), float(e2.abs().max())]
print(json.dumps(res))
