"""Developer experiment: pair-metric throughput at every AudioMetrics(rate) size (ssr_eval/metrics.py:16-19), 4 s signals."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B

n = int(os.environ.get("PAIRS", "512"))
dev = torch.device("cuda", 0)
for rate in (16000, 24000, 32000, 44100, 48000):
    hop, n_fft = int(rate / 100), int(2048 / (44100 / rate))
    g = torch.Generator(device=dev).manual_seed(rate)
    tgt = (0.1 * torch.randn((n, 4 * rate), generator=g, device=dev)).contiguous()
    est = (tgt + 0.01 * torch.randn((n, 4 * rate), generator=g, device=dev)).contiguous()
    plan = B.get_plan(n_fft, hop, "f64", dev)
    b = B.PairBatch(plan, B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
    stft = bench.event_time_ms(lambda: b.run(B.M_ALL, stages=1), 3)
    full = bench.event_time_ms(lambda: b.run(B.M_ALL), 3)
    print(json.dumps({"rate": rate, "n_fft": n_fft, "hop": hop, "pairs": n, "stft_ms": round(stft, 3), "full_ms": round(full, 3),
                      "pairs_per_s": round(n / full * 1e3, 1)}), flush=True)
