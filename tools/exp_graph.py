"""Developer experiment: the cfg-2 step (pair metrics + the rank's sums) replayed from a captured HIP graph against plain launches."""
import os, sys, json, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ssr_eval_amd import backend as B

dev = torch.device("cuda", 0)
n = 1024
est, tgt = bench.make_inputs(n, dev, 1)
b = B.PairBatch(B.get_plan(2048, 512, "f64", dev), B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
agg = torch.zeros(3, dtype=torch.float64, device=dev)
mask = B.M_LSD | B.M_SSIM

def step():
    out = b.run(mask)
    torch.sum(out[:, 0::3], dim=0, out=agg[:2])

def timed(fn, k=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3

res = {"plain_ms": round(timed(step), 4)}
ref = agg.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
res["graph_ms"] = round(timed(g.replay), 4)
res["same_values"] = bool(torch.equal(agg, ref))
print(json.dumps(res))
