"""Developer experiment: ssr_pair_metrics_multi alone (7 keys x 1024 targets of 4 s, plan 2048 / 512), HIP-event time per call.
With a -DSSR_DEV_KNOBS build (SSR_DEV_LIB) SSR_SPEC_KG selects the keys per wave of k_specred_wave."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B


def main():
    dev = torch.device("cuda", 0)
    n, K = int(os.environ.get("N_ITEMS", "1024")), int(os.environ.get("N_KEYS", "7"))
    g = torch.Generator(device=dev).manual_seed(1)
    tgt = (0.1 * torch.randn((n, bench.N_SAMPLES), generator=g, device=dev)).contiguous()
    est = (tgt[None] + 0.01 * torch.randn((K, n, bench.N_SAMPLES), generator=g, device=dev)).contiguous()
    plan = B.get_plan(2048, 512, "f64", dev)
    b = B.MultiPairBatch(plan, B.Ragged.from_uniform(est.reshape(K * n, -1)), B.Ragged.from_uniform(tgt), K)
    for _ in range(3): b.run(B.M_ALL)
    torch.cuda.synchronize()
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): b.run(B.M_ALL)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"kg": os.environ.get("SSR_SPEC_KG", "default"), "n": n, "keys": K, "ms_per_call": round(e0.elapsed_time(e1) / reps, 4)}))


main()
