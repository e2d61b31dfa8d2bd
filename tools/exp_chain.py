"""Developer experiment: the fused resampling chain (ssr_resample_poly_chain: 16 kHz -> 44.1 kHz -> 48 kHz) alone, HIP-event time per
launch for N_ITEMS utterances of 64,000 samples; FUSED=0 times the two ssr_resample_poly launches instead."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
from ssr_eval_amd import backend as B


def main():
    dev = torch.device("cuda", 0)
    n = int(os.environ.get("N_ITEMS", "4096"))
    fused = os.environ.get("FUSED", "1") == "1"
    g = torch.Generator(device=dev).manual_seed(1)
    x = (0.1 * torch.randn((n, 64000), generator=g, device=dev)).contiguous()
    b = B.ResampleChainBatch(B.Ragged.from_uniform(x), 16000, 44100, 48000, fused=None if fused else False)
    for _ in range(3): b.run()
    torch.cuda.synchronize()
    reps = int(os.environ.get("REPS", "10"))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): b.run()
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"n": n, "fused": bool(b.ran_fused), "ms_per_chain": round(e0.elapsed_time(e1) / reps, 4)}))


main()
