"""Developer experiment: time the stages of ssr_pair_metrics under different metric masks / precisions, and check the
first pairs against the oracle.  SSR_DEV_LIB selects the library build (A/B of kernel variants)."""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B


def main():
    n = int(os.environ.get("PAIRS", "1024"))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    amp = float(os.environ.get("AMP", "0.1"))          # e.g. 1e30: magnitudes overflow float32 - is any kernel's time data-dependent?
    tgt = (amp * torch.randn((n, bench.N_SAMPLES), generator=g, device=dev)).contiguous()
    est = (tgt + 0.1 * amp * torch.randn((n, bench.N_SAMPLES), generator=g, device=dev)).contiguous()
    res = {"lib": os.environ.get("SSR_DEV_LIB", "default")}
    for prec in os.environ.get("PRECS", "f64").split(","):
        plan = B.get_plan(2048, 512, prec, dev)
        b = B.PairBatch(plan, B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
        for name, mask, st in [("stft lsd+mag", B.M_LSD | B.M_SSIM, 1), ("stft lsd only", B.M_LSD, 1),
                               ("stft all4+mag", B.M_ALL, 1), ("ssim", B.M_SSIM, 2), ("full lsd+ssim", B.M_LSD | B.M_SSIM, 7)]:
            res["%s %s" % (prec, name)] = round(bench.event_time_ms(lambda: b.run(mask, stages=st), 10), 4)
        if prec == "f64" and not os.environ.get("NO_CHECK"):
            from oracle import metrics as om
            got = b.run(B.M_ALL).cpu().numpy()
            worst = 0.0
            for i in (0, n // 2, n - 1):
                w = om.evaluation(est[i].cpu().numpy(), tgt[i].cpu().numpy(), n_fft=2048, hop=512)
                w = np.array([w[k] for k in ("lsd", "log_sispec", "sispec", "ssim")])
                worst = max(worst, float(np.abs((got[i] - w) / w).max()))
            res["max_rel_err_vs_oracle"] = worst
    print(json.dumps(res))


if __name__ == "__main__":
    main()
