"""Developer experiment: time the stages of ssr_pair_metrics under different metric masks / precisions."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ssr_eval_amd import backend as B

def main():
    n = int(os.environ.get("PAIRS", "1024"))
    dev = torch.device("cuda", 0)
    est, tgt = bench.make_inputs(n, dev, 1)
    res = {}
    for prec in ("f64", "f32"):
        plan = B.get_plan(2048, 512, prec, dev)
        b = B.PairBatch(plan, B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
        for name, mask, st in [("stft lsd+mag", B.M_LSD | B.M_SSIM, 1), ("stft mag only", B.M_SSIM, 1), ("stft lsd only", B.M_LSD, 1),
                               ("stft all4+mag", B.M_ALL, 1), ("ssim", B.M_SSIM, 2), ("full lsd+ssim", B.M_LSD | B.M_SSIM, 7)]:
            res["%s %s" % (prec, name)] = round(bench.event_time_ms(lambda: b.run(mask, stages=st), 10), 4)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main()
