"""Developer experiment: k_ssim alone (1024 pairs of 376 x 1025 magnitudes) for the library builds in LIBS (A/B, ablations)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
if os.environ.get("_ONE"):
    import torch, bench
    from ssr_eval_amd import backend as B
    dev = torch.device("cuda", 0)
    est, tgt = bench.make_inputs(1024, dev, 1)
    b = B.PairBatch(B.get_plan(2048, 512, "f64", dev), B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
    b.run(B.M_ALL)
    out = b.run(B.M_SSIM | B.M_LSD)
    vals = out.double().cpu().numpy() if hasattr(out, "cpu") else out
    import hashlib, numpy as np
    print(json.dumps({"lib": os.path.basename(os.environ.get("SSR_DEV_LIB", "default")),
                      "ssim_ms": round(bench.event_time_ms(lambda: b.run(B.M_SSIM | B.M_LSD, stages=2), 10), 4),
                      "step_ms": round(bench.event_time_ms(lambda: b.run(B.M_SSIM | B.M_LSD), 10), 4),
                      "values_sha": hashlib.sha1(np.ascontiguousarray(vals).tobytes()).hexdigest()[:12]}), flush=True)
else:
    for lib in os.environ.get("LIBS", "").split(","):
        env = dict(os.environ, _ONE="1")
        if lib:
            env["SSR_DEV_LIB"] = os.path.join(ROOT, lib)
        subprocess.call([sys.executable, os.path.abspath(__file__)], env=env)
