"""Developer experiment: time ssr_fft_lowpass (1024 x 4 s @ 48 kHz, FDomainHelper 2048/441) for the library builds listed in
LIBS (comma-separated paths, A/B of kernel variants; each runs in its own process)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build


def one():
    import torch
    import bench
    from ssr_eval_amd import backend as B
    n = int(os.environ.get("ITEMS", "1024"))
    hop = int(os.environ.get("HOP", "441"))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    x = (0.1 * torch.randn((n, bench.N_SAMPLES), generator=g, device=dev)).contiguous()
    prec = os.environ.get("PREC", "f64")
    plan = B.get_plan(2048, hop, prec, dev)
    lb = B.LowpassBatch(plan, B.Ragged.from_uniform(x), [256] * n)
    ms = bench.event_time_ms(lambda: lb.run(), 10)
    print(json.dumps({"lib": os.path.basename(os.environ.get("SSR_DEV_LIB", "default")), "prec": prec, "hop": hop, "fft_lowpass_ms": round(ms, 4)}), flush=True)


if __name__ == "__main__":
    if os.environ.get("_ONE"):
        one()
    else:
        for lib in os.environ.get("LIBS", "").split(",") or [""]:
            env = dict(os.environ, _ONE="1")
            if lib:
                env["SSR_DEV_LIB"] = os.path.join(ROOT, lib)
            subprocess.call([sys.executable, os.path.abspath(__file__)], env=env)
