"""Developer experiment: cfg-3's launch sequence (low-pass -> STFT + LSD + SISpec -> SSIM -> finalize, per cutoff) with HIP events
BETWEEN the stages, i.e. each stage timed IN the sequence rather than alone in a loop - does the kind of low-pass kernel change
what the kernels after it cost (clock / power management state)?  SSR_NO_FUSED_OLA=1 (with a -DSSR_DEV_KNOBS build) selects the
round-2 low-pass."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B


def main():
    dev = torch.device("cuda", 0)
    n = 1024
    g = torch.Generator(device=dev).manual_seed(1)
    tgt = (0.1 * torch.randn((n, bench.N_SAMPLES), generator=g, device=dev)).contiguous()
    tr = B.Ragged.from_uniform(tgt)
    lp = B.LowpassBatch(B.get_plan(2048, 441, "f64", dev), tr, [256] * n)
    batch = B.PairBatch(B.get_plan(2048, 512, "f64", dev), lp.out_ragged(), tr)
    cuts = list(bench.CUT_BINS)
    BIGLDS = int(os.environ.get("BIG_LDS", "0"))     # a tiny launch of a kernel with 139 KB of LDS per workgroup after every low-pass
    if BIGLDS:
        small = (0.1 * torch.randn((2, 30000), generator=g, device=dev)).contiguous()
        big = B.PairBatch(B.get_plan(3001, 750, "f64", dev), B.Ragged.from_uniform(small), B.Ragged.from_uniform(small.clone()))
        big.run(B.M_LSD)
    COOL = int(os.environ.get("COOL_MB", "0"))
    if COOL:
        scratch_a = torch.zeros(COOL * 1024 * 1024 // 4, dtype=torch.float32, device=dev); scratch_b = torch.empty_like(scratch_a)
    def seq(evs=None):
        for c in cuts:
            lp.set_cuts(c)
            if evs is not None: evs.append(torch.cuda.Event(enable_timing=True)); evs[-1].record()
            lp.run()
            if BIGLDS:
                big.run(B.M_LSD)
            if COOL:                      # a memory-bound "break" after the low-pass (what k_ola_paired was): 2.3 GB read + written
                scratch_b.copy_(scratch_a, non_blocking=True)
            if evs is not None: evs.append(torch.cuda.Event(enable_timing=True)); evs[-1].record()
            batch.run(B.M_ALL, stages=1)
            if evs is not None: evs.append(torch.cuda.Event(enable_timing=True)); evs[-1].record()
            batch.run(B.M_ALL, stages=6)
        if evs is not None: evs.append(torch.cuda.Event(enable_timing=True)); evs[-1].record()
    for _ in range(3): seq()
    torch.cuda.synchronize()
    acc = [0.0, 0.0, 0.0]; reps = 8
    import time
    t0 = time.perf_counter()
    all_evs = []
    for _ in range(reps):
        evs = []; seq(evs); all_evs.append(evs)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    for evs in all_evs:
        for k in range(len(cuts)):
            for s in range(3):
                acc[s] += evs[3 * k + s].elapsed_time(evs[3 * k + s + 1])
    m = reps * len(cuts)
    print(json.dumps({"no_fused_ola": os.environ.get("SSR_NO_FUSED_OLA", "0"), "cool_MB": COOL, "big_lds_probe": BIGLDS, "rpc": os.environ.get("SSR_LG_RPC", "auto"), "in_sequence_ms": {"fft_lowpass": round(acc[0] / m, 4), "stft+lsd+sispec": round(acc[1] / m, 4), "ssim+finalize": round(acc[2] / m, 4)},
                      "ms_per_step_of_7_cutoffs": round(wall * 1e3, 3)}))


if __name__ == "__main__":
    main()
