"""Developer tool: the SSR_LOWPASS_CONV engine against oracle/tl_chain.c and torch-CPU's own conv1d (bit-exact expected) and its
time per launch.  PARITY=0 skips the checks; UTT = utterances of the timing runs (default 1024 x 4 s @ 48 kHz = BASELINE cfg-3);
with a -DSSR_DEV_KNOBS build (SSR_DEV_LIB) VARIANTS="128:2,128:3,64:2" times the inverse product's tile variants."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build  # noqa: E402,E702
from ssr_eval_amd import backend as B  # noqa: E402
from oracle import tl_chain, stft as ostft  # noqa: E402

CUTS = (42, 85, 170, 256, 341, 512, 683)


def torch_conv_lowpass(x, cut):
    """stft_hard_lowpass_v0 through the published torchlibrosa arithmetic on torch-CPU (oracle restatement of the modules)."""
    re, im = ostft.tl_stft_conv(x[None])
    mag = np.clip(re ** 2 + im ** 2, np.float32(1e-8), np.inf) ** np.float32(0.5)
    c, s_ = re / mag, im / mag
    mag[..., cut:] = 0
    return ostft.tl_istft_conv(mag * c, mag * s_, len(x))[0]


def report(what, got, want):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    print("%-46s mismatching %d / %d, max |diff| %.3e" % (what, int((got != want).sum()), want.size, float(np.abs(got - want).max())), flush=True)


def parity(plan):
    n_fft, hop = plan.n_fft, plan.hop
    rng = np.random.default_rng(5)
    lens = [30000, 12345, 48000, 1500, 28223, 26001]
    cuts = [85, 683, 256, 1025, 42, 557]
    sigs = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]
    ys = B.fft_lowpass(plan, sigs, cuts)                       # per-item cuts
    for x, c, y in zip(sigs, cuts, ys):
        report("per-item cut n=%d cut=%d vs tl_chain" % (len(x), c), y, tl_chain.stft_hard_lowpass(x, c, n_fft, hop))
    ys = B.fft_lowpass(plan, sigs, [341] * len(sigs))          # one cut: row tiles across items
    for x, y in zip(sigs, ys):
        report("uniform cut 341 n=%d vs tl_chain" % len(x), y, tl_chain.stft_hard_lowpass(x, 341, n_fft, hop))
    yk = B.fft_lowpass_multi(plan, sigs, CUTS)
    for c, ys in zip(CUTS, yk):
        bad = sum(int((y.cpu().numpy() != tl_chain.stft_hard_lowpass(x, c, n_fft, hop)).sum()) for x, y in zip(sigs, ys))
        print("multi cut %4d vs tl_chain: %d mismatching samples over %d signals" % (c, bad, len(sigs)), flush=True)
    torch.set_num_threads(8)
    for x, c in ((sigs[0], 85), (sigs[2], 683), (sigs[5], 557)):
        y = B.fft_lowpass(plan, [x], [c])[0]
        report("n=%d cut=%d vs TORCH conv1d (8 threads)" % (len(x), c), y, torch_conv_lowpass(x, c))
    re, im = B.stft(plan, sigs[:3], kind="complex", torch_style_pad=True)
    for x, r, i in zip(sigs[:3], re, im):
        wr, wi = tl_chain.stft(x, n_fft, hop)
        report("stft complex n=%d re" % len(x), r, wr)
        report("stft complex n=%d im" % len(x), i, wi)
    wr, wi = tl_chain.stft(sigs[0], n_fft, hop)
    y = B.istft(plan, [torch.from_numpy(wr)], [torch.from_numpy(wi)], [lens[0]])[0]
    report("istft", y, tl_chain.istft(wr, wi, lens[0], n_fft, hop))
    print("istft round trip err %.3e" % float(np.abs(y.cpu().numpy() - sigs[0]).max()))


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / reps


def main():
    n_fft, hop = 2048, 441
    plan = B.get_plan(n_fft, hop, "f64", lowpass_engine="conv")
    if os.environ.get("PARITY", "1") != "0":
        parity(plan)
    n_utt = int(os.environ.get("UTT", 1024))
    g = torch.Generator(device="cuda").manual_seed(1)
    data = 0.1 * torch.randn(n_utt, 192000, device="cuda", generator=g)
    r = B.Ragged.from_uniform(data)
    T = 1 + 192000 // hop
    variants = [v.split(":") for v in os.environ.get("VARIANTS", "").split(",") if v] or [(None, None)]
    for bm, ns in variants:
        if bm is not None:
            os.environ["SSR_TL_BM"], os.environ["SSR_TL_NS"] = bm, ns
        print("== inverse tile rows %s, stages %s" % (bm or "default", ns or "default"), flush=True)
        for cut in [int(v) for v in os.environ.get("CUTS_TIMED", "42,85,170,256,341,512,683,1025").split(",") if v]:
            b = B.LowpassBatch(plan, r, [cut] * n_utt)
            dt = timeit(b.run)
            K = cut + min(cut - 1, 1023)
            flops = n_utt * T * (2.0 * 2048 * 2 * cut + 2.0 * 2048 * 2 * K)
            print("cut %4d: %7.2f ms per %d utterances, %.1f TFLOP/s useful" % (cut, dt * 1e3, n_utt, flops / dt / 1e12), flush=True)
            del b
        if os.environ.get("MULTI", "1") == "0":
            continue
        m = B.MultiLowpassBatch(plan, r, CUTS)
        dt = timeit(m.run, reps=2)
        flops = n_utt * T * (2.0 * 2048 * 2 * max(CUTS) + sum(2.0 * 2048 * 2 * (c + c - 1) for c in CUTS))
        print("multi %s: %.2f ms per %d utterances x %d keys = %.1f k pairs/s of degradation alone, %.1f TFLOP/s" % (
            CUTS, dt * 1e3, n_utt, len(CUTS), n_utt * len(CUTS) / dt / 1e3, flops / dt / 1e12), flush=True)
        del m
    if os.environ.get("MULTI", "1") != "0":
        p64 = B.get_plan(n_fft, hop, "f64")
        b = B.LowpassBatch(p64, r, [256] * n_utt)
        print("float64 FFT engine: %.2f ms" % (timeit(b.run) * 1e3))


if __name__ == "__main__":
    main()
