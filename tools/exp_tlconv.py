"""Developer tool: the SSR_LOWPASS_CONV engine against oracle/tl_chain.c (bit-exact expected) and its time per launch."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build  # noqa: E402,E702
from ssr_eval_amd import backend as B, _lib  # noqa: E402
from oracle import tl_chain, stft as ostft  # noqa: E402


def lib_weights(n_fft):
    lib = _lib.load()
    F = n_fft // 2 + 1
    a, b = np.empty((n_fft, F), np.float32), np.empty((n_fft, F), np.float32)
    c, d = np.empty((n_fft, n_fft), np.float32), np.empty((n_fft, n_fft), np.float32)
    w2 = np.empty(n_fft, np.float32)
    _lib.check(lib.ssr_tl_weights(n_fft, *[x.ctypes.data_as(C.c_void_p) for x in (a, b, c, d, w2)]))
    # oracle layout: fwd [F, n_fft], inv [n_fft(sample), n_fft(channel)]
    return (np.ascontiguousarray(a.T), np.ascontiguousarray(b.T), np.ascontiguousarray(c.T), np.ascontiguousarray(d.T)), w2


def main():
    n_fft, hop = 2048, 441
    wts, w2 = lib_weights(n_fft)
    ref = ostft.tl_weights(n_fft)
    for name, x, y in zip(("fwd_re", "fwd_im", "inv_re", "inv_im"), wts, ref):
        print("weights", name, "identical", float((x == y).mean()), "max diff", float(np.abs(x - y).max()))
    plan = B.get_plan(n_fft, hop, "f64", lowpass_engine="conv")
    rng = np.random.default_rng(5)
    lens = [30000, 12345, 48000, 1500, 28223]
    cuts = [85, 683, 256, 1025, 42]
    sigs = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]
    ys = B.fft_lowpass(plan, sigs, cuts)
    torch.cuda.synchronize()
    for x, c, y in zip(sigs, cuts, ys):
        want = tl_chain.stft_hard_lowpass(x, c, n_fft, hop, weights=wts)
        got = y.cpu().numpy()
        print("lowpass n=%d cut=%d: max |diff| %.3e, mismatching samples %d / %d, max |y| %.3f" % (
            len(x), c, np.abs(got - want).max(), int((got != want).sum()), len(x), np.abs(want).max()))
    # stft complex
    re, im = B.stft(plan, sigs[:2], kind="complex", torch_style_pad=True)
    for x, r, i in zip(sigs[:2], re, im):
        wr, wi = tl_chain.stft(x, n_fft, hop, weights=wts)
        print("stft: mismatches re %d im %d of %d" % (int((r.cpu().numpy() != wr).sum()), int((i.cpu().numpy() != wi).sum()), wr.size))
    # istft of given spectra
    wr, wi = tl_chain.stft(sigs[0], n_fft, hop, weights=wts)
    y = B.istft(plan, [torch.from_numpy(wr)], [torch.from_numpy(wi)], [lens[0]])[0].cpu().numpy()
    want = tl_chain.istft(wr, wi, lens[0], n_fft, hop, weights=wts)
    print("istft: mismatches %d of %d, max diff %.3e, round trip err %.3e" % (int((y != want).sum()), len(y), np.abs(y - want).max(), np.abs(y - sigs[0]).max()))
    # timing: 256 x 4 s @ 48 kHz per cut
    n_utt = int(os.environ.get("UTT", 256))
    g = torch.Generator(device="cuda").manual_seed(1)
    data = 0.1 * torch.randn(n_utt, 192000, device="cuda", generator=g)
    wavs = [data[i] for i in range(n_utt)]
    for cut in (42, 85, 170, 256, 341, 512, 683, 1025):
        B.fft_lowpass(plan, wavs, [cut] * n_utt)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(3):
            B.fft_lowpass(plan, wavs, [cut] * n_utt)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 3
        T = 1 + 192000 // hop
        K = cut + min(cut - 1, 1023)
        flops = n_utt * T * (2.0 * 2048 * 2 * cut + 2.0 * 2048 * 2 * K)
        print("cut %4d: %.2f ms per %d utterances, %.1f TFLOP/s useful" % (cut, dt * 1e3, n_utt, flops / dt / 1e12))
    p64 = B.get_plan(n_fft, hop, "f64")
    B.fft_lowpass(p64, wavs, [256] * n_utt)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(3):
        B.fft_lowpass(p64, wavs, [256] * n_utt)
    torch.cuda.synchronize()
    print("float64 FFT engine: %.2f ms" % ((time.time() - t) / 3 * 1e3))


if __name__ == "__main__":
    main()
