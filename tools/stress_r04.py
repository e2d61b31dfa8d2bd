"""Developer tool: randomized differential runs of the round-4 additions against their oracles (SciPy / oracle/tl_chain.c), larger
than what the -m gpu suite affords: the fused resampling chain on ragged batches, ssr_plan_create_ex on random window / padding /
size combinations, ssr_pair_metrics_multi against K plain calls."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from scipy import signal
from ssr_eval_amd import backend as B, _lib
from oracle import tl_chain, stft as ostft
import ctypes as C

rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
res = {}
# 1. fused chain, ragged
bad = 0
for rep in range(4):
    lens = rng.integers(1, 30000, 300)
    sig = [(0.1 * rng.standard_normal(int(n))).astype(np.float32) for n in lens]
    for sr_in, up1, down1 in ((16000, 441, 160), (8000, 441, 80)):
        b = B.ResampleChainBatch(B.Ragged.from_list(sig), sr_in, 44100, 48000, fused=True)
        out = b.run().cpu().numpy()
        for i in rng.choice(len(sig), 40, replace=False):
            w = signal.resample_poly(signal.resample_poly(sig[i], up1, down1), 160, 147)
            g = out[b.out_off[i]:b.out_off[i] + b.out_len[i]]
            bad += int(g.shape != w.shape or not np.array_equal(g, w))
res["chain_mismatching_items"] = bad
# 2. ex plans
def wts_of(n_fft, win):
    F = n_fft // 2 + 1
    a, b_ = np.empty((n_fft, F), np.float32), np.empty((n_fft, F), np.float32)
    c, d = np.empty((n_fft, n_fft), np.float32), np.empty((n_fft, n_fft), np.float32)
    w = None if win is None else np.ascontiguousarray(win, np.float64)
    _lib.check(_lib.load().ssr_tl_weights_ex(n_fft, None if w is None else w.ctypes.data, *[v.ctypes.data_as(C.c_void_p) for v in (a, b_, c, d)], None))
    return tuple(np.ascontiguousarray(v.T) for v in (a, b_, c, d))
bad = 0; n_cfg = 0
for n_fft in (256, 1024, 2048):
    for window in ("hann", "hamming", "blackmanharris", ("tukey", 0.3)):
        for center, pad_mode in ((True, "reflect"), (True, "constant"), (False, "reflect")):
            hop = int(rng.integers(n_fft // 8, n_fft // 2 + 1))
            win = None if window == "hann" else ostft.window_array(window, n_fft)
            wts = None          # (round 5: the plan multiplies by torchlibrosa's own numpy-built tables = the oracle's default)
            plan = B.get_plan_ex(n_fft, hop, window, win, center, pad_mode)
            lens = [int(v) for v in rng.integers(n_fft + 1, 6 * n_fft, 3)]
            sigs = [(0.2 * rng.standard_normal(n)).astype(np.float32) for n in lens]
            cuts = [int(v) for v in rng.integers(1, n_fft // 2 + 2, 3)]
            ys = B.fft_lowpass(plan, sigs, cuts)
            for x, c_, y in zip(sigs, cuts, ys):
                want = tl_chain.stft_hard_lowpass(x, c_, n_fft, hop, weights=wts, window=window, center=center, pad_mode=pad_mode)
                bad += int(not np.array_equal(y.cpu().numpy(), want))
            n_cfg += 1
res["ex_plan_configs"] = n_cfg; res["ex_plan_mismatching_signals"] = bad
# 3. multi vs K calls
worst = 0.0
for n_fft, hop in ((2048, 512), (2229, 480), (1114, 240)):
    plan = B.get_plan(n_fft, hop, "f64")
    n, K = 9, int(rng.integers(2, 8))
    lens = [int(v) for v in rng.integers(8 * hop + 100, 40000, n)]
    tgt = [(0.1 * rng.standard_normal(m)).astype(np.float32) for m in lens]
    ests = [[(t + 0.02 * (k + 1) * rng.standard_normal(len(t))).astype(np.float32) for t in tgt] for k in range(K)]
    got = B.pair_metrics_multi(plan, ests, tgt)
    for k in range(K):
        ref = B.pair_metrics(plan, ests[k], tgt)
        worst = max(worst, float(np.max(np.abs(got[:, k] - ref) / np.maximum(np.abs(ref), 1e-3))))
res["multi_vs_k_calls_worst_rel"] = worst
print(json.dumps(res))
