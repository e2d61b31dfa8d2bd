"""Developer tool: randomized differential runs of the round-5 conv low-pass engine against oracle/tl_chain.c (bit for bit expected),
larger and more irregular than the suite's cases.  Prints one JSON line; profiles/r05_stress.log keeps the output of the last GPU run.

  1. ssr_fft_lowpass_multi on ragged batches (lengths 1,100 ... 60,000; 1-40 items; 1-9 random cuts incl. 0 and n_bins): every output
     sample against tl_chain, and against the same batch through per-item-cut ssr_fft_lowpass launches (tiles that never span two items);
  2. batches large enough for the 128-row inverse tile (>= 4096 frame rows) and small ones on the 64-row tile;
  3. ssr_plan_create_ex plans (other n_fft / window / center / pad_mode) through the multi entry;
  4. ssr_istft of random spectra (the transpose-pack path) against tl_chain.istft.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssr_eval_amd import backend as B  # noqa: E402
from oracle import tl_chain, stft as ostft  # noqa: E402


def main():
    rng = np.random.default_rng(int(os.environ.get("SEED", 5)))
    res = {"multi_batches": 0, "multi_signals_x_keys": 0, "multi_mismatching": 0, "per_item_mismatching": 0, "ex_configs": 0,
           "ex_mismatching": 0, "istft_mismatching": 0, "rows_range": [10 ** 9, 0]}
    plan = B.get_plan(2048, 441, "f64", lowpass_engine="conv")
    for trial in range(int(os.environ.get("TRIALS", 10))):
        n_items = int(rng.integers(1, 41))
        big = trial % 3 == 0
        lens = [int(v) for v in rng.integers(1100, 60000 if big else 9000, n_items)]
        sigs = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]
        K = int(rng.integers(1, 10))
        cuts = [int(v) for v in rng.integers(0, 1026, K)]
        if trial == 0:
            cuts[0] = 0
        if trial == 1:
            cuts[-1] = 1025
        rows = sum(1 + n // 441 for n in lens)
        res["rows_range"] = [min(res["rows_range"][0], rows), max(res["rows_range"][1], rows)]
        ys = B.fft_lowpass_multi(plan, sigs, cuts)
        check = rng.choice(n_items, min(n_items, 5), replace=False)
        for k, c in enumerate(cuts):
            for i in check:
                want = tl_chain.stft_hard_lowpass(sigs[i], c)
                res["multi_mismatching"] += int(not np.array_equal(ys[k][i].cpu().numpy(), want))
                res["multi_signals_x_keys"] += 1
        per_item = [cuts[int(rng.integers(0, K))] for _ in range(n_items)]
        yp = B.fft_lowpass(plan, sigs, per_item)
        for i in range(n_items):
            res["per_item_mismatching"] += int(not torch.equal(yp[i], ys[cuts.index(per_item[i])][i]))
        res["multi_batches"] += 1
    for n_fft in (256, 1024):
        for window in ("hann", "hamming", ("tukey", 0.3)):
            for center, pad_mode in ((True, "reflect"), (True, "constant"), (False, "reflect")):
                hop = int(rng.integers(n_fft // 8, n_fft // 2 + 1))
                win = None if window == "hann" else ostft.window_array(window, n_fft)
                p = B.get_plan_ex(n_fft, hop, window, win, center, pad_mode)
                lens = [int(v) for v in rng.integers(n_fft + 1, 6 * n_fft, 3)]
                sigs = [(0.2 * rng.standard_normal(n)).astype(np.float32) for n in lens]
                cuts = [int(v) for v in rng.integers(1, n_fft // 2 + 2, 2)]
                ys = B.fft_lowpass_multi(p, sigs, cuts)
                for k, c in enumerate(cuts):
                    for x, y in zip(sigs, ys[k]):
                        want = tl_chain.stft_hard_lowpass(x, c, n_fft, hop, window=window, center=center, pad_mode=pad_mode)
                        res["ex_mismatching"] += int(not np.array_equal(y.cpu().numpy(), want))
                res["ex_configs"] += 1
    for n in (5000, 31000):
        x = (0.1 * rng.standard_normal(n)).astype(np.float32)
        wr, wi = tl_chain.stft(x)
        wr = (wr + 0.01 * rng.standard_normal(wr.shape)).astype(np.float32)
        y = B.istft(plan, [torch.from_numpy(wr)], [torch.from_numpy(wi)], [n])[0].cpu().numpy()
        res["istft_mismatching"] += int(not np.array_equal(y, tl_chain.istft(wr, wi, n)))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
