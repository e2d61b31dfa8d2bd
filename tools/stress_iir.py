"""Developer tool (GPU): randomized differential test of ssr_sosfiltfilt / ssr_sosfiltfilt_multi against scipy.signal.sosfiltfilt -
every output sample must be equal.  Random filter families / orders (1 to 16 sections: every group width of ssr_iir.h), random batch
sizes and lengths from padlen + 1 (shorter than a 16-sample chunk) to tens of thousands, float32 and float64 signals, multi-design
launches with mixed section counts.  ROUNDS=... SEED=..."""
import os
import sys

import numpy as np
import torch
from scipy import signal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssr_eval_amd import backend as B  # noqa: E402


def design(rng):
    ft = rng.choice(["butter", "cheby1", "ellip", "bessel"])
    order = int(rng.integers(1, 33))
    wn = float(rng.uniform(0.02, 0.9))
    if ft == "butter":
        return signal.butter(order, wn, output="sos")
    if ft == "cheby1":
        return signal.cheby1(min(order, 20), 0.5, wn, output="sos")
    if ft == "ellip":
        return signal.ellip(min(order, 14), 0.5, 60.0, wn, output="sos")
    return signal.bessel(min(order, 24), wn, output="sos")


def main():
    rng = np.random.default_rng(int(os.environ.get("SEED", 7)))
    rounds = int(os.environ.get("ROUNDS", 60))
    bad = checked = 0
    widths = set()
    for r in range(rounds):
        sos = design(rng)
        S = sos.shape[0]
        if S > 16:
            continue
        widths.add(1 if S <= 1 else 2 if S <= 2 else 4 if S <= 4 else 8 if S <= 8 else 16)
        multi = [d for d in (design(rng) for _ in range(int(rng.integers(2, 9)))) if d.shape[0] <= 8] if r % 3 == 0 else []
        pad_of = lambda d: 3 * (2 * d.shape[0] + 1 - min((d[:, 2] == 0).sum(), (d[:, 5] == 0).sum()))   # noqa: E731  (SciPy's default padlen)
        padlen = max(pad_of(d) for d in [sos] + multi)
        n_sig = int(rng.choice([1, 2, 3, 17, 64, 65, 130, 400]))
        hi = int(rng.choice([padlen + 40, 600, 5000, 40000]))
        lens = rng.integers(padlen + 1, max(hi, padlen + 2), n_sig)
        f64 = bool(rng.integers(0, 4) == 0)
        sigs = [rng.standard_normal(int(n)).astype(np.float64 if f64 else np.float32) for n in lens]
        got = B.sosfiltfilt(sos, sigs)
        for s_, g in zip(sigs, got):
            bad += int((g.cpu().numpy() != signal.sosfiltfilt(sos, s_)).sum())
            checked += len(s_)
        if multi and not f64:                          # a multi-design launch with mixed section counts over the same batch
            outs = B.sosfiltfilt_multi(multi, sigs)
            for d, per in zip(multi, outs):
                for s_, g in zip(sigs, per):
                    bad += int((g.cpu().numpy() != signal.sosfiltfilt(d, s_)).sum())
                    checked += len(s_)
    print("stress_iir: %d rounds, group widths %s, %d samples checked, %d differ from SciPy" % (rounds, sorted(widths), checked, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
