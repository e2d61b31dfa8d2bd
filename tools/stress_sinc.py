"""Developer tool: randomized differential runs of ssr_resample_sinc (round 5: phase-major table, scalar quad loads, per-lane rows) against
the NumPy restatement of resampy (oracle/resampy.py), bit for bit expected: random pairs of common audio rates (also kaiser_fast), ragged
batches of 1-12 signals of 1 ... 250,000 samples, silence, an impulse and a full-scale square wave among them.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssr_eval_amd import backend as B  # noqa: E402
from oracle import resampy as orsy  # noqa: E402

RATES = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000]


def main():
    rng = np.random.default_rng(int(os.environ.get("SEED", 7)))
    res = {"launches": 0, "signals": 0, "samples_out": 0, "mismatching": 0, "pairs": []}
    for trial in range(int(os.environ.get("TRIALS", 24))):
        a, b = (int(v) for v in rng.choice(RATES, 2, replace=False))
        if trial == 0:
            a, b = 44100, 48000
        if trial == 1:
            a, b = 48000, 44100
        name = "kaiser_fast" if trial % 6 == 5 else "kaiser_best"
        n_sig = int(rng.integers(1, 13))
        lens = [int(v) for v in np.exp(rng.uniform(0, np.log(250000 if trial % 4 == 0 else 40000), n_sig))]
        sigs = [(0.3 * rng.standard_normal(n)).astype(np.float32) for n in lens]
        if n_sig > 2:
            sigs[0][:] = 0.0
            sigs[1][:] = 0.0; sigs[1][len(sigs[1]) // 2] = 1.0
            sigs[2][:] = np.where(np.arange(len(sigs[2])) % 64 < 32, 1.0, -1.0)
        got = B.resample_sinc(sigs, a, b, name)
        for x, y in zip(sigs, got):
            want = orsy.librosa_resample_kaiser(x, a, b, name)
            y = y.cpu().numpy()
            res["mismatching"] += int(y.shape != want.shape) or int((y != want).sum())
            res["samples_out"] += int(want.size)
        res["launches"] += 1
        res["signals"] += n_sig
        res["pairs"].append("%d>%d%s" % (a, b, "f" if name == "kaiser_fast" else ""))
    print(json.dumps(res), flush=True)
    return 1 if res["mismatching"] else 0


if __name__ == "__main__":
    sys.exit(main())
