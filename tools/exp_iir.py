"""Developer tool: ssr_sosfiltfilt_multi on an evaluate()-shaped batch (FILES files of 1.5-9 s at 44.1 kHz, 36 designs = 4 filter types x 3
cutoffs x orders 2 / 4 / 8): ms per launch, and every output against the single-design launch (SSR_DEV_LIB: an alternative build)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # noqa: E402,E702
import bench  # noqa: E402
from ssr_eval_amd import backend as B  # noqa: E402
from ssr_eval_amd.lowpass import _design  # noqa: E402


def main():
    rng = np.random.default_rng(3)
    n_files = int(os.environ.get("FILES", 128))
    xs = [(0.1 * rng.standard_normal(int(rng.uniform(1.5, 9.0) * 44100))).astype(np.float32) for _ in range(n_files)]
    designs = [_design(hc, 44100, order, ft) for ft in ("butter", "cheby1", "ellip", "bessel") for hc in (2000, 4000, 6000) for order in (2, 4, 8)]
    r = B.Ragged.from_list(xs)
    got = B.sosfiltfilt_multi(designs, r)
    bad = 0
    for d in (0, 7, 20, 35):
        one = B.sosfiltfilt(designs[d], r)
        bad += sum(int((a != b).sum()) for a, b in zip(one, got[d]))
    ms = bench.event_time_ms(lambda: B.sosfiltfilt_multi(designs, r), 3)
    per_g = {}
    if os.environ.get("PER_G"):        # one design per launch: the step latency of each group width (the longest utterance sets the time)
        steps = 2.0 * (max(len(x) for x in xs) + 2 * 27)
        for order in (2, 4, 8, 16, 20):
            d = _design(4000, 44100, order, "butter")
            t = bench.event_time_ms(lambda: B.sosfiltfilt(d, r), 3)
            per_g["sections_%d" % d.shape[0]] = {"ms": round(t, 2), "ns_per_step": round(1e6 * t / steps, 1)}
    print(json.dumps({"lib": os.environ.get("SSR_DEV_LIB", ""), "files": n_files, "designs": len(designs), "ms_per_launch": round(ms, 2),
                      "samples_differing_from_single_design_launches": bad, "per_group": per_g}), flush=True)


if __name__ == "__main__":
    main()
