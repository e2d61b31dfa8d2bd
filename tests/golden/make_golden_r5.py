#!/usr/bin/env python
"""Round-5 additions to the golden vectors -> tests/golden/reference_vectors_r5.npz (build container only).

Same rules as make_golden.py (which this imports for its stubs + the reference import): /root/reference/ssr_eval is imported
as-is and executed; the file written holds data only (seeds of the inputs + the reference's outputs).

Why: round 5 established that the HIP conv engine can reproduce torch-CPU's conv1d BIT FOR BIT for signals of >= 55 frames (the
earlier lp_* / c3_* vectors are 9,000 / 14,400 samples = 21 / 33 frames, where torch runs its strided forward convolution in a
different summation order).  These vectors are long enough:

  lp5_*   ssr_eval.lowpass.lowpass(x, highcut, fs, _type="stft_hard") (ssr_eval/lowpass.py:156-196 -> stft_hard_lowpass_v0, :17-28) on
          0.1 N(0,1) noise of 26,000 samples (59 frames) and on a speech-like 44.1 kHz signal of 1 s, three cutoffs each
  c35_*   BASELINE cfg-3 in small: SSR_Eval_Helper.lowpass_stft_hard (ssr_eval/eval.py:401-410) on a 48 kHz target of 30,000
          samples for the sweep {2, 4, 8, 12, 16, 24, 32} kHz, with AudioMetrics(48000, 2048, 512).evaluation of every key
The inputs are regenerated from their seeds by the tests (numpy Generator streams are stable); tables_sha256 pins the float32
weight tables torchlibrosa's numpy expressions gave on the generating host.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the stubs, imports the reference)

from oracle import stft as ostft  # noqa: E402


def noise(seed, n):
    return (0.1 * np.random.default_rng(seed).standard_normal(n)).astype(np.float32)


def main():
    torch.set_num_threads(8)
    out = {}
    ref_lowpass = G.ref_lowpass
    ref_lowpass.f_helper = None
    lp_cases = [("noise", 501, 26000, [(4000, 44100), (12000, 44100), (6000, 48000)]),
                ("speech", 502, 44100, [(2000, 44100), (8000, 44100), (16000, 44100)])]
    for name, seed, n, cuts in lp_cases:
        x = noise(seed, n) if name == "noise" else G.speechlike(seed, n, 44100)
        if name != "noise":
            out["lp5_%s_x" % name] = x               # (the noise inputs are regenerated from their seeds by the tests)
        for hc, fs in cuts:
            y = ref_lowpass.lowpass(x, hc, fs, order=1, _type="stft_hard")
            out["lp5_%s_%d_%d" % (name, hc, fs)] = np.asarray(y, np.float32)
    out["lp5_cases"] = np.array(json.dumps([[c[0], c[1], c[2], c[3]] for c in lp_cases]))
    # cfg-3 in small
    x = noise(503, 30000)
    helper = G.SSR_Eval_Helper(G.BasicTestee(), input_sr=48000, output_sr=48000, evaluation_sr=48000,
                               setting_fft={"cutoff_freq": [2000, 4000, 8000, 12000, 16000, 24000, 32000]})
    degraded = helper.lowpass_stft_hard("x.wav", x, 48000)
    am = G.AudioMetrics(48000)
    am.hop_length, am.n_fft = 512, 2048
    keys, mets = [], []
    for k, y in degraded.items():
        keys.append(k)
        out["c35_y_" + k] = np.asarray(y, np.float32)
        r = am.evaluation(np.asarray(y, np.float32), x, "")
        mets.append([r["lsd"], r["log_sispec"], r["sispec"], r["ssim"]])
    out["c35_keys"] = np.array(keys)
    out["c35_metrics_2048_512"] = np.array(mets, np.float64)
    out["c35_seed_n"] = np.array([503, 30000])
    h = hashlib.sha256()
    for t in ostft.tl_weights(2048):
        h.update(np.ascontiguousarray(t).tobytes())
    out["tables_sha256"] = np.array(h.hexdigest())
    out["torch_threads"] = np.array(torch.get_num_threads())
    path = os.path.join(HERE, "reference_vectors_r5.npz")
    np.savez_compressed(path, **out)
    print("wrote %d arrays, %.1f KB" % (len(out), os.path.getsize(path) / 1024), keys, out["c35_metrics_2048_512"][:, 0])


if __name__ == "__main__":
    main()
