"""Developer experiment: per-call latency of the reference-shaped API (serial usage, host arrays in, floats out)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
from ssr_eval_amd import AudioMetrics, lowpass

rng = np.random.default_rng(0)
t = (0.1 * rng.standard_normal(192000)).astype(np.float32)
e = (t + 0.01 * rng.standard_normal(192000)).astype(np.float32)
for name, am in (("2048/512", AudioMetrics(48000, n_fft=2048, hop_length=512)), ("2229/480 (AudioMetrics(48000))", AudioMetrics(48000))):
    am.evaluation(e, t, "")
    t0 = time.perf_counter()
    for _ in range(50):
        am.evaluation(e, t, "")
    dt = (time.perf_counter() - t0) / 50
    print("evaluation() %s: %.2f ms per call -> %.0f pairs/s serial" % (name, dt * 1e3, 1 / dt))
x = t[:176400]
lowpass(x, 6000, 44100, 1, "stft_hard")
t0 = time.perf_counter()
for _ in range(20):
    lowpass(x, 6000, 44100, 1, "stft_hard")
print("lowpass(stft_hard): %.2f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
