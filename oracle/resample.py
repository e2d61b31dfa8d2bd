"""Oracle (test infrastructure): polyphase resampling as the reference uses it.

librosa.resample(y, orig_sr, target_sr, res_type="polyphase") (ssr_eval/eval.py:145-150) reduces the
rates by their gcd and calls scipy.signal.resample_poly; ssr_eval/lowpass.py:137-140 calls
resample_poly directly.  SciPy is installed here and on the GPU box, so ``resample_poly`` below IS
the third-party routine (the pin); ``poly_plan``/``resample_poly_restated`` restate its integer
plan and arithmetic (scipy/signal/_signaltools.py resample_poly, _upfirdn_apply.pyx) so the HIP
kernel's host-side plan can be checked field by field.
"""
import math

import numpy as np
from scipy import signal


def resample_poly(x, up, down):
    return signal.resample_poly(x, up, down)


def librosa_resample_polyphase(y, orig_sr, target_sr):
    g = math.gcd(int(orig_sr), int(target_sr))
    out = signal.resample_poly(y, int(target_sr) // g, int(orig_sr) // g, axis=-1)
    return np.asarray(out, dtype=y.dtype)


def _output_len(len_h, in_len, up, down):
    return (((in_len - 1) * up + len_h) - 1) // down + 1


def poly_plan(n_in, up, down, dtype=np.float32):
    """Integer plan + taps of scipy.signal.resample_poly(x[n_in], up, down) for a float input."""
    g = math.gcd(up, down)
    up //= g
    down //= g
    n_out = n_in * up
    n_out = n_out // down + bool(n_out % down)
    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = signal.firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)).astype(dtype)
    h *= up
    n_pre_pad = down - half_len % down
    n_post_pad = 0
    n_pre_remove = (half_len + n_pre_pad) // down
    while _output_len(len(h) + n_pre_pad + n_post_pad, n_in, up, down) < n_out + n_pre_remove:
        n_post_pad += 1
    h_full = np.concatenate((np.zeros(n_pre_pad, dtype), h, np.zeros(n_post_pad, dtype)))
    return dict(up=up, down=down, n_out=n_out, half_len=half_len, n_pre_pad=n_pre_pad,
                n_post_pad=n_post_pad, n_pre_remove=n_pre_remove, h=h, h_full=h_full)


def resample_poly_restated(x, up, down):
    """Sequential float32 multiply-then-add in ascending input order, as _upfirdn_apply does."""
    x = np.asarray(x)
    p = poly_plan(x.shape[0], up, down, x.dtype)
    up, down = p["up"], p["down"]
    if up == down == 1:
        return x.copy()
    h = p["h_full"]
    hpp = -(-len(h) // up)
    hp = np.zeros(hpp * up, h.dtype)
    hp[:len(h)] = h
    out = np.zeros(p["n_out"], x.dtype)
    for m in range(p["n_out"]):
        t = (m + p["n_pre_remove"]) * down
        q, ph = divmod(t, up)
        acc = x.dtype.type(0)
        for i in range(hpp - 1, -1, -1):
            j = q - i
            if 0 <= j < x.shape[0]:
                acc = x.dtype.type(acc + x.dtype.type(x[j] * hp[ph + i * up]))
        out[m] = acc
    return out
