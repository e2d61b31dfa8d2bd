"""Oracle (test infrastructure): resampy.resample(x, sr_orig, sr_new, filter="kaiser_best" / "kaiser_fast") and the
librosa.load / librosa.resample(res_type="kaiser_best") wrapper around it, restated in NumPy.

resampy is a third-party dependency of librosa that is absent from /root/reference and from the image, so
this restates its published algorithm (resampy 0.2.x: filters.sinc_window + interpn.resample_f, after J.O. Smith's
"Digital Audio Resampling" bandlimited interpolation) and is pinned only by closed-form / cross-implementation checks in
tests/test_oracle.py - PARITY WITH THE PACKAGE ITSELF IS UNPINNED:
  * the stored kaiser_best table of the package (data/kaiser_best.npz) is regenerated here from its documented parameters
    (64 zero crossings, 2^9 table samples per crossing, Kaiser beta 14.769656459379492, roll-off 0.9475937167399596);
  * resampy 0.2.x accumulates the time register (t += 1/ratio) and adds every tap into the float32 output array; both
    are reproduced (sequential float64 cumsum, float32 rounding after every tap).
"""
import numpy as np
from scipy.signal.windows import kaiser

FILTERS = {  # name -> (num_zeros, precision, beta, rolloff)   (resampy/filters.py docstrings)
    "kaiser_best": (64, 9, 14.769656459379492, 0.9475937167399596),
    "kaiser_fast": (16, 9, 8.555504641634386, 0.85),
}


def sinc_window(num_zeros, precision, beta, rolloff):
    """resampy.filters.sinc_window with window = scipy.signal.kaiser(beta): the right half of the interpolation filter."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def filter_tables(ratio, name="kaiser_best"):
    """(interp_win, interp_delta, num_table, index_step, scale) as resampy.core.resample prepares them."""
    nz, prec, beta, roll = FILTERS[name]
    win, num_table = sinc_window(nz, prec, beta, roll)
    win = win.copy()
    if ratio < 1:
        win *= ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    return win, delta, num_table, int(scale * num_table), scale


def time_register(n_out, ratio):
    """t_0 = 0, t_k = fl(t_{k-1} + 1/ratio): resampy 0.2.x's running time register (np.cumsum adds sequentially)."""
    if n_out <= 0:
        return np.zeros(0)
    tr = np.empty(n_out)
    tr[0] = 0.0
    if n_out > 1:
        tr[1:] = np.cumsum(np.full(n_out - 1, 1.0 / ratio))
    return tr


def resample(x, sr_orig, sr_new, name="kaiser_best"):
    """resampy.resample for a 1-D float32 signal: [n] -> [int(n * ratio)] float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    ratio = float(sr_new) / sr_orig
    n_in = x.shape[0]
    n_out = int(n_in * ratio)
    win, delta, num_table, step, scale = filter_tables(ratio, name)
    nwin = win.shape[0]
    tr = time_register(n_out, ratio)
    n = tr.astype(np.int64)
    y = np.zeros(n_out, np.float32)
    frac = scale * (tr - n)
    for wing in range(2):
        index_frac = frac * num_table
        offset = index_frac.astype(np.int64)
        eta = index_frac - offset
        room = (nwin - offset) // step
        cnt = np.minimum(room, n + 1 if wing == 0 else n_in - n - 1)
        for i in range(int(cnt.max()) if n_out else 0):           # tap i of every output at once: same order per output
            live = i < cnt
            idx = np.where(live, offset + i * step, 0)
            xi = np.where(live, (n - i) if wing == 0 else (n + i + 1), 0)
            weight = win[idx] + eta * delta[idx]
            upd = (y.astype(np.float64) + weight * x[xi].astype(np.float64)).astype(np.float32)
            y = np.where(live, upd, y)
        frac = scale - frac
    return y


def librosa_resample_kaiser(x, orig_sr, target_sr, name="kaiser_best"):
    """librosa.resample(y, orig_sr, target_sr, res_type=name) (librosa 0.8 / 0.9: fix=True, scale=False)."""
    x = np.asarray(x, dtype=np.float32)
    if orig_sr == target_sr:
        return x
    ratio = float(target_sr) / orig_sr
    n_samples = int(np.ceil(x.shape[-1] * ratio))
    y = resample(x, orig_sr, target_sr, name)
    if y.shape[0] < n_samples:                                     # util.fix_length
        y = np.pad(y, (0, n_samples - y.shape[0]))
    return np.ascontiguousarray(y[:n_samples], dtype=np.float32)
