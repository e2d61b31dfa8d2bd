"""Degradations - drop-in for ssr_eval.lowpass (ssr_eval/lowpass.py:17-256) on MI355X.

``lowpass(data, highcut, fs, order, _type)`` keeps the reference's dispatcher semantics (order clamp to
[2, 10], 1-D check, *substring* match of ``_type``).  Hot-path types run in libssrhip.so:

* ``stft_hard``   -> ``ssr_fft_lowpass`` (K6): STFT(2048/441) -> zero bins >= cut -> ISTFT.
* ``subsampling`` -> ``ssr_resample_poly`` (K7) down then up, then align_length.

* butter / cheby1 / ellip / bessel -> ``ssr_sosfiltfilt``: the section coefficients are
  designed on the host with SciPy exactly as the reference does (filter design is a plan, not data), the
  zero-phase filtering itself (odd extension, forward and backward second-order-section recurrences) runs on the
  GPU, bit-identical to ``scipy.signal.sosfiltfilt`` for float32 and for float64 input.
"""
import functools

import numpy as np
import torch

from . import backend as B

N_FFT, HOP = 2048, 441          # FDomainHelper() defaults used by the reference's global f_helper (lowpass.py:167)
_precision = "f64"
# Engine of the STFT-domain low-pass (include/ssr_hip.h: ssr_plan_set_lowpass_engine).  "conv" = the reference's arithmetic
# (torchlibrosa's dense float32 DFT convolutions with the module's own weights, accumulated in torch-CPU's F.conv1d order, on the
# fp32 matrix cores): for a signal of >= 55 frames (0.55 s at 44.1 kHz) the degraded waveform is the reference's bit for bit, so
# its stop band - the transform's round-off floor, whose logarithm LSD / log-SISpec take - is the reference's too (DESIGN.md
# section 4); shorter signals, where torch switches its forward convolution to another summation order, agree to the spread of
# the arithmetic class.  "segments" / "fused" = float64 FFT engines: the exact low-pass, 10-30x faster, 2-7 % off in LSD of the
# degraded input - for callers that want the ideal filter rather than the reference's numbers.
DEFAULT_ENGINE = "conv"


def cut_bin(lowpass_ratio, n_bins=N_FFT // 2 + 1):
    """First zeroed bin: int(F * ratio) (lowpass.py:24) - integer, bit-exact with the reference."""
    return int(n_bins * lowpass_ratio)


def stft_hard_lowpass_v0(data, lowpass_ratio, engine=None):
    """lowpass.py:17-28.  data: 1-D ndarray / tensor -> float32 ndarray of the same length."""
    return stft_hard_lowpass_batch([data], [lowpass_ratio], engine=engine)[0]


def stft_hard_lowpass_batch(datas, ratios, device=None, keep_on_device=False, engine=None):
    """keep_on_device: the degraded signals stay device tensors (views of the launch's output) for a GPU consumer.
    engine: None = DEFAULT_ENGINE ("conv": the reference's arithmetic), or "segments" / "fused" (float64 FFT)."""
    plan = B.get_plan(N_FFT, HOP, _precision, device, lowpass_engine=engine or DEFAULT_ENGINE)
    ys = B.fft_lowpass(plan, [d if isinstance(d, torch.Tensor) else np.asarray(d, np.float32) for d in datas],
                       [cut_bin(r) for r in ratios])
    return list(ys) if keep_on_device else [y.cpu().numpy() for y in ys]


def stft_hard_lowpass_multi(datas, ratios, device=None, keep_on_device=False, engine=None):
    """Every waveform of `datas` low-passed at EVERY ratio of `ratios` - the loop of SSR_Eval_Helper.lowpass_stft_hard over
    setting_fft (ssr_eval/eval.py:401-410) for a batch of files: [[outputs at ratios[0]], [at ratios[1]], ...].  One
    ssr_fft_lowpass_multi call: on the conv engine the padded copy and the forward dense-DFT product are shared by the ratios."""
    plan = B.get_plan(N_FFT, HOP, _precision, device, lowpass_engine=engine or DEFAULT_ENGINE)
    ys = B.fft_lowpass_multi(plan, [d if isinstance(d, torch.Tensor) else np.asarray(d, np.float32) for d in datas],
                             [cut_bin(r) for r in ratios])
    return [list(y) if keep_on_device else [v.cpu().numpy() for v in y] for y in ys]


def align_length(x, y):
    """Zero-pad or cut y to len(x) (lowpass.py:31-51)."""
    if y.shape[0] == x.shape[0]:                  # (what every filter of this module returns: nothing to pad or cut)
        return y
    if len(y) < len(x):
        if isinstance(y, torch.Tensor):
            return torch.nn.functional.pad(y, (0, len(x) - len(y)))
        return np.pad(y, (0, len(x) - len(y)), mode="constant")
    return y[:len(x)]


def _f32_unless_f64(x):
    """The dtype rule of the degradations: a float64 signal is processed on its float64 values, anything else as float32."""
    if isinstance(x, torch.Tensor):
        return x if x.dtype in (torch.float32, torch.float64) else x.float()
    return x if x.dtype == np.float64 else x.astype(np.float32)


def subsampling(data, lowpass_ratio, fs_ori=44100):
    """Down- then up-sample through fs_down = int(ratio * 44100) (lowpass.py:134-144)."""
    fs_down = int(lowpass_ratio * fs_ori)
    x = np.asarray(data)
    return _subsample_batch([x], fs_down, fs_ori)[0]


def _subsample_batch(xs, fs_down, fs_ori=44100, keep_on_device=False):
    """float64 signals are resampled in float64, everything else in float32 (what SciPy does per dtype)."""
    outs = [None] * len(xs)
    for want64 in (False, True):
        idx = [i for i, x in enumerate(xs) if B._is_f64(x) == want64]
        if not idx:
            continue
        sel = [_f32_unless_f64(xs[i]) for i in idx]
        up = B.resample_poly(B.resample_poly(sel, fs_down, fs_ori), fs_ori, fs_down)
        for i, u in zip(idx, up):
            outs[i] = align_length(xs[i], u if keep_on_device else u.cpu().numpy())
    return outs


def limit(integer, high, low):
    return high if integer > high else (low if integer < low else int(integer))


def _design(highcut, fs, order, ftype, lowcut=None):
    """Section design of lowpass.py:70-85 / :110-125 (SciPy on the host: a plan, not data; cached per argument tuple - an evaluate()
    pass asks for the same 36 designs once per batch of files).  The caller gets its own copy."""
    return _design_cached(highcut, fs, order, ftype, lowcut).copy()


@functools.lru_cache(maxsize=512)
def _design_cached(highcut, fs, order, ftype, lowcut):
    from scipy import signal
    nyq = 0.5 * fs
    wn = highcut / nyq if lowcut is None else [lowcut / nyq, highcut / nyq]
    bt = "low" if lowcut is None else "band"
    design = {"butter": lambda: signal.butter(order, wn, btype=bt, output="sos"),
              "cheby1": lambda: signal.cheby1(order, 0.1, wn, btype=bt, output="sos"),
              "cheby2": lambda: signal.cheby2(order, 60, wn, btype=bt, output="sos"),
              "ellip": lambda: signal.ellip(order, 0.1, 60, wn, btype=bt, output="sos"),
              "bessel": lambda: signal.bessel(order, wn, btype=bt, output="sos")}
    if ftype not in design:
        raise Exception("The %s filter %s is not supported!" % ("lowpass" if lowcut is None else "bandpass", ftype))
    return design[ftype]()


def _iir_batch(xs, highcut, fs, order, ftype, lowcut=None, keep_on_device=False):
    """lowpass.py:54-131 for a list of signals: one ssr_sosfiltfilt launch (GPU, float64, bit-exact with SciPy)."""
    sos = _design(highcut, fs, order, ftype, lowcut)
    xs = [x if isinstance(x, torch.Tensor) else np.asarray(x) for x in xs]
    outs = [None] * len(xs)
    for want64 in (False, True):                     # float64 signals are filtered on their float64 values
        idx = [i for i, x in enumerate(xs) if B._is_f64(x) == want64]
        if idx:
            ys = B.sosfiltfilt(sos, [_f32_unless_f64(xs[i]) for i in idx])
            for i, y in zip(idx, ys):
                outs[i] = align_length(xs[i], y if keep_on_device else y.cpu().numpy())
    return outs


def lowpass_iir_multi(datas, specs, fs, keep_on_device=False):
    """lowpass(d, highcut, fs, order, _type) for every (highcut, order, _type) of `specs` over one list of signals - what
    SSR_Eval_Helper.preprocess's three nested loops apply to a waveform (ssr_eval/eval.py:243-258) - with the designs side by side in
    one launch (backend.sosfiltfilt_multi).  -> [spec][signal]; each entry equals lowpass_batch(datas, highcut, fs, order, _type)."""
    plans = []                         # (int(highcut), clamped order, design name) per spec: BOTH paths below use these
    for highcut, order, _type in specs:
        name = next((n for n in ("butter", "cheby1", "ellip", "bessel") if _type in n), None)
        if name is None:
            raise ValueError("Error: Unexpected filter type " + _type)
        plans.append((int(highcut), limit(order, high=10, low=2), name))
    for d in datas:
        _check_1d(d)
    if not plans:                      # an empty filter / cutoff / order list: no keys, as the reference's loops
        return []
    xs = [x if isinstance(x, torch.Tensor) else np.asarray(x) for x in datas]
    if not xs or any(B._is_f64(x) for x in xs):
        # float64 signals are filtered on their float64 values: the per-design path handles the mix (same integer cutoff,
        # clamped order and design name as the one-launch path)
        return [_iir_batch(xs, hc, fs, order, name, keep_on_device=keep_on_device) for hc, order, name in plans]
    designs = [_design(hc, fs, order, name) for hc, order, name in plans]
    ys = B.sosfiltfilt_multi(designs, [_f32_unless_f64(x) for x in xs])
    return [[align_length(x, y if keep_on_device else y.cpu().numpy()) for x, y in zip(xs, per_design)] for per_design in ys]


def _iir(x, highcut, fs, order, ftype, lowcut=None):
    return _iir_batch([x], highcut, fs, order, ftype, lowcut)[0]


def lowpass_filter(x, highcut, fs, order, ftype):
    return _iir(x, highcut, fs, order, ftype)


def bandpass_filter(x, lowcut, highcut, fs, order, ftype):
    return _iir(x, highcut, fs, order, ftype, lowcut=lowcut)


def _check_1d(data):
    if len(list(data.shape)) != 1:
        raise ValueError("Error (chebyshev_lowpass_filter): Data " + str(data.shape)
                         + " should be type 1d time array, (samples,) , can not be (samples, 1)")


def lowpass(data, highcut, fs, order=5, _type="butter"):
    """lowpass.py:156-196."""
    order = limit(order, high=10, low=2)
    _check_1d(data)
    for name in ("butter", "cheby1", "ellip", "bessel"):      # `_type in name`: substring test, as the reference
        if _type in name:
            return lowpass_filter(x=data, highcut=int(highcut), fs=fs, order=order, ftype=name)
    if _type in "subsampling":
        return subsampling(data, lowpass_ratio=highcut / int(fs / 2))
    if _type in "stft_hard":
        return stft_hard_lowpass_v0(data, lowpass_ratio=highcut / int(fs / 2))
    raise ValueError("Error: Unexpected filter type " + _type)


def lowpass_batch(datas, highcut, fs, order=5, _type="butter", keep_on_device=False):
    """lowpass() for a LIST of 1-D signals with one batched launch sequence per call (same dispatch semantics).
    keep_on_device: results stay device tensors (inputs may be device tensors too) - the resident evaluation path."""
    order = limit(order, high=10, low=2)
    for d in datas:
        _check_1d(d)
    for name in ("butter", "cheby1", "ellip", "bessel"):
        if _type in name:
            return _iir_batch(datas, int(highcut), fs, order, name, keep_on_device=keep_on_device)
    if _type in "subsampling":
        ratio = highcut / int(fs / 2)
        fs_down = int(ratio * 44100)
        return _subsample_batch([d if isinstance(d, torch.Tensor) else np.asarray(d) for d in datas], fs_down, keep_on_device=keep_on_device)
    if _type in "stft_hard":
        return stft_hard_lowpass_batch(datas, [highcut / int(fs / 2)] * len(datas), keep_on_device=keep_on_device)
    raise ValueError("Error: Unexpected filter type " + _type)


def bandpass(data, lowcut, highcut, fs, order=5, _type="butter"):
    """lowpass.py:199-256."""
    _check_1d(data)
    for name in ("butter", "cheby1", "ellip", "bessel"):
        if _type in name:
            return bandpass_filter(x=data, lowcut=int(lowcut), highcut=int(highcut), fs=fs,
                                   order=limit(order, high=10, low=2), ftype=name)
    raise ValueError("Error: Unexpected filter type " + _type)
