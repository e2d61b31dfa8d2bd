"""Oracle (test infrastructure): the degradations of ssr_eval/lowpass.py.

stft_hard_lowpass (lowpass.py:17-28 on FDomainHelper(2048, 441), dsp.py:76-119), subsampling
(lowpass.py:134-144), align_length (lowpass.py:31-51), the dispatcher's integer arithmetic
(lowpass.py:156-196) and the zero-phase IIR low-pass (lowpass.py:94-131; SciPy is the primitive).
"""
import numpy as np
from scipy import signal

from . import stft as _stft

N_FFT, HOP = 2048, 441  # FDomainHelper() defaults, ssr_eval/dsp.py:9-10 via lowpass.py:167


def cut_bin(highcut, fs, n_bins=N_FFT // 2 + 1):
    """lowpass.py:193-194 ratio, lowpass.py:24 int(): bit-exact integer."""
    return int(n_bins * (highcut / int(fs / 2)))


ARITHMETIC = ("conv", "chain", "ideal")


def spectrogram_phase(x, eps=1e-8, n_fft=N_FFT, hop=HOP, arithmetic="conv"):
    """dsp.py:76-81 with eps=1e-8 (dsp.py:83): mag, cos, sin as float32 [B,1,T,F]."""
    re, im = (_stft.tl_stft_ideal if arithmetic == "ideal" else _stft.tl_stft_conv)(x, n_fft, hop)
    mag = np.clip(re ** 2 + im ** 2, np.float32(eps), np.inf) ** np.float32(0.5)
    return mag, re / mag, im / mag


def stft_hard_lowpass(data, lowpass_ratio, n_fft=N_FFT, hop=HOP, arithmetic="conv"):
    """lowpass.py:17-28.

    arithmetic: "conv"  = torchlibrosa as published (float32 dense-DFT convolutions on torch-CPU): the reference's class;
                "chain" = the same with the accumulation order fixed (oracle/tl_chain.c): what the HIP conv engine computes, bit for bit;
                "ideal" = float64 FFT rounded once: the exact low-pass, the yardstick of the HIP float64 engines - NOT the
                          reference's arithmetic (LSD of the result is 2-7 % higher: its stop band is 20 dB cleaner)."""
    data = np.asarray(data, dtype=np.float32)
    length = data.shape[0]
    if arithmetic == "chain":
        from . import tl_chain
        return tl_chain.stft_hard_lowpass(data, int((n_fft // 2 + 1) * lowpass_ratio), n_fft, hop)
    mag, cos, sin = spectrogram_phase(data[None, :], 1e-8, n_fft, hop, arithmetic)
    cut = int(mag.shape[-1] * lowpass_ratio)
    mag[..., cut:] = 0
    istft = _stft.tl_istft_ideal if arithmetic == "ideal" else _stft.tl_istft_conv
    return istft(mag * cos, mag * sin, length, n_fft, hop)[0]


def align_length(x, y):
    """lowpass.py:31-51."""
    if len(x) == len(y):
        return y
    if len(x) > len(y):
        return np.pad(y, (0, len(x) - len(y)), mode="constant")
    return y[:len(x)]


def subsampling(data, lowpass_ratio, fs_ori=44100):
    """lowpass.py:134-144."""
    fs_down = int(lowpass_ratio * fs_ori)
    y = signal.resample_poly(data, fs_down, fs_ori)
    y = signal.resample_poly(y, fs_ori, fs_down)
    return align_length(data, y)


def iir_sos(highcut, fs, order, ftype, lowcut=None):
    """Filter design of lowpass.py:110-125 / :70-85."""
    nyq = 0.5 * fs
    wn = highcut / nyq if lowcut is None else [lowcut / nyq, highcut / nyq]
    bt = "low" if lowcut is None else "band"
    if ftype == "butter":
        return signal.butter(order, wn, btype=bt, output="sos")
    if ftype == "cheby1":
        return signal.cheby1(order, 0.1, wn, btype=bt, output="sos")
    if ftype == "cheby2":
        return signal.cheby2(order, 60, wn, btype=bt, output="sos")
    if ftype == "ellip":
        return signal.ellip(order, 0.1, 60, wn, btype=bt, output="sos")
    if ftype == "bessel":
        return signal.bessel(order, wn, btype=bt, output="sos")
    raise Exception("The filter %s is not supported!" % ftype)


def lowpass_filter(x, highcut, fs, order, ftype):
    """lowpass.py:94-131 (the discarded subsampling() call at :130 is dead compute, not reproduced)."""
    return align_length(x, signal.sosfiltfilt(iir_sos(highcut, fs, order, ftype), x))


def lowpass(data, highcut, fs, order=5, _type="butter", arithmetic="conv"):
    """lowpass.py:156-196 including the substring dispatch."""
    order = 10 if order > 10 else (2 if order < 2 else int(order))  # limit(), lowpass.py:147-153
    if data.ndim != 1:
        raise ValueError("data should be 1-D")
    for name in ("butter", "cheby1", "ellip", "bessel"):
        if _type in name:
            return lowpass_filter(data, int(highcut), fs, order, name)
    if _type in "subsampling":
        return subsampling(data, highcut / int(fs / 2))
    if _type in "stft_hard":
        return stft_hard_lowpass(data, highcut / int(fs / 2), arithmetic=arithmetic)
    raise ValueError("Error: Unexpected filter type " + _type)
