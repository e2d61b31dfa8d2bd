"""Oracle (test infrastructure): ctypes driver of oracle/tl_chain.c - torchlibrosa's dense float32 DFT low-pass with the
accumulation order written out: the order torch-CPU's F.conv1d runs (forward: one chain of n_fft fused multiply-adds; inverse:
chains over blocks of 256 channels of the full spectrum), which the HIP `SSR_LOWPASS_CONV` engine computes bit for bit.
See the C file's header for the reference citations and how the order was established."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import stft as _stft

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "tl_chain.c")
SO = os.path.join(_HERE, "_build", "libtlchain.so")
KB_FWD = 0        # forward chain length: 0 = one chain over the n_fft samples of a frame (torch's strided conv1d, >= 55 frames)
KB_INV = 256      # inverse: channels of the FULL spectrum per chain block (ssr_eval_amd/csrc/ssr_tl_gemm.h: SSR_TL_KBF)


def build(force=False):
    if force or not os.path.exists(SO) or os.path.getmtime(SRC) > os.path.getmtime(SO):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", SO, SRC, "-lm"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def transposed_weights(n_fft, weights=None, window="hann"):
    """(fwd_re_t [n_fft, F], fwd_im_t, inv_re_t [n_fft(bin), n_fft(sample)], inv_im_t, window^2 float32)."""
    fr, fi, ir, ii = weights if weights is not None else _stft.tl_weights(n_fft, window)
    w2 = (_stft.window_array(window, n_fft) ** 2).astype(np.float32)
    return (np.ascontiguousarray(fr.T), np.ascontiguousarray(fi.T), np.ascontiguousarray(ir.T), np.ascontiguousarray(ii.T), w2)


def stft(x, n_fft=2048, hop=441, kb=KB_FWD, weights=None, nb=None, window="hann", center=True, pad_mode="reflect"):
    """x [n] float32 -> (re, im) [T, nb] float32 (nb = n_fft//2+1 by default)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    frt, fit, _, _, _ = transposed_weights(n_fft, weights, window)
    nb = n_fft // 2 + 1 if nb is None else nb
    if not center:
        xp = x
    elif pad_mode == "reflect":
        xp = np.ascontiguousarray(_stft.reflect_pad(x, n_fft // 2, torch_style=True))
    elif pad_mode == "constant":
        xp = np.ascontiguousarray(np.pad(x, n_fft // 2))
    else:
        raise ValueError(pad_mode)
    if len(xp) < n_fft:
        raise ValueError("signal shorter than one frame")
    T = 1 + (len(xp) - n_fft) // hop
    re = np.empty((T, nb), np.float32)
    im = np.empty((T, nb), np.float32)
    lib().tl_chain_stft(_f(xp), T, n_fft, hop, _f(frt), _f(fit), C.c_int64(frt.shape[1]), nb, kb, _f(re), _f(im))
    return re, im


def istft(re, im, length, n_fft=2048, hop=441, kbf=KB_INV, weights=None, nbz=None, window="hann", center=True):
    re = np.ascontiguousarray(re, dtype=np.float32)
    im = np.ascontiguousarray(im, dtype=np.float32)
    T, nb = re.shape
    _, _, irt, iit, w2 = transposed_weights(n_fft, weights, window)
    out = np.empty(length, np.float32)
    lib().tl_chain_istft(_f(re), _f(im), T, nb, nb if nbz is None else nbz, n_fft, hop, _f(irt), _f(iit), _f(w2), kbf, length,
                         n_fft // 2 if center else 0, _f(out))
    return out


def stft_hard_lowpass(x, cut, n_fft=2048, hop=441, kb=KB_FWD, kbf=KB_INV, weights=None, window="hann", center=True, pad_mode="reflect"):
    """ssr_eval/lowpass.py:17-28 with cut = the first zeroed bin."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    F = n_fft // 2 + 1
    cut = min(int(cut), F)
    re, im = stft(x, n_fft, hop, kb, weights, nb=cut, window=window, center=center, pad_mode=pad_mode)
    lib().tl_chain_magphase_cut(_f(re), _f(im), C.c_int64(re.shape[0]), cut, cut, C.c_float(1e-8))
    return istft(re, im, len(x), n_fft, hop, kbf, weights, window=window, center=center)
