"""Oracle (test infrastructure): short-time Fourier transforms as the reference consumes them.

Restated third-party semantics (sources are NOT in /root/reference; parity at these
boundaries is unpinned, see oracle/__init__.py):

* ``librosa_stft``  - librosa 0.8/0.9 ``librosa.stft(y, n_fft, hop_length)`` with the defaults the
  reference relies on (win_length=n_fft, periodic Hann in float64, center=True,
  pad_mode="reflect"), called at ssr_eval/metrics.py:27 and ssr_eval/eval.py:29,37-38.
* ``librosa_istft`` - librosa 0.8/0.9 ``librosa.istft(S, length=...)`` (ssr_eval/eval.py:40).
* ``tl_stft_conv`` / ``tl_istft_conv`` - torchlibrosa 0.0.7-0.0.9 ``STFT`` / ``ISTFT`` modules as wrapped by
  ``FDomainHelper`` (ssr_eval/dsp.py:1,21-39) AS PUBLISHED: the DFT x periodic-Hann matrices are computed in
  float64 and stored as float32 ``Conv1d`` weights, the transform is two dense float32 convolutions
  (``F.conv1d(stride=hop)`` forward; Hermitian mirror + two 1x1 ``conv1d`` inverse), the overlap-add is ``F.fold``,
  divided by the folded ``hann**2`` clamped at 1e-11.  Evaluated on torch-CPU float32 - the arithmetic CLASS of the
  reference (the summation order inside ``conv1d`` is the BLAS / oneDNN kernel's, i.e. box-dependent: see
  ``tl_istft_conv(order=...)`` for how far a member of the class moves when only that order changes).
  ``tl_stft`` / ``tl_istft`` are these (round 4; they were the idealisation below before).
* ``tl_stft_ideal`` / ``tl_istft_ideal`` - the same transforms evaluated with a float64 FFT and rounded once: the
  mathematically exact STFT / least-squares ISTFT.  NOT the reference's arithmetic: a low-passed signal's stop band is
  the transform's own round-off floor, which a float64 FFT puts 20 dB below a 2048-term float32 dot product
  (tests/test_oracle.py::test_lowpass_arithmetic_class_sensitivity).  Kept as the yardstick of the float64 HIP engine.
"""
import functools

import numpy as np


def hann_periodic(n):
    """scipy.signal.get_window("hann", n, fftbins=True) in float64."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def window_array(window, n):
    """librosa.filters.get_window(window, n, fftbins=True) in float64 (torchlibrosa STFT.__init__ / ISTFT.__init__ with
    win_length = n_fft, as dsp.py:21-58 builds them): "hann" is restated, every other name goes to scipy as librosa does."""
    if window == "hann":
        return hann_periodic(n)
    import scipy.signal
    return np.asarray(scipy.signal.get_window(window, n, fftbins=True), dtype=np.float64)


def num_frames(n, n_fft, hop):
    """T for a centred STFT: 1 + (n + 2*(n_fft//2) - n_fft)//hop."""
    return 1 + (n + 2 * (n_fft // 2) - n_fft) // hop


def reflect_pad(y, pad, torch_style=False):
    """numpy.pad(mode="reflect") as librosa calls it (the reflection repeats when the signal is shorter than the pad);
    torch_style: F.pad(mode="reflect") of torchlibrosa, which refuses pad >= length."""
    if y.shape[-1] < 1:
        raise ValueError("empty signal")
    if torch_style and y.shape[-1] <= pad:
        raise ValueError("reflect padding needs len(y) > n_fft//2 (got %d <= %d)" % (y.shape[-1], pad))
    return np.pad(y, pad, mode="reflect")


def frame_matrix(y, n_fft, hop, dtype=np.float64, torch_style=False):
    """[T, n_fft] matrix of centred, reflect-padded frames."""
    yp = reflect_pad(np.asarray(y), n_fft // 2, torch_style)
    T = 1 + (yp.shape[0] - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(T)[:, None]
    return yp[idx].astype(dtype)


def librosa_stft(y, n_fft=2048, hop_length=None):
    """complex64 [1+n_fft//2, T] for float32 input (complex128 for float64 input)."""
    y = np.asarray(y)
    if hop_length is None:
        hop_length = n_fft // 4
    frames = frame_matrix(y, n_fft, hop_length)                  # f64 [T, n_fft]
    spec = np.fft.rfft(frames * hann_periodic(n_fft)[None, :], axis=1)  # f64 pocketfft
    out_dtype = np.complex128 if y.dtype == np.float64 else np.complex64
    return np.ascontiguousarray(spec.T).astype(out_dtype)


def stft_mag_TF(y, n_fft, hop):
    """|librosa.stft| transposed to [T, F] float32: ssr_eval/metrics.py:27-28."""
    return np.ascontiguousarray(np.abs(librosa_stft(y, n_fft, hop)).T)


def window_sumsquare(n_frames, n_fft, hop):
    """Sum of squared periodic-Hann windows at every sample of the padded OLA buffer (float64)."""
    w2 = hann_periodic(n_fft) ** 2
    out = np.zeros(n_fft + hop * (n_frames - 1), dtype=np.float64)
    for t in range(n_frames):
        out[t * hop:t * hop + n_fft] += w2
    return out


def librosa_istft(S, hop_length=None, length=None):
    """librosa 0.8/0.9 istft (window=hann, center=True)."""
    S = np.asarray(S)
    n_fft = 2 * (S.shape[0] - 1)
    if hop_length is None:
        hop_length = n_fft // 4
    n_frames = S.shape[1]
    if length is not None:
        padded = length + n_fft
        n_frames = min(n_frames, int(np.ceil(padded / hop_length)))
    rdtype = np.float64 if S.dtype == np.complex128 else np.float32
    win = hann_periodic(n_fft)
    y = np.zeros(n_fft + hop_length * (n_frames - 1), dtype=np.float64)
    frames = np.fft.irfft(S[:, :n_frames].astype(np.complex128), n=n_fft, axis=0) * win[:, None]
    for t in range(n_frames):
        y[t * hop_length:t * hop_length + n_fft] += frames[:, t]
    y = y.astype(rdtype)
    wss = window_sumsquare(n_frames, n_fft, hop_length).astype(rdtype)
    nz = wss > np.finfo(rdtype).tiny
    y[nz] /= wss[nz]
    start = n_fft // 2
    if length is None:
        return y[start:-start]
    y = y[start:]
    if y.shape[0] >= length:
        return y[:length]
    return np.pad(y, (0, length - y.shape[0]))


def tl_stft_ideal(x, n_fft=2048, hop=441):
    """torchlibrosa STFT.forward evaluated exactly (float64 FFT, one rounding): x [B, n] float32 -> (real, imag) each
    [B, 1, T, F] float32."""
    x = np.asarray(x, dtype=np.float32)
    win = hann_periodic(n_fft)
    re, im = [], []
    for b in range(x.shape[0]):
        spec = np.fft.rfft(frame_matrix(x[b], n_fft, hop, torch_style=True) * win[None, :], axis=1)
        re.append(spec.real.astype(np.float32))
        im.append(spec.imag.astype(np.float32))
    return np.stack(re)[:, None], np.stack(im)[:, None]


def tl_istft_ideal(real, imag, length, n_fft=2048, hop=441):
    """torchlibrosa ISTFT.forward(real, imag, length) evaluated exactly (float64): [B,1,T,F] x2 -> [B, length] float32.

    Hermitian-extend, inverse DFT fused with the synthesis Hann window, overlap-add at stride hop,
    divide by the overlap-added squared window clamped to [1e-11, inf), keep
    [n_fft//2 : n_fft//2 + length].
    """
    real = np.asarray(real, dtype=np.float64)
    imag = np.asarray(imag, dtype=np.float64)
    B, _, T, F = real.shape
    assert F == n_fft // 2 + 1
    win = hann_periodic(n_fft)
    wss = np.maximum(window_sumsquare(T, n_fft, hop), 1e-11)
    out = np.zeros((B, length), dtype=np.float32)
    for b in range(B):
        spec = real[b, 0] + 1j * imag[b, 0]
        # the conv-IDFT uses Re/Im of every mirrored bin; a plain irfft drops Im of DC/Nyquist,
        # which is what the real-valued result of the full Hermitian-extended IDFT also does.
        frames = np.fft.irfft(spec, n=n_fft, axis=1) * win[None, :]
        y = np.zeros(n_fft + hop * (T - 1), dtype=np.float64)
        for t in range(T):
            y[t * hop:t * hop + n_fft] += frames[t]
        y = y / wss
        seg = y[n_fft // 2:n_fft // 2 + length]
        out[b, :seg.shape[0]] = seg.astype(np.float32)
    return out


# ------------------------------------------------------------------------------------------------------------------
# torchlibrosa as published: float32 dense-DFT convolutions on torch-CPU
# ------------------------------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=8)
def tl_weights(n_fft, window="hann"):
    """The four float32 weight matrices torchlibrosa builds (DFTBase.dft_matrix / idft_matrix, STFT.__init__,
    ISTFT.init_real_imag_conv): computed in float64 / complex128, stored float32.

    fwd_re, fwd_im : [F, n_fft]      Re / Im of (W[:, :F] * hann[:, None]).T,  W[j, k] = exp(-2 pi i j k / n)
    inv_re, inv_im : [n_fft, n_fft]  Re / Im of (conj-W / n * hann[None, :]).T  (row = output sample, column = bin)
    """
    n = n_fft
    F = n // 2 + 1
    x, y = np.meshgrid(np.arange(n), np.arange(n))
    omega = np.exp(-2 * np.pi * 1j / n)
    W = np.power(omega, x * y)                         # DFTBase.dft_matrix
    Wi = np.power(np.exp(2 * np.pi * 1j / n), x * y)   # DFTBase.idft_matrix
    win = window_array(window, n)
    fw = W[:, :F] * win[:, None]
    iw = (Wi / n) * win[None, :]
    return (np.ascontiguousarray(np.real(fw).T).astype(np.float32), np.ascontiguousarray(np.imag(fw).T).astype(np.float32),
            np.ascontiguousarray(np.real(iw).T).astype(np.float32), np.ascontiguousarray(np.imag(iw).T).astype(np.float32))


def tl_stft_conv(x, n_fft=2048, hop=441, window="hann", center=True, pad_mode="reflect"):
    """torchlibrosa STFT.forward as published: x [B, n] float32 -> (real, imag) each [B, 1, T, F] float32.
    `if self.center: x = F.pad(x, (n_fft//2, n_fft//2), mode=self.pad_mode)` + two F.conv1d(stride=hop) with the float32
    DFT x window weights.  (The non-default center / pad_mode / window branches are restated from the published module text;
    the reference itself only ever builds FDomainHelper() with the defaults - lowpass.py:18 - so no golden vector covers them.)"""
    import torch
    import torch.nn.functional as Fn
    x = torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
    if center and pad_mode == "reflect" and x.shape[-1] <= n_fft // 2:
        raise ValueError("reflect padding needs len(y) > n_fft//2 (got %d <= %d)" % (x.shape[-1], n_fft // 2))
    fr, fi, _, _ = tl_weights(n_fft, window)
    xp = x[:, None, :]
    if center:
        xp = Fn.pad(xp, (n_fft // 2, n_fft // 2), mode=pad_mode)
    if xp.shape[-1] < n_fft:
        raise ValueError("signal shorter than one frame")
    re = Fn.conv1d(xp, torch.from_numpy(fr)[:, None, :], stride=hop)      # [B, F, T]
    im = Fn.conv1d(xp, torch.from_numpy(fi)[:, None, :], stride=hop)
    return (re[:, None].transpose(2, 3).contiguous().numpy(), im[:, None].transpose(2, 3).contiguous().numpy())


def tl_window_sum_f32(T, n_fft, hop, window="hann"):
    """ISTFT._get_ifft_window: F.fold of the float32 window**2 (float32 additions in frame order), clamped at 1e-11."""
    import torch
    import torch.nn.functional as Fn
    w2 = torch.from_numpy((window_array(window, n_fft) ** 2).astype(np.float32))
    L = (T - 1) * hop + n_fft
    s = Fn.fold(w2[None, :, None].repeat(1, 1, T), output_size=(1, L), kernel_size=(1, n_fft), stride=(1, hop))
    return torch.clamp(s.squeeze(), 1e-11, np.inf)


def tl_istft_conv(real, imag, length, n_fft=2048, hop=441, order=None, window="hann", center=True):
    """torchlibrosa ISTFT.forward as published: [B,1,T,F] x2 float32 -> [B, length] float32.

    Hermitian mirror (torch.cat + flip), s = conv_real(full_re) - conv_imag(full_im) (two 1x1 conv1d with the float32
    IDFT x window weights), F.fold overlap-add, / clamp(folded window**2, 1e-11), ISTFT._trim_edges: [start : start + length]
    with start = n_fft//2 if center else 0 (a slice past the overlap-added signal just ends; the rest of `out` stays zero).

    order: None = torch's own conv1d.  A permutation of range(n_fft) = the SAME float32 products summed in that bin
    order (a sequential float32 accumulation per output sample): a different member of the same arithmetic class, used to
    measure how far a metric moves under the one thing the published code leaves to the BLAS kernel."""
    import torch
    import torch.nn.functional as Fn
    real = torch.as_tensor(np.ascontiguousarray(np.asarray(real, dtype=np.float32)))
    imag = torch.as_tensor(np.ascontiguousarray(np.asarray(imag, dtype=np.float32)))
    B, _, T, F = real.shape
    assert F == n_fft // 2 + 1
    _, _, ir, ii = tl_weights(n_fft, window)
    re = real[:, 0].transpose(1, 2)                       # [B, F, T]
    im = imag[:, 0].transpose(1, 2)
    full_re = torch.cat((re, torch.flip(re[:, 1:-1, :], dims=[1])), dim=1)
    full_im = torch.cat((im, -torch.flip(im[:, 1:-1, :], dims=[1])), dim=1)
    if order is None:
        s = Fn.conv1d(full_re, torch.from_numpy(ir)[:, :, None]) - Fn.conv1d(full_im, torch.from_numpy(ii)[:, :, None])
    else:
        s = torch.from_numpy(_seq_f32_matmul(ir, full_re.numpy(), order) - _seq_f32_matmul(ii, full_im.numpy(), order))
    L = (T - 1) * hop + n_fft
    y = Fn.fold(s, output_size=(1, L), kernel_size=(1, n_fft), stride=(1, hop))[:, 0, 0, :]
    y = y / tl_window_sum_f32(T, n_fft, hop, window)[None, :]
    start = n_fft // 2 if center else 0
    y = y[:, start:start + length]
    out = np.zeros((B, length), dtype=np.float32)
    out[:, :y.shape[1]] = y.numpy()
    return out


def _seq_f32_matmul(w, x, order, fma=False):
    """out[b, o, t] = sum_k w[o, k] x[b, k, t], accumulated SEQUENTIALLY in float32 over k in `order`, every product and
    every sum rounded (fma=True: one rounding per term, computed through float64 - exact for float32 operands).
    Terms whose x row is all zero are skipped (adding an exact zero changes nothing)."""
    B, K, T = x.shape
    out = np.zeros((B, w.shape[0], T), dtype=np.float32)
    for k in order:
        xk = x[:, k, :]
        if not xk.any():
            continue
        if fma:
            out = (out.astype(np.float64) + w[None, :, k, None].astype(np.float64) * xk[:, None, :].astype(np.float64)
                   ).astype(np.float32)
        else:
            out = out + w[None, :, k, None] * xk[:, None, :]
    return out


tl_stft = tl_stft_conv
tl_istft = tl_istft_conv
