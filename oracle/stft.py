"""Oracle (test infrastructure): short-time Fourier transforms as the reference consumes them.

Restated third-party semantics (sources are NOT in /root/reference; parity at these
boundaries is unpinned, see oracle/__init__.py):

* ``librosa_stft``  - librosa 0.8/0.9 ``librosa.stft(y, n_fft, hop_length)`` with the defaults the
  reference relies on (win_length=n_fft, periodic Hann in float64, center=True,
  pad_mode="reflect"), called at ssr_eval/metrics.py:27 and ssr_eval/eval.py:29,37-38.
* ``librosa_istft`` - librosa 0.8/0.9 ``librosa.istft(S, length=...)`` (ssr_eval/eval.py:40).
* ``tl_stft`` / ``tl_istft`` - torchlibrosa 0.0.7-0.0.9 ``STFT`` / ``ISTFT`` modules as wrapped by
  ``FDomainHelper`` (ssr_eval/dsp.py:21-39).  torchlibrosa evaluates the DFT as a float32
  conv1d; here the same transform is evaluated with a float64 FFT and rounded once.
"""
import numpy as np


def hann_periodic(n):
    """scipy.signal.get_window("hann", n, fftbins=True) in float64."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


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


def tl_stft(x, n_fft=2048, hop=441):
    """torchlibrosa STFT.forward: x [B, n] float32 -> (real, imag) each [B, 1, T, F] float32."""
    x = np.asarray(x, dtype=np.float32)
    win = hann_periodic(n_fft)
    re, im = [], []
    for b in range(x.shape[0]):
        spec = np.fft.rfft(frame_matrix(x[b], n_fft, hop, torch_style=True) * win[None, :], axis=1)
        re.append(spec.real.astype(np.float32))
        im.append(spec.imag.astype(np.float32))
    return np.stack(re)[:, None], np.stack(im)[:, None]


def tl_istft(real, imag, length, n_fft=2048, hop=441):
    """torchlibrosa ISTFT.forward(real, imag, length): [B,1,T,F] x2 -> [B, length] float32.

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
