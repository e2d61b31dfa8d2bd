"""Oracle (test infrastructure): AudioMetrics restated on torch-CPU fp32, as the reference runs it.

Follows ssr_eval/metrics.py:15-132 and ssr_eval/utils.py:43-92 operation by operation (same dtype,
same operation order); the two third-party calls are replaced by oracle.stft / oracle.ssim.
"""
import numpy as np
import torch

from . import stft as _stft
from . import ssim as _ssim

EPS = 1e-12  # ssr_eval/metrics.py:12, ssr_eval/utils.py:7


def stft_params(rate):
    """ssr_eval/metrics.py:16-19 -> (n_fft, hop)."""
    return int(2048 / (44100 / rate)), int(rate / 100)


def wav_to_spectrogram(wav, n_fft, hop):
    """ssr_eval/metrics.py:26-30 -> torch float32 [1, 1, T, F].

    As in the reference the tensor is built from the TRANSPOSED VIEW of the [F, T] magnitude array,
    so it keeps T-contiguous strides (0, 0, 1, T); torch's fp32 reductions then run in that memory
    order, which is what defines the round-off of lsd / sispec.
    """
    f = np.abs(_stft.librosa_stft(wav, n_fft, hop))
    return torch.tensor(np.transpose(f, (1, 0))[None, None, ...])


def to_log(x):
    """ssr_eval/utils.py:43-44."""
    return torch.log10(x + 1e-12)


def from_log(x):
    """ssr_eval/utils.py:47-49."""
    return 10 ** torch.clip(x, min=-np.inf, max=5)


def energy_unify(estimated, original):
    """ssr_eval/utils.py:79-82."""
    target = _inner_last_dims(estimated, original) * original
    target /= _sq_norm_all_but_batch(original) + EPS
    return estimated, target


def lsd(est, target):
    """ssr_eval/metrics.py:109-112; est/target [B, C, T, F] float32 -> [B, C, 1, 1]."""
    ratio = target ** 2 / ((est + EPS) ** 2)
    d = torch.log10(ratio + EPS) ** 2
    per_frame = torch.mean(d, dim=3) ** 0.5
    return torch.mean(per_frame, dim=2)[..., None, None]


def _sq_norm_all_but_batch(x):
    """ssr_eval/utils.py:68-76 (pow_p_norm)."""
    dims = list(range(1, x.dim()))
    return torch.pow(torch.norm(x, p=2, dim=dims, keepdim=True), 2)


def _inner_last_dims(a, b):
    """ssr_eval/utils.py:85-92 (pow_norm)."""
    dims = list(range(2, a.dim()))
    return torch.sum(a * b, dim=dims, keepdim=True)


def sispec(est, target):
    """ssr_eval/metrics.py:114-121 with energy_unify (ssr_eval/utils.py:79-82) inlined."""
    scaled = _inner_last_dims(est, target) * target
    scaled /= _sq_norm_all_but_batch(target) + EPS
    noise = est - scaled
    ratio = _sq_norm_all_but_batch(scaled) / (_sq_norm_all_but_batch(noise) + EPS) + EPS
    loss = 10 * torch.log10(ratio)
    return torch.sum(loss) / loss.size()[0]


def sispec_exact(est, target):
    """The SAME formula (metrics.py:114-121 + utils.py:79-82) evaluated in float64 on the float32 inputs: what the
    reference would return if its reductions were exact.  The reference's float32 torch.norm / torch.sum over the
    ~1e6 elements of a long utterance are only good to ~1e-5 relative (measured in
    tests/test_oracle.py::test_reference_float32_sispec_noise_is_measured), so this is the yardstick that separates
    "the kernel is wrong" from "the reference's own round-off"."""
    return sispec(est.double(), target.double())


def ssim(est, target):
    """ssr_eval/metrics.py:123-132 -> float64 [B, C, 1, 1]."""
    e, t = est.numpy(), target.numpy()
    res = np.zeros([e.shape[0], e.shape[1]])
    for b in range(e.shape[0]):
        for c in range(e.shape[1]):
            res[b, c] = _ssim.structural_similarity(e[b, c], t[b, c], win_size=7)
    return torch.tensor(res)[..., None, None]


def spectrogram_metrics(est_sp, target_sp):
    """The four reductions of ssr_eval/metrics.py:98-106 on precomputed [1,1,T,F] spectrograms."""
    return {
        "lsd": float(lsd(est_sp.clone(), target_sp.clone())),
        "log_sispec": float(sispec(to_log(est_sp.clone()), to_log(target_sp.clone()))),
        "sispec": float(sispec(est_sp.clone(), target_sp.clone())),
        "ssim": float(ssim(est_sp.clone(), target_sp.clone())),
    }


def evaluation_with_exact(est, target, rate=None, n_fft=None, hop=None):
    """(evaluation(...), {"log_sispec": ..., "sispec": ...} through sispec_exact) on the same spectrograms."""
    assert est.ndim == 1 and target.ndim == 1
    if n_fft is None:
        n_fft, hop = stft_params(rate)
    m = min(target.shape[0], est.shape[0])
    es, ts = wav_to_spectrogram(est[:m], n_fft, hop), wav_to_spectrogram(target[:m], n_fft, hop)
    exact = {"log_sispec": float(sispec_exact(to_log(es.clone()), to_log(ts.clone()))), "sispec": float(sispec_exact(es, ts))}
    return spectrogram_metrics(es, ts), exact


def evaluation(est, target, rate=None, n_fft=None, hop=None):
    """ssr_eval/metrics.py:51-107 for ndarray inputs; (n_fft, hop) default to AudioMetrics(rate)."""
    if type(est) != type(target):
        raise ValueError("The input value should either both be numpy array or strings")
    assert est.ndim == 1 and target.ndim == 1
    assert abs(target.shape[0] - est.shape[0]) < 100
    if n_fft is None:
        n_fft, hop = stft_params(rate)
    m = min(target.shape[0], est.shape[0])
    target, est = target[:m], est[:m]
    return spectrogram_metrics(wav_to_spectrogram(est, n_fft, hop), wav_to_spectrogram(target, n_fft, hop))
