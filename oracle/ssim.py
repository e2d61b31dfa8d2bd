"""Oracle (test infrastructure): skimage.metrics.structural_similarity as the reference calls it.

Call site: ssr_eval/metrics.py:131 ``ssim(output[bs, c], target[bs, c], win_size=7)`` on float32
[T, F] magnitude images, WITHOUT data_range.  scikit-image is not in /root/reference nor in this
image; the semantics restated are those of releases that still infer the range of float images
(data_range = dtype_range[float] span = 2.0), uniform 7x7 window, sample covariance, K1=.01, K2=.03,
border crop (win_size-1)//2, float64 internals (skimage <= 0.18).  Parity unpinned at this boundary.
"""
import numpy as np
from scipy.ndimage import uniform_filter

WIN = 7
DATA_RANGE = 2.0
K1, K2 = 0.01, 0.03
C1 = (K1 * DATA_RANGE) ** 2
C2 = (K2 * DATA_RANGE) ** 2
COV_NORM = (WIN * WIN) / (WIN * WIN - 1.0)


def ssim_map(im1, im2, internal=np.float64):
    a = np.asarray(im1).astype(internal)
    b = np.asarray(im2).astype(internal)
    if a.shape != b.shape or a.ndim != 2:
        raise ValueError("ssim needs two 2-D images of the same shape")
    if min(a.shape) < WIN:
        raise ValueError("win_size exceeds image extent")
    ux = uniform_filter(a, size=WIN)
    uy = uniform_filter(b, size=WIN)
    uxx = uniform_filter(a * a, size=WIN)
    uyy = uniform_filter(b * b, size=WIN)
    uxy = uniform_filter(a * b, size=WIN)
    vx = COV_NORM * (uxx - ux * ux)
    vy = COV_NORM * (uyy - uy * uy)
    vxy = COV_NORM * (uxy - ux * uy)
    num = (2 * ux * uy + C1) * (2 * vxy + C2)
    den = (ux * ux + uy * uy + C1) * (vx + vy + C2)
    return num / den


def structural_similarity(im1, im2, win_size=WIN, internal=np.float64):
    assert win_size == WIN
    pad = (WIN - 1) // 2
    S = ssim_map(im1, im2, internal)
    return float(S[pad:-pad, pad:-pad].astype(np.float64).mean())


def structural_similarity_direct(im1, im2):
    """Independent formulation (explicit valid-window box sums, no ndimage) used to pin ssim_map."""
    a = np.asarray(im1, dtype=np.float64)
    b = np.asarray(im2, dtype=np.float64)

    def box(z):
        c = np.cumsum(np.cumsum(np.pad(z, ((1, 0), (1, 0))), axis=0), axis=1)
        return (c[WIN:, WIN:] - c[:-WIN, WIN:] - c[WIN:, :-WIN] + c[:-WIN, :-WIN]) / (WIN * WIN)

    ux, uy, uxx, uyy, uxy = box(a), box(b), box(a * a), box(b * b), box(a * b)
    vx = COV_NORM * (uxx - ux * ux)
    vy = COV_NORM * (uyy - uy * uy)
    vxy = COV_NORM * (uxy - ux * uy)
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    return float(S.mean())
