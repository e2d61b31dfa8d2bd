"""Oracle (test infrastructure): BasicTestee's integer cutoff search restated (ssr_eval/eval.py:21-31).

librosa.stft (third party, absent) is oracle.stft.librosa_stft - complex64 [F, T], C-contiguous, the layout
tests/golden/make_golden.py's stub hands the imported reference (golden `bt_cutoff_index`); NumPy's float32 np.sum over the last,
contiguous axis is a pairwise summation, np.cumsum a sequential one.  Nothing in the product imports this."""
import numpy as np

from . import stft as _stft


def find_cutoff(x, threshold=0.95):
    """ssr_eval/eval.py:21-26, statement for statement."""
    threshold = x[-1] * threshold
    for i in range(1, x.shape[0]):
        if x[-i] < threshold:
            return x.shape[0] - i
    return 0


def bin_energy(x):
    """np.sum(np.abs(librosa.stft(x)), axis=-1): float32 [1025] (ssr_eval/eval.py:29-30)."""
    return np.sum(np.abs(_stft.librosa_stft(np.asarray(x), 2048, 512)), axis=-1)


def get_cutoff_index(x):
    """ssr_eval/eval.py:28-31."""
    return find_cutoff(np.cumsum(bin_energy(x)), 0.97)
