"""AudioMetrics - drop-in for ssr_eval.metrics.AudioMetrics (ssr_eval/metrics.py:15-132) on MI355X.

Same constructor, method names, argument order (estimate first, target second) and result keys as the
reference.  All arithmetic runs in libssrhip.so (HIP, gfx950):

* ``evaluation`` -> one ``ssr_pair_metrics`` call: STFT of both signals (two-for-one complex FFT),
  fused LSD + SISpec + log-SISpec epilogue, SSIM kernel, finalisation.
* ``wav_to_spectrogram`` -> ``ssr_stft``;  ``lsd`` / ``sispec`` / ``ssim`` on tensors ->
  ``ssr_spectrogram_metrics``.

Extras (keyword-only, not in the reference): ``precision`` ("f64" parity mode / "f32"), ``device``,
``n_fft`` / ``hop_length`` overrides, and ``evaluation_batch`` for lists of pairs.
"""
import numpy as np
import torch

from . import backend as B

EPS = 1e-12
_KEYS = ("lsd", "log_sispec", "sispec", "ssim")


class AudioMetrics:
    def __init__(self, rate, *, precision="f64", device=None, n_fft=None, hop_length=None):
        self.rate = rate
        # integer table of ssr_eval/metrics.py:18-19 (bit-exact): 48000 -> (2229, 480), 44100 -> (2048, 441)
        self.hop_length = int(rate / 100) if hop_length is None else int(hop_length)
        self.n_fft = int(2048 / (44100 / rate)) if n_fft is None else int(n_fft)
        self.precision = precision
        self._device = device

    # ---- plumbing
    def _plan(self):
        return B.get_plan(self.n_fft, self.hop_length, self.precision, self._device)

    def read(self, est, target):
        """ssr_eval/metrics.py:21-24 (file decode + resample is host I/O)."""
        from .io import load_audio
        return load_audio(est, self.rate), load_audio(target, self.rate)

    # ---- reference API
    def wav_to_spectrogram(self, wav, *, keep_on_device=False):
        """[n] waveform -> magnitude spectrogram tensor [1, 1, T, F] float32 (metrics.py:26-30)."""
        plan = self._plan()
        sp = B.stft(plan, [np.asarray(wav) if not isinstance(wav, torch.Tensor) else wav])[0][None, None]
        return sp if keep_on_device else sp.cpu()

    def _prepare_pair(self, est, target, resident=False):
        # resident (internal callers): the signals may be device tensors the previous stage left in HBM, next to ndarrays
        if type(est) != type(target) and not (resident and not isinstance(est, str) and not isinstance(target, str)):
            raise ValueError("The input value should either both be numpy array or strings")
        if isinstance(est, str):
            est, target = self.read(est, target)
        if est.shape == target.shape and len(est.shape) == 1:        # the common case (every key of a file): nothing to check or cut
            return est, target
        assert len(est.shape) == 1 and len(target.shape) == 1, (
            "The input numpy array shape should be [samples,]. Got input shape %s and %s. " % (est.shape, target.shape))
        assert abs(target.shape[0] - est.shape[0]) < 100, (
            "Error: Shape mismatch between target and estimation %s and %s" % (str(target.shape), str(est.shape)))
        m = min(target.shape[0], est.shape[0])           # metrics.py:89-90
        return est[:m], target[:m]

    def evaluation(self, est, target, file=None):
        """{lsd, log_sispec, sispec, ssim} for one (estimate, target) pair (metrics.py:51-107)."""
        return self.evaluation_batch([est], [target])[0]

    def evaluation_batch(self, ests, targets, mask=B.M_ALL, resident=False, deferred=False):
        """The same four metrics for lists of pairs, one fused launch sequence for the whole batch.
        deferred: the launches are queued and a function is returned that waits for the values and builds the list (the caller
        queues its next batch in between)."""
        pairs = [self._prepare_pair(e, t, resident) for e, t in zip(ests, targets)]
        # A float64 estimate (IIR-degraded input passed through a testee, eval.py:138-150) makes the reference's est
        # spectrogram - and with it every metric - float64; float64 targets (arrays decoded as float64 by the caller)
        # likewise.  Pairs are grouped by dtype combination and each group runs its own launch sequence:
        # 0 = both float32, 1 = float64 estimate / float32 target, 2 = float64 target (estimate widened if needed).
        kind = [2 if B._is_f64(p[1]) else (1 if B._is_f64(p[0]) else 0) for p in pairs]
        groups = []
        for want in (0, 1, 2):
            idx = [i for i, f in enumerate(kind) if f == want]
            if idx:
                groups.append((want, idx, B.pair_metrics(self._plan(), [pairs[i][0] for i in idx], [pairs[i][1] for i in idx], mask,
                                                         deferred=True)))

        def finish():
            out = [None] * len(pairs)
            for want, idx, pending in groups:
                for i, row in zip(idx, pending()):
                    # float32 pairs: lsd / sispec are float32 tensors in the reference (float() of fp32), ssim float64
                    out[i] = self._row_dict(row, mask, bool(want))
            return out
        return finish if deferred else finish()

    @staticmethod
    def _row_dict(row, mask, wide):
        d = {}
        for k, v in zip(_KEYS, row):
            if not np.isnan(v) or (mask & (1 << _KEYS.index(k))):
                d[k] = float(v) if (k == "ssim" or wide) else float(np.float32(v))
        return d

    def evaluation_multi(self, ests_by_key, targets, mask=B.M_ALL, resident=False, deferred=False):
        """K estimates per target (the degradation keys of a file, ssr_eval/eval.py:136-154): ests_by_key = K lists of n
        waveforms, targets = n waveforms -> n lists of K dicts.  Every target is transformed once per GROUP of keys: the float32 keys
        of the files (FFT low-pass, subsampling, mp3) through one ssr_pair_metrics_multi launch sequence, the float64 keys (every IIR
        design: sosfiltfilt returns float64 and the reference keeps it, eval.py:138-150) through one ssr_pair_metrics_multi_est64.
        Needs float32 targets and, per item, one truncated length for all keys (metrics.py:89-90) - otherwise (and for a group of a
        single key) the pairs go through evaluation_batch.  deferred: as evaluation_batch."""
        K, n = len(ests_by_key), len(targets)
        pairs = [[self._prepare_pair(ests_by_key[k][i], targets[i], resident) for i in range(n)] for k in range(K)]
        same_len = all(len({pairs[k][i][0].shape[0] for k in range(K)}) == 1 for i in range(n))
        tgt32 = not any(B._is_f64(pairs[k][i][1]) for k in range(K) for i in range(n))
        # a key is float64 / float32 if ALL its estimates are; a key with both kinds sends everything through evaluation_batch
        kinds = [{bool(B._is_f64(pairs[k][i][0])) for i in range(n)} for k in range(K)]
        if K < 2 or n == 0 or not same_len or not tgt32 or any(len(kd) != 1 for kd in kinds):
            flat = self.evaluation_batch([ests_by_key[k][i] for i in range(n) for k in range(K)],
                                         [targets[i] for i in range(n) for _ in range(K)], mask, resident, deferred=True)
            finish = lambda: (lambda rows: [rows[i * K:(i + 1) * K] for i in range(n)])(flat())    # noqa: E731
            return finish if deferred else finish()
        tgts = [pairs[0][i][1] for i in range(n)]

        def one_key(k):                                              # a group of ONE key: the plain pair path -> [n][1] dicts
            flat = self.evaluation_batch([pairs[k][i][0] for i in range(n)], tgts, mask, True, deferred=True)
            return lambda: [[row] for row in flat()]

        def many_keys(keys, wide):                                   # one multi launch sequence -> [n][len(keys)] dicts
            pending = B.pair_metrics_multi(self._plan(), [[pairs[k][i][0] for i in range(n)] for k in keys], tgts, mask, deferred=True)

            def collect():
                vals = pending()
                return [[self._row_dict(vals[i, j], mask, wide) for j in range(len(keys))] for i in range(n)]
            return collect

        parts = []                                                   # (key indices, collector)
        for wide in (False, True):
            keys = [k for k in range(K) if next(iter(kinds[k])) == wide]
            if keys:
                parts.append((keys, one_key(keys[0]) if len(keys) == 1 else many_keys(keys, wide)))

        def finish():
            out = [[None] * K for _ in range(n)]
            for keys, collect in parts:
                rows = collect()
                for i in range(n):
                    for j, k in enumerate(keys):
                        out[i][k] = rows[i][j]
            return out
        return finish if deferred else finish()

    # ---- reductions on [B, C, T, F] tensors (est first)
    @staticmethod
    def _images(x):
        """The B*C [T, F] images of a [B, C, T, F] tensor, batch-major (the reference loops b, then c: metrics.py:128-131)."""
        if x.dim() != 4:
            raise ValueError("expected a [B, C, T, F] tensor, got %s" % (tuple(x.shape),))
        return [x[b, c] for b in range(x.shape[0]) for c in range(x.shape[1])]

    def _reduce(self, est, target, mask):
        if est.shape != target.shape:
            raise ValueError("spectrogram shape mismatch: %s vs %s" % (tuple(est.shape), tuple(target.shape)))
        return B.spectrogram_metrics(self._images(est), self._images(target), mask)

    def lsd(self, est, target):
        """[B, C, T, F] x2 -> [B, C, 1, 1] float32 (metrics.py:109-112; one value per image)."""
        v = self._reduce(est, target, B.M_LSD)[:, 0]
        return v.to(torch.float32).to(est.device).reshape(est.shape[0], est.shape[1], 1, 1)

    def _sispec_multichannel(self, est, target, log_domain):
        """metrics.py:114-121 for C > 1 (one ratio per batch item over an all-channel target energy): ssr_sispec_multichannel."""
        if est.shape != target.shape:
            raise ValueError("spectrogram shape mismatch: %s vs %s" % (tuple(est.shape), tuple(target.shape)))
        return B.sispec_multichannel(est, target, log_domain).to(torch.float32).to(est.device)

    def sispec(self, est, target):
        """Scale-invariant spectrogram-to-noise ratio, mean over the batch, 0-dim float32 (metrics.py:114-121)."""
        if est.dim() == 4 and est.shape[1] != 1:
            return self._sispec_multichannel(est, target, False)
        v = self._reduce(est, target, B.M_SISPEC)[:, 2]
        return (v.sum() / v.shape[0]).to(torch.float32).to(est.device)

    def log_sispec(self, est, target):
        """sispec(to_log(est), to_log(target)) of metrics.py:99-101 with the log10(x + 1e-12) fused in-kernel."""
        if est.dim() == 4 and est.shape[1] != 1:
            return self._sispec_multichannel(est, target, True)
        v = self._reduce(est, target, B.M_LOG_SISPEC)[:, 1]
        return (v.sum() / v.shape[0]).to(torch.float32).to(est.device)

    def ssim(self, est, target):
        """[B, C, T, F] x2 -> [B, C, 1, 1] float64 (metrics.py:123-132; one skimage call per image)."""
        v = self._reduce(est, target, B.M_SSIM)[:, 3]
        return v.to(est.device).reshape(est.shape[0], est.shape[1], 1, 1)

    def center_crop(self, x, y):
        """Crop the longer of two [B, C, T, F] tensors around its centre (metrics.py:32-49; unused there)."""
        d = x.size(2) - y.size(2)
        if d == 0:
            return x, y
        assert abs(d) < 10, "Error: the offset %s is too large, check the code please" % (abs(d))
        a = abs(d) // 2
        b = abs(d) - a
        if d > 0:
            return x[:, :, a:x.size(2) - b, :], y
        return x, y[:, :, a:y.size(2) - b, :]
