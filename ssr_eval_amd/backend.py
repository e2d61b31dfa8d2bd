"""Host-side plumbing between the reference-shaped Python API and libssrhip.so.

PyTorch-ROCm is used for device memory, streams and (in ``ssr_eval_amd.dist``) RCCL; every number is
produced by the HIP kernels behind the C ABI.  Batches are ragged: one flat float32 device buffer plus
int64 offsets / int32 lengths, all resident in HBM.
"""
import ctypes as C
import os
import math
import threading

import numpy as np
import torch

from . import _lib
from ._lib import M_ALL, M_LOG_SISPEC, M_LSD, M_SISPEC, M_SSIM, SsrHipError  # noqa: F401

_PREC = {"f32": _lib.SSR_F32, "f64": _lib.SSR_F64, _lib.SSR_F32: _lib.SSR_F32, _lib.SSR_F64: _lib.SSR_F64}


def require_gpu():
    if not torch.cuda.is_available():
        raise SsrHipError("ssr_eval_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")


def default_device():
    require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def num_frames(n, n_fft, hop):
    """T of a centred STFT (bit-exact integer)."""
    return 1 + (int(n) + 2 * (n_fft // 2) - n_fft) // hop


def _tl_dft_matrices(n):
    """torchlibrosa DFTBase.dft_matrix / idft_matrix (stft.py): W[x, y] = omega ** (x y) with omega = exp(-/+ 2 pi i / n), evaluated
    as the module evaluates it - numpy's complex128 power on the integer product grid - so that the float32 weights below are the
    reference's to the last bit (ssr_eval/dsp.py:21-39 builds STFT / ISTFT, which build these).  The power is a function of the
    exponent alone, so it is taken once per DISTINCT product x y (a quarter of the grid) and gathered: the same values, 3-4x sooner."""
    x, y = np.meshgrid(np.arange(n), np.arange(n))
    k, inv = np.unique(x * y, return_inverse=True)
    inv = inv.reshape(n, n)
    return np.power(np.exp(-2 * np.pi * 1j / n), k)[inv], np.power(np.exp(2 * np.pi * 1j / n), k)[inv]


def tl_conv_weights(n_fft, window=None):
    """The float32 Conv1d weights of torchlibrosa's STFT / ISTFT for a float64 window [n_fft] (None: periodic Hann =
    librosa.filters.get_window("hann", n_fft, fftbins=True)), in the modules' layout:
    (fwd_re, fwd_im [n_bins, n_fft]; inv_re, inv_im [n_fft (output sample), n_fft (channel)]; float32(window ** 2))."""
    n, F = int(n_fft), int(n_fft) // 2 + 1
    W, Wi = _tl_dft_matrices(n)
    if window is None:
        import scipy.signal
        window = scipy.signal.get_window("hann", n, fftbins=True)
    win = np.asarray(window, dtype=np.float64)
    fw = W[:, :F] * win[:, None]                      # STFT.__init__: conv weights = (W[:, 0:out_channels] * window[:, None]).T
    iw = (Wi / n) * win[None, :]                      # ISTFT.init_real_imag_conv: (W / n_fft * ifft_window[None, :]).T
    c32 = lambda a: np.ascontiguousarray(a).astype(np.float32)          # noqa: E731
    return (c32(np.real(fw).T), c32(np.imag(fw).T), c32(np.real(iw).T), c32(np.imag(iw).T), (win ** 2).astype(np.float32))


class Plan:
    """An immutable ssr_plan (window / twiddle / Bluestein tables in HBM) for one (n_fft, hop, precision)."""

    def __init__(self, n_fft, hop, precision="f64", device=None):
        require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device) if device is not None else default_device()
        self.n_fft, self.hop, self.n_bins = int(n_fft), int(hop), int(n_fft) // 2 + 1
        self.precision = _PREC[precision]
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ssr_plan_create(self.n_fft, self.hop, self.precision, C.byref(h)))
        self.handle = h
        q = [C.c_int() for _ in range(6)]
        _lib.check(self.lib.ssr_plan_query(h, *[C.byref(x) for x in q]))
        self.fft_len, self.bluestein = q[3].value, bool(q[4].value)

    def frames(self, n):
        return num_frames(n, self.n_fft, self.hop)

    def check_lengths(self, lens_host):
        """torch-style reflect padding (torchlibrosa STFT / ISTFT: F.pad refuses a pad >= the signal length)."""
        if len(lens_host) and int(np.min(lens_host)) <= self.n_fft // 2:
            raise ValueError("reflect padding needs every signal longer than n_fft//2 = %d samples" % (self.n_fft // 2))

    def set_lowpass_engine(self, engine):
        """"segments" (frame kernel + overlap-add kernel through the workspace), "fused" (one kernel, no workspace traffic;
        float64 2048-point plans with 228 <= hop <= 914) or "conv" (torchlibrosa's dense float32 DFT products on the matrix cores:
        the reference's arithmetic class).  See include/ssr_hip.h: ssr_plan_set_lowpass_engine.  A plan handed out by
        get_plan() is cached under its engine and refuses a change."""
        if getattr(self, "_cached", False) and engine != getattr(self, "lowpass_engine", "segments"):
            raise ValueError("this plan is shared through get_plan() under its engine; ask get_plan(..., lowpass_engine=%r)" % engine)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ssr_plan_set_lowpass_engine(self.handle, _ENGINES[engine]))
            if engine == "conv":
                self._upload_tl_weights(None)
        self.lowpass_engine = engine
        return self

    def _upload_tl_weights(self, window):
        """The conv engine multiplies by torchlibrosa's OWN weight values (tl_conv_weights: the module's numpy expressions), not by
        the library's independently computed tables (1 ulp apart in ~0.5 % of the entries)."""
        tabs = tl_conv_weights(self.n_fft, window)
        _lib.check(self.lib.ssr_plan_set_tl_weights(self.handle, *[t.ctypes.data_as(C.c_void_p) for t in tabs]))

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle.value:
                self.lib.ssr_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class PlanEx(Plan):
    """ssr_plan_create_ex: FDomainHelper(center=, pad_mode=, window=) beyond the defaults (ssr_eval/dsp.py:7-59) - the conv engine
    with torchlibrosa's weights built from `window` (float64 [n_fft], librosa.filters.get_window(name, n_fft, fftbins=True); None =
    periodic Hann), no padding when center is False, zeros instead of the reflection for pad_mode "constant".  Serves stft(kind=
    "complex"), istft and LowpassBatch / lowpass only."""

    _PAD = {"reflect": _lib.PAD_REFLECT, "constant": _lib.PAD_CONSTANT}

    def __init__(self, n_fft, hop, window=None, center=True, pad_mode="reflect", device=None):
        require_gpu()
        if pad_mode not in self._PAD:
            raise NotImplementedError("pad_mode %r: torch's F.pad modes 'reflect' and 'constant' are implemented" % (pad_mode,))
        self.lib = _lib.load()
        self.device = torch.device(device) if device is not None else default_device()
        self.n_fft, self.hop, self.n_bins = int(n_fft), int(hop), int(n_fft) // 2 + 1
        self.precision = _PREC["f32"]
        self.center, self.pad_mode = bool(center), pad_mode
        w = None if window is None else np.ascontiguousarray(window, dtype=np.float64)
        if w is not None and w.shape != (self.n_fft,):
            raise ValueError("window must hold n_fft values")
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ssr_plan_create_ex(self.n_fft, self.hop, None if w is None else w.ctypes.data, int(self.center),
                                                   self._PAD[pad_mode], C.byref(h)))
            self.handle = h
            self._upload_tl_weights(w)
        self.fft_len, self.bluestein = 0, False
        self.lowpass_engine = "conv"

    def frames(self, n):
        pad = self.n_fft // 2 if self.center else 0
        return 0 if int(n) + 2 * pad < self.n_fft else 1 + (int(n) + 2 * pad - self.n_fft) // self.hop

    def check_lengths(self, lens_host):
        if not len(lens_host):
            return
        lo = int(np.min(lens_host))
        if self.center and self.pad_mode == "reflect" and lo <= self.n_fft // 2:
            raise ValueError("reflect padding needs every signal longer than n_fft//2 = %d samples" % (self.n_fft // 2))
        if lo < 1 or (not self.center and lo < self.n_fft):
            raise ValueError("every signal needs one whole frame (n_fft = %d samples without centring)" % self.n_fft)

    def set_lowpass_engine(self, engine):
        if engine != "conv":
            raise ValueError("a PlanEx has the conv engine only")
        return self


_ENGINES = {"segments": _lib.LOWPASS_SEGMENTS, "fused": _lib.LOWPASS_FUSED, "conv": _lib.LOWPASS_CONV}
_plans = {}
_plans_lock = threading.Lock()


_plans_ex = {}


def get_plan_ex(n_fft, hop, window_name, window, center, pad_mode, device=None):
    """The cached PlanEx of (n_fft, hop, window name, center, pad_mode, device); `window` = the float64 array of that name."""
    dev = torch.device(device) if device is not None else default_device()
    key = (int(n_fft), int(hop), window_name, bool(center), pad_mode, dev.index if dev.index is not None else torch.cuda.current_device())
    with _plans_lock:
        p = _plans_ex.get(key)
        if p is None:
            p = _plans_ex[key] = PlanEx(n_fft, hop, window, center, pad_mode, dev)
            p._cached = True
        return p


def get_plan(n_fft, hop, precision="f64", device=None, lowpass_engine="segments"):
    """The cached plan of (n_fft, hop, precision, device, low-pass engine)."""
    dev = torch.device(device) if device is not None else default_device()
    key = (int(n_fft), int(hop), _PREC[precision], dev.index if dev.index is not None else torch.cuda.current_device(), lowpass_engine)
    with _plans_lock:
        p = _plans.get(key)
        if p is None:
            p = Plan(n_fft, hop, precision, dev)
            if lowpass_engine != "segments":
                p.set_lowpass_engine(lowpass_engine)
            p.lowpass_engine = lowpass_engine
            p._cached = True
            _plans[key] = p
        return p


class _DescRing:
    """Page-locked staging for the descriptor uploads (offsets, lengths, cut bins: a few KB per launch sequence): ONE 4 MB block used
    as a ring, so that an upload costs a memcpy and an asynchronous copy.  (tensor.pin_memory() per upload looked free and was not:
    while launches are queued ahead of the GPU every earlier staging block is still waiting for its copy, the caching host allocator
    has nothing to hand back and goes to hipHostMalloc - ~1 ms each, visible as idle gaps in front of every stage in the kernel
    trace.)  The ring has two halves; on entering a half the host waits for the events recorded when it last left it (one per stream
    that copied out of it) - hundreds of launch sequences earlier."""
    SIZE = 1 << 22
    _by_dev, _lock = {}, threading.Lock()

    def __init__(self):
        self.buf = torch.empty(self.SIZE, dtype=torch.uint8, pin_memory=True)
        self.host = self.buf.numpy()
        self.pos, self.half = 0, 0
        self.streams = {}                       # streams that copied out of the current half
        self.guard = [[], []]                   # events to wait for before a half is written again
        self.lock = threading.Lock()

    @classmethod
    def get(cls, idx):
        with cls._lock:
            r = cls._by_dev.get(idx)
            if r is None:
                r = cls._by_dev[idx] = cls()
            return r

    def put(self, a, dev):
        nbytes = a.nbytes
        room = (nbytes + 63) & ~63
        with self.lock:
            pos = self.pos
            if pos + room > (self.half + 1) * (self.SIZE // 2):          # leave this half: remember what must finish first
                evs = []
                for st in self.streams.values():
                    ev = torch.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
                self.guard[self.half], self.streams = evs, {}
                self.half ^= 1
                pos = self.half * (self.SIZE // 2)
                for ev in self.guard[self.half]:
                    ev.synchronize()
                self.guard[self.half] = []
            self.pos = pos + room
            st = torch.cuda.current_stream(dev)
            self.streams[st.cuda_stream] = st
            self.host[pos:pos + nbytes] = a.reshape(-1).view(np.uint8)
            src = self.buf[pos:pos + nbytes].view(_TORCH_OF[a.dtype.type]).view(a.shape)
            out = torch.empty(a.shape, dtype=src.dtype, device=dev)
            out.copy_(src, non_blocking=True)
            return out


_TORCH_OF = {np.int32: torch.int32, np.int64: torch.int64, np.float32: torch.float32, np.float64: torch.float64,
             np.int16: torch.int16, np.uint8: torch.uint8}


def _h2d(a, dev):
    """Host ndarray -> device tensor WITHOUT a stream synchronisation: through page-locked memory (_DescRing) and an asynchronous
    copy.  (A pageable source makes the runtime order the copy behind everything queued on the stream and makes the host wait for
    it - every descriptor upload was a full GPU drain: rocprofv3 showed an evaluate() pass with 42 ms of kernels in 65 ms of wall
    clock.)"""
    a = np.ascontiguousarray(a)
    dev = torch.device(dev)
    if dev.type != "cuda" or a.size == 0 or a.nbytes > _DescRing.SIZE // 8 or a.dtype.type not in _TORCH_OF:
        return torch.from_numpy(a).to(dev)
    with torch.cuda.device(dev):
        return _DescRing.get(torch.cuda.current_device()).put(a, dev)


def _is_f64(a):
    return (a.dtype == torch.float64) if isinstance(a, torch.Tensor) else (getattr(a, "dtype", None) == np.float64)


class Ragged:
    """A ragged batch of 1-D signals resident on the device: float32, or float64 where the reference would hold
    float64 (IIR-degraded / resampled-from-float64 estimates)."""

    def __init__(self, data, off, lens_dev, lens_host):
        self.data, self.off, self.len = data, off, lens_dev
        self.lens_host = np.asarray(lens_host, dtype=np.int64)
        self.n = len(self.lens_host)
        self.max_len = int(self.lens_host.max()) if self.n else 0
        self.device = data.device
        self.packed = True                      # the signals sit back to back in `data` (from_list(allow_gaps=True) may say otherwise)

    @staticmethod
    def from_list(arrays, device=None, dtype=torch.float32, allow_gaps=False):
        """allow_gaps (callers that only hand data / off / len to a kernel: the metric stage): views into ONE device buffer in
        increasing address order are taken where they lie even with unused samples between them - `off` says where each starts
        (est[:m] / target[:m] of metrics.py:89-90 cut a sample off an item here and there, which used to send the whole batch
        through a per-signal `.to()` and a concatenation).  Such a batch is not `packed`: split() refuses it."""
        dev = torch.device(device) if device is not None else default_device()
        arrays = list(arrays)
        if allow_gaps and len(arrays) > 1:
            g = Ragged._from_views_with_gaps(arrays, dev, dtype)
            if g is not None:
                return g
        # Views that already sit back to back in ONE device buffer (the output of an earlier launch: upload_decoded,
        # resample_sinc, fft_lowpass, resample_poly ...) are taken as they are: no per-signal transfer call, no concatenation.
        if len(arrays) > 1 and all(isinstance(a, torch.Tensor) and a.is_cuda and a.dtype == dtype and a.dim() == 1 and a.is_contiguous()
                                   for a in arrays):
            # (address adjacency is NOT enough - two separate allocations are routinely back to back in the caching
            # allocator, and set_() past the end of a0's storage silently reallocates it: every view must live in a0's storage
            # and the whole run must fit inside it.)
            a0 = arrays[0]
            es, ok, end, st0 = a0.element_size(), True, a0.data_ptr(), a0.untyped_storage().data_ptr()
            for a in arrays:
                ok = ok and a.data_ptr() == end and a.device == a0.device and a.untyped_storage().data_ptr() == st0
                end += a.numel() * es
            ok = ok and end <= st0 + a0.untyped_storage().nbytes()
            if ok and arrays[0].device.index == (dev.index if dev.index is not None else torch.cuda.current_device()):
                lens = np.array([a.shape[0] for a in arrays], dtype=np.int64)
                total = int(lens.sum())
                if 0 < total and lens.max() < 2 ** 31:
                    a0 = arrays[0]
                    data = torch.empty(0, dtype=dtype, device=a0.device).set_(a0.untyped_storage(), a0.storage_offset(), (total,), (1,))
                    off = np.concatenate(([0], np.cumsum(lens)[:-1]))
                    desc = _h2d(off.astype(np.int64), a0.device)
                    return Ragged(data, desc, _h2d(lens.astype(np.int32), a0.device), lens)
        ts = []
        for a in arrays:
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
            if t.dim() != 1:
                raise ValueError("expected 1-D signals, got shape %s" % (tuple(t.shape),))
            ts.append(t.to(device=dev, dtype=dtype, non_blocking=True))
        lens = np.array([t.shape[0] for t in ts], dtype=np.int64)
        if len(ts) and lens.max() >= 2 ** 31:
            raise ValueError("signal too long")
        data = torch.cat(ts) if len(ts) else torch.empty(0, dtype=dtype, device=dev)
        off = np.concatenate(([0], np.cumsum(lens)[:-1])) if len(ts) else np.zeros(0, np.int64)
        return Ragged(data, _h2d(off.astype(np.int64), dev), _h2d(lens.astype(np.int32), dev), lens)

    @staticmethod
    def _from_views_with_gaps(arrays, dev, dtype):
        a0 = arrays[0]
        if not (isinstance(a0, torch.Tensor) and a0.is_cuda):
            return None
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        es, st = a0.element_size(), a0.untyped_storage()
        st0, p0, end = st.data_ptr(), a0.data_ptr(), a0.data_ptr()
        off, lens = [], []
        for a in arrays:
            if not (isinstance(a, torch.Tensor) and a.dtype == dtype and a.dim() == 1 and a.is_contiguous() and a.device == a0.device
                    and a.untyped_storage().data_ptr() == st0):
                return None
            p = a.data_ptr()
            if p < end:                                  # out of order or overlapping: not this path
                return None
            off.append((p - p0) // es); lens.append(a.shape[0])
            end = p + a.shape[0] * es
        lens = np.array(lens, dtype=np.int64)
        if a0.device.index != idx or end > st0 + st.nbytes() or lens.sum() == 0 or lens.max() >= 2 ** 31:
            return None
        data = torch.empty(0, dtype=dtype, device=a0.device).set_(st, a0.storage_offset(), ((end - p0) // es,), (1,))
        r = Ragged(data, _h2d(np.array(off, dtype=np.int64), a0.device), _h2d(lens.astype(np.int32), a0.device), lens)
        r.packed = bool(off[-1] + lens[-1] == lens.sum())
        return r

    @staticmethod
    def from_list_keep64(arrays, device=None, allow_gaps=False):
        """float64 batch if ANY signal is float64 (exact for the float32 ones), else float32."""
        arrays = list(arrays)
        return Ragged.from_list(arrays, device, torch.float64 if any(_is_f64(a) for a in arrays) else torch.float32, allow_gaps)

    @staticmethod
    def from_uniform(x):
        """x: [N, n] float32 contiguous device tensor - no copy, descriptors built on the device."""
        if x.dim() != 2 or x.dtype != torch.float32 or not x.is_contiguous() or not x.is_cuda:
            raise ValueError("from_uniform needs a contiguous float32 [N, n] device tensor")
        N, n = x.shape
        off = torch.arange(N, device=x.device, dtype=torch.int64) * n
        lens = torch.full((N,), n, device=x.device, dtype=torch.int32)
        return Ragged(x.view(-1), off, lens, np.full(N, n, dtype=np.int64))

    def split(self, flat=None):
        if not self.packed:
            raise ValueError("a batch of views with gaps has no packed layout to split")
        flat = self.data if flat is None else flat
        o = np.concatenate(([0], np.cumsum(self.lens_host)))
        return [flat[o[i]:o[i + 1]] for i in range(self.n)]


class _Rows:
    """Row (frame) descriptors of a ragged batch under a plan."""

    def __init__(self, plan, lens_host, device):
        self.T = np.array([plan.frames(n) for n in lens_host], dtype=np.int64)
        self.total = int(self.T.sum())
        self.max_T = int(self.T.max()) if len(self.T) else 0
        off = np.concatenate(([0], np.cumsum(self.T)[:-1])) if len(self.T) else np.zeros(0, np.int64)
        self.off_host = off
        self.off = _h2d(off.astype(np.int64), device)


def _check_reflect(plan, lens_host):
    plan.check_lengths(lens_host)


def _check_nonempty(lens_host):
    """librosa-style reflect padding (numpy.pad) repeats the reflection for short signals; only empty ones fail."""
    if len(lens_host) and int(np.min(lens_host)) < 1:
        raise ValueError("empty signal")


# ------------------------------------------------------------------------------------------------------
class PairBatch:
    """(est, target) ragged batch + cached descriptors/workspace for repeated ssr_pair_metrics calls."""

    def __init__(self, plan, est, tgt):
        if est.n != tgt.n or not np.array_equal(est.lens_host, tgt.lens_host):
            raise ValueError("est and target must have identical lengths (truncate to min_len first)")
        _check_nonempty(est.lens_host)
        if tgt.data.dtype == torch.float64 and est.data.dtype != torch.float64:
            # float32 estimate against a float64 target: widening the estimate is exact, and the reference promotes
            # every mixed operation to float64 anyway
            est = Ragged(est.data.to(torch.float64), est.off, est.len, est.lens_host)
            self.est = est
        self.plan, self.est, self.tgt = plan, est, tgt
        self.rows = _Rows(plan, est.lens_host, est.device)
        # the workspace is allocated on first use for the mask at hand: without SSIM no magnitude image is materialised
        # (8 bytes per bin of the batch otherwise), and it only grows when a later call asks for more
        self.ws_bytes, self.ws, self._ws_mask = 0, None, 0
        self.out = torch.empty((est.n, 4), dtype=torch.float64, device=est.device)

    def _workspace(self, mask):
        if self.ws is None or (mask & ~self._ws_mask):
            want = mask | self._ws_mask
            p, e = self.plan, self.est
            self.ws_bytes = int(p.lib.ssr_pair_metrics_workspace_bytes_for(p.handle, e.n, e.max_len, self.rows.total, want))
            self.ws = None                                          # release before the larger allocation
            self.ws = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=e.device)
            self._ws_mask = want

    def run(self, mask=M_ALL, stages=7):
        p, e, t = self.plan, self.est, self.tgt
        if e.n == 0:
            return self.out
        self._workspace(mask)
        if (mask & M_SSIM) and (self.rows.T.min() < 7 or p.n_bins < 7):
            raise ValueError("win_size exceeds image extent")  # what skimage raises for images smaller than 7x7
        if e.data.dtype == torch.float64:
            if stages != 7:
                raise ValueError("stage selection is a bench facility of the float32 path")
            fn = p.lib.ssr_pair_metrics_f64 if t.data.dtype == torch.float64 else p.lib.ssr_pair_metrics_est64
            _lib.check(fn(
                p.handle, _vp(e.data), _vp(e.off), _vp(t.data), _vp(t.off), _vp(e.len), _vp(self.rows.off), e.n,
                e.max_len, self.rows.total, mask, _vp(self.out), _vp(self.ws), self.ws_bytes, _stream()))
            return self.out
        _lib.check(p.lib.ssr_pair_metrics_stages(
            p.handle, _vp(e.data), _vp(e.off), _vp(t.data), _vp(t.off), _vp(e.len), _vp(self.rows.off), e.n, e.max_len,
            self.rows.total, mask, _vp(self.out), _vp(self.ws), self.ws_bytes, _stream(), stages))
        return self.out


class MultiPairBatch:
    """ONE target, K estimates per item (ssr_pair_metrics_multi): `est` holds n_keys * n items KEY-MAJOR (estimate k of item i
    at index k * n + i), every estimate of item i as long as target i.  The target is transformed once and its magnitude image
    stored once; estimates 1 .. K-1 are transformed two per complex transform.  out: [n, n_keys, 4] float64.
    A float64 `est` against a float32 target (every IIR key of a file, ssr_eval/eval.py:243-258): ssr_pair_metrics_multi_est64."""

    def __init__(self, plan, est, tgt, n_keys):
        n_keys = int(n_keys)
        if n_keys < 1 or est.n != tgt.n * n_keys or not np.array_equal(est.lens_host, np.tile(tgt.lens_host, n_keys)):
            raise ValueError("est must hold n_keys estimates per target, key-major, each as long as its target")
        if tgt.data.dtype != torch.float32 or est.data.dtype not in (torch.float32, torch.float64):
            raise ValueError("ssr_pair_metrics_multi takes float32 targets and float32 or float64 estimates (float64 targets go through PairBatch)")
        _check_nonempty(tgt.lens_host)
        self.plan, self.est, self.tgt, self.n_keys = plan, est, tgt, n_keys
        self.est64 = est.data.dtype == torch.float64
        self.rows = _Rows(plan, tgt.lens_host, tgt.device)
        self.ws_bytes, self.ws, self._ws_mask = 0, None, 0
        self.out = torch.empty((tgt.n, n_keys, 4), dtype=torch.float64, device=tgt.device)

    def _workspace(self, mask):
        if self.ws is None or (mask & ~self._ws_mask):
            want = mask | self._ws_mask
            p, t = self.plan, self.tgt
            size_fn = p.lib.ssr_pair_metrics_multi_est64_workspace_bytes if self.est64 else p.lib.ssr_pair_metrics_multi_workspace_bytes
            self.ws_bytes = int(size_fn(p.handle, t.n, self.n_keys, t.max_len, self.rows.total, want))
            self.ws = None
            self.ws = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=t.device)
            self._ws_mask = want

    def run(self, mask=M_ALL):
        p, e, t = self.plan, self.est, self.tgt
        if t.n == 0:
            return self.out
        self._workspace(mask)
        if (mask & M_SSIM) and (self.rows.T.min() < 7 or p.n_bins < 7):
            raise ValueError("win_size exceeds image extent")
        fn = p.lib.ssr_pair_metrics_multi_est64 if self.est64 else p.lib.ssr_pair_metrics_multi
        _lib.check(fn(
            p.handle, _vp(e.data), _vp(e.off), _vp(t.data), _vp(t.off), _vp(t.len), _vp(self.rows.off), t.n, self.n_keys, t.max_len,
            self.rows.total, mask, _vp(self.out), _vp(self.ws), self.ws_bytes, _stream()))
        return self.out


class Pending:
    """A device result on its way to the host: the copy into page-locked memory is queued behind the launch sequence that produces
    it, and calling the object waits for THAT copy only (an event) and returns the ndarray - the host can queue the next batch's
    launches in between.  The page-locked buffers come from a small free list: with launches queued ahead torch's caching
    host allocator has no block whose event has completed and falls back to hipHostMalloc (~1 ms per result, the gap _DescRing removed
    from the uploads); a buffer returns to the list when its result has been taken."""

    _free = {}                 # bytes (rounded up to 4 KiB) -> [page-locked uint8 tensors]
    _KEEP = 8                  # per size class

    @classmethod
    def _take(cls, nbytes):
        size = max(4096, (nbytes + 4095) & ~4095)
        lst = cls._free.setdefault(size, [])
        return size, (lst.pop() if lst else torch.empty(size, dtype=torch.uint8, pin_memory=True))

    def __init__(self, t):
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        self._size, self._buf = self._take(nbytes)
        self.host = self._buf[:nbytes].view(t.dtype).view(t.shape)
        self.host.copy_(t, non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record(torch.cuda.current_stream(t.device))
        self._value = None

    def __call__(self):
        if self._value is None:
            self.ev.synchronize()
            self._value = self.host.numpy().copy()          # the caller's array does not alias the recycled buffer
            lst = self._free.setdefault(self._size, [])
            if len(lst) < self._KEEP:
                lst.append(self._buf)
            self._buf = self.host = None
        return self._value


def pair_metrics_multi(plan, est_lists, tgt_list, mask=M_ALL, deferred=False):
    """est_lists: K lists (one per key) of n waveforms - all float32, or all float64 (the IIR keys of a file: they stay float64, as in the
    reference); tgt_list: n float32 targets -> [n, K, 4] float64 (deferred: a Pending)."""
    with torch.cuda.device(plan.device):
        flat = [e for key in est_lists for e in key]
        b = MultiPairBatch(plan, Ragged.from_list_keep64(flat, plan.device, allow_gaps=True), Ragged.from_list(tgt_list, plan.device, allow_gaps=True),
                           len(est_lists))
        return Pending(b.run(mask)) if deferred else b.run(mask).cpu().numpy()


def pair_metrics(plan, est_list, tgt_list, mask=M_ALL, deferred=False):
    """[n, 4] float64 (lsd, log_sispec, sispec, ssim) for lists of equal-length (est, target) waveforms.
    float64 signals stay float64 (ssr_pair_metrics_est64 / ssr_pair_metrics_f64).  deferred: a Pending instead of the ndarray."""
    with torch.cuda.device(plan.device):
        b = PairBatch(plan, Ragged.from_list_keep64(est_list, plan.device, allow_gaps=True),
                      Ragged.from_list_keep64(tgt_list, plan.device, allow_gaps=True))
        return Pending(b.run(mask)) if deferred else b.run(mask).cpu().numpy()


def stft(plan, wavs, kind="mag", torch_style_pad=False):
    """STFT of a list of waveforms.  kind "mag": list of [T, F] tensors; "complex": (re list, im list).
    torch_style_pad: refuse signals not longer than n_fft//2 the way torch's reflect padding does (torchlibrosa);
    otherwise short signals are reflect-padded repeatedly, as numpy.pad / librosa do."""
    with torch.cuda.device(plan.device):
        r = wavs if isinstance(wavs, Ragged) else Ragged.from_list(wavs, plan.device)
        if r.n == 0:
            return ([], []) if kind == "complex" else []
        if torch_style_pad:
            _check_reflect(plan, r.lens_host)
        _check_nonempty(r.lens_host)
        rows = _Rows(plan, r.lens_host, r.device)
        a = torch.empty((rows.total, plan.n_bins), dtype=torch.float32, device=r.device)
        b = torch.empty_like(a) if kind == "complex" else None
        _lib.check(plan.lib.ssr_stft(plan.handle, _vp(r.data), _vp(r.off), _vp(r.len), _vp(rows.off), r.n, r.max_len,
                                     _lib.STFT_COMPLEX if kind == "complex" else _lib.STFT_MAG, _vp(a), _vp(b), _stream()))
        cut = lambda m: [m[rows.off_host[i]:rows.off_host[i] + rows.T[i]] for i in range(r.n)]
        return (cut(a), cut(b)) if kind == "complex" else cut(a)


def magphase(re, im, eps):
    """(mag, cos, sin) of ssr_eval/dsp.py:76-81 for float32 device tensors of any shape."""
    re, im = re.contiguous(), im.contiguous()
    mag, cos, sin = torch.empty_like(re), torch.empty_like(re), torch.empty_like(re)
    with torch.cuda.device(re.device):
        _lib.check(_lib.load().ssr_magphase(_vp(re), _vp(im), re.numel(), float(eps), _vp(mag), _vp(cos), _vp(sin), _stream()))
    return mag, cos, sin


def _dev_f32(x, dev=None):
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    dev = dev if dev is not None else (t.device if t.is_cuda else default_device())
    return t.to(device=dev, dtype=torch.float32).contiguous()


def elementwise(op, x):
    """op "to_log": log10(x + 1e-12); "from_log": 10 ** min(x, 5) (ssr_eval/utils.py:43-50).  float32 device tensor."""
    require_gpu()
    t = _dev_f32(x)
    out = torch.empty_like(t)
    fn = {"to_log": _lib.load().ssr_to_log, "from_log": _lib.load().ssr_from_log}[op]
    with torch.cuda.device(t.device):
        _lib.check(fn(_vp(t), t.numel(), _vp(out), _stream()))
    return out


def energy_sums(a, b, n_items):
    """[n_items, 3] float64 device tensor {sum a^2, sum b^2, sum a*b} over n_items equal contiguous slices."""
    require_gpu()
    ta = _dev_f32(a)
    tb = _dev_f32(b, ta.device)
    if ta.shape != tb.shape or n_items <= 0 or ta.numel() % n_items:
        raise ValueError("energy_sums needs two tensors of one shape that split evenly into n_items slices")
    out = torch.empty((n_items, 3), dtype=torch.float64, device=ta.device)
    with torch.cuda.device(ta.device):
        _lib.check(_lib.load().ssr_energy_sums(_vp(ta), _vp(tb), n_items, ta.numel() // n_items, _vp(out), _stream()))
    return out


def scale_items(x, mul, div):
    """(x[i] * mul[i]) / div[i] per leading-dimension slice, float32 with two roundings (energy_unify)."""
    require_gpu()
    t = _dev_f32(x)
    m, d = _dev_f32(mul, t.device).reshape(-1), _dev_f32(div, t.device).reshape(-1)
    n_items = m.numel()
    if d.numel() != n_items or n_items == 0 or t.numel() % n_items:
        raise ValueError("one multiplier and one divisor per slice")
    out = torch.empty_like(t)
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().ssr_scale_items(_vp(t), _vp(m), _vp(d), n_items, t.numel() // n_items, _vp(out), _stream()))
    return out


def sispec_multichannel(est, target, log_domain):
    """SISpec (or its to_log variant) of [B, C, T, F] tensors with C > 1 -> 0-dim float64 device tensor (the batch mean)."""
    require_gpu()
    e = _dev_f32(est)
    t = _dev_f32(target, e.device)
    Bn, Cn = int(e.shape[0]), int(e.shape[1])
    per = e[0, 0].numel()
    out = torch.empty(Bn + 1, dtype=torch.float64, device=e.device)
    ws = torch.empty(Bn * Cn * 3, dtype=torch.float64, device=e.device)
    with torch.cuda.device(e.device):
        _lib.check(_lib.load().ssr_sispec_multichannel(_vp(e), _vp(t), Bn, Cn, per, 1 if log_domain else 0, _vp(out), _vp(ws),
                                                       ws.numel() * 8, _stream()))
    return out[Bn]


def spectrogram_metrics(est_sps, tgt_sps, mask=M_ALL):
    """Metrics on lists of [T_i, F] float32 magnitude spectrograms -> [n, 4] float64 device tensor."""
    require_gpu()
    lib = _lib.load()
    dev = est_sps[0].device if isinstance(est_sps[0], torch.Tensor) and est_sps[0].is_cuda else default_device()
    with torch.cuda.device(dev):
        to_dev = lambda s: (s if isinstance(s, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(s))).to(
            device=dev, dtype=torch.float32)
        es, ts = [to_dev(s) for s in est_sps], [to_dev(s) for s in tgt_sps]
        F = es[0].shape[1]
        for e, t in zip(es, ts):
            if e.shape != t.shape or e.dim() != 2 or e.shape[1] != F:
                raise ValueError("spectrogram shape mismatch")
        T = np.array([e.shape[0] for e in es], dtype=np.int64)
        if (mask & M_SSIM) and (T.min() < 7 or F < 7):
            raise ValueError("win_size exceeds image extent")
        x = torch.cat([e.reshape(-1) for e in es])
        y = torch.cat([t.reshape(-1) for t in ts])
        off = _h2d(np.concatenate(([0], np.cumsum(T)[:-1])).astype(np.int64), dev)
        rows = _h2d(T.astype(np.int32), dev)
        n, max_T = len(es), int(T.max())
        ws_bytes = int(lib.ssr_spectrogram_metrics_workspace_bytes(n, max_T, F))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        out = torch.empty((n, 4), dtype=torch.float64, device=dev)
        _lib.check(lib.ssr_spectrogram_metrics(_vp(x), _vp(y), _vp(off), _vp(rows), n, max_T, F, mask, _vp(out), _vp(ws),
                                               ws_bytes, _stream()))
        return out


class LowpassBatch:
    """A ragged batch + cached descriptors / workspace / output for repeated ssr_fft_lowpass calls (K6): one cut bin per
    item.  `run()` enqueues the launch sequence on the current stream and returns the flat output buffer (same layout as
    the input batch); `out_ragged()` views it as a Ragged batch for the metric stage."""

    def __init__(self, plan, ragged, cut_bins, out=None):
        _check_reflect(plan, ragged.lens_host)
        if ragged.data.dtype != torch.float32:
            raise ValueError("the STFT-domain low-pass takes float32 signals (torchlibrosa's convolution does too)")
        self.plan, self.r = plan, ragged
        self.rows = _Rows(plan, ragged.lens_host, ragged.device)
        self.set_cuts(cut_bins)
        self.ws_bytes = int(plan.lib.ssr_ola_workspace_bytes(plan.handle, self.rows.total))
        self.ws = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=ragged.device)
        # out: a caller-owned float32 buffer of the batch's size (e.g. one key's slice of a multi-key estimate buffer)
        if out is not None and (out.dtype != torch.float32 or out.numel() != ragged.data.numel() or not out.is_contiguous()):
            raise ValueError("out must be a contiguous float32 buffer of the batch's size")
        self.out = out if out is not None else torch.empty_like(ragged.data)

    def set_cuts(self, cut_bins):
        """One cut bin per item (a scalar: the same for every item).  A batch with ONE cut runs through ssr_fft_lowpass_multi (row
        tiles across item boundaries), per-item cuts through ssr_fft_lowpass."""
        cuts = np.asarray(cut_bins, dtype=np.int32).reshape(-1)
        if cuts.size == 1 and self.r.n != 1:
            cuts = np.repeat(cuts, self.r.n)
        if cuts.size != self.r.n:
            raise ValueError("one cut bin per item")
        self.uniform = int(cuts[0]) if cuts.size and bool((cuts == cuts[0]).all()) else None
        self.cut = _h2d(cuts, self.r.device)

    def run(self):
        p, r = self.plan, self.r
        if not r.n:
            return self.out
        if self.uniform is not None:
            # one cut for the whole batch: the launch's row tiles run across item boundaries (ssr_fft_lowpass_multi with one key)
            cuts = (C.c_int32 * 1)(self.uniform)
            _lib.check(p.lib.ssr_fft_lowpass_multi(p.handle, _vp(r.data), _vp(r.off), _vp(r.len), cuts, 1, _vp(self.rows.off), r.n,
                                                   r.max_len, self.rows.total, _vp(self.out), 0, _vp(self.ws), self.ws_bytes, _stream()))
        else:
            _lib.check(p.lib.ssr_fft_lowpass(p.handle, _vp(r.data), _vp(r.off), _vp(r.len), _vp(self.cut), _vp(self.rows.off),
                                             r.n, r.max_len, self.rows.total, _vp(self.out), _vp(self.ws), self.ws_bytes,
                                             _stream()))
        return self.out

    def out_ragged(self):
        return Ragged(self.out, self.r.off, self.r.len, self.r.lens_host)


class MultiLowpassBatch:
    """One ragged batch, K cut bins applied to every item (SSR_Eval_Helper.lowpass_stft_hard's loop over setting_fft,
    ssr_eval/eval.py:401-410): ssr_fft_lowpass_multi.  `run()` returns the [K, batch samples] float32 output (key-major; row k has
    the input batch's layout); `out_ragged(k)` views key k as a Ragged batch for the metric stage.  On the conv engine the padded
    copy and the forward product are computed once for the K keys."""

    def __init__(self, plan, ragged, cut_bins, out=None):
        _check_reflect(plan, ragged.lens_host)
        if ragged.data.dtype != torch.float32:
            raise ValueError("the STFT-domain low-pass takes float32 signals (torchlibrosa's convolution does too)")
        self.plan, self.r = plan, ragged
        self.rows = _Rows(plan, ragged.lens_host, ragged.device)
        self.cuts = np.asarray(cut_bins, dtype=np.int32).reshape(-1)
        self.n_keys = int(self.cuts.size)
        self.ws_bytes = int(plan.lib.ssr_ola_workspace_bytes(plan.handle, self.rows.total))
        self.ws = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=ragged.device)
        total = ragged.data.numel()
        if out is not None and (out.dtype != torch.float32 or out.numel() != self.n_keys * total or not out.is_contiguous()):
            raise ValueError("out must be a contiguous float32 buffer of n_keys x the batch's size")
        self.out = (out if out is not None else torch.empty(self.n_keys * total, dtype=torch.float32, device=ragged.device)).view(self.n_keys, total)

    def run(self):
        p, r = self.plan, self.r
        if r.n and self.n_keys:
            _lib.check(p.lib.ssr_fft_lowpass_multi(p.handle, _vp(r.data), _vp(r.off), _vp(r.len), self.cuts.ctypes.data_as(C.c_void_p),
                                                   self.n_keys, _vp(self.rows.off), r.n, r.max_len, self.rows.total, _vp(self.out),
                                                   r.data.numel(), _vp(self.ws), self.ws_bytes, _stream()))
        return self.out

    def out_ragged(self, k):
        return Ragged(self.out[k], self.r.off, self.r.len, self.r.lens_host)


def fft_lowpass_multi(plan, wavs, cut_bins):
    """K hard low-passes of every waveform of a list: [[key 0's outputs], [key 1's], ...]."""
    with torch.cuda.device(plan.device):
        r = wavs if isinstance(wavs, Ragged) else Ragged.from_list(wavs, plan.device)
        if r.n == 0:
            return [[] for _ in cut_bins]
        b = MultiLowpassBatch(plan, r, cut_bins)
        out = b.run()
        return [r.split(out[k]) for k in range(b.n_keys)]


def fft_lowpass(plan, wavs, cut_bins):
    """STFT-domain hard low-pass (K6) of a list of waveforms; cut_bins: first zeroed bin per item."""
    with torch.cuda.device(plan.device):
        r = wavs if isinstance(wavs, Ragged) else Ragged.from_list(wavs, plan.device)
        if r.n == 0:
            return []
        b = LowpassBatch(plan, r, cut_bins)
        return r.split(b.run())


def istft(plan, res, ims, lengths):
    """Inverse STFT (K6') of lists of [T_i, F] float32 (re, im); lengths: output samples per item."""
    with torch.cuda.device(plan.device):
        dev = plan.device
        lens = np.asarray(lengths, dtype=np.int64)
        if len(lens) == 0:
            return []
        _check_reflect(plan, lens)
        rows = _Rows(plan, lens, dev)
        for r_, t_ in zip(res, rows.T):
            if r_.shape[0] != t_ or r_.shape[1] != plan.n_bins:
                raise ValueError("spectrogram rows do not match ssr_num_frames(length)")
        re = torch.cat([r_.to(device=dev, dtype=torch.float32).reshape(-1) for r_ in res])
        im = torch.cat([i_.to(device=dev, dtype=torch.float32).reshape(-1) for i_ in ims])
        out_off = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
        out = torch.empty(int(lens.sum()), dtype=torch.float32, device=dev)
        ws_bytes = int(plan.lib.ssr_ola_workspace_bytes(plan.handle, rows.total))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        # descriptor tensors must stay referenced until the launch has been enqueued: a temporary would be
        # returned to the caching allocator (and its block re-used by the next temporary) before the call
        lens_d = _h2d(lens.astype(np.int32), dev)
        out_off_d = _h2d(out_off, dev)
        _lib.check(plan.lib.ssr_istft(plan.handle, _vp(re), _vp(im), _vp(rows.off), _vp(lens_d), _vp(out_off_d), len(lens),
                                      int(lens.max()), rows.total, _vp(out), _vp(ws), ws_bytes, _stream()))
        return [out[out_off[i]:out_off[i] + lens[i]] for i in range(len(lens))]


# ------------------------------------------------------------------------------------------------------
class ResamplePlan:
    """Integer plan + taps of scipy.signal.resample_poly(x, up, down): float32 taps for a float32
    signal, the unrounded float64 design for a float64 one - as SciPy casts `h` to the signal's dtype."""

    _cache = {}

    def __init__(self, up, down, device):
        from scipy.signal import firwin  # tap design exactly as SciPy does it (host, once per rate pair)
        lib = _lib.load()
        u, d, n_out, hl, pp, pr = C.c_int(), C.c_int(), C.c_int64(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.ssr_resample_plan(0, int(up), int(down), C.byref(u), C.byref(d), C.byref(n_out), C.byref(hl),
                                         C.byref(pp), C.byref(pr)))
        self.up, self.down, self.half_len = u.value, d.value, hl.value
        self.n_pre_pad, self.n_pre_remove = pp.value, pr.value
        self.identity = self.up == 1 and self.down == 1
        if self.identity:                      # scipy returns x.copy() before designing any filter
            self.taps_host, self.taps, self.taps64 = None, None, None
            return
        h64 = firwin(2 * self.half_len + 1, 1.0 / max(self.up, self.down), window=("kaiser", 5.0))
        h = h64.astype(np.float32)
        h *= self.up                            # scipy: h = h.astype(x.dtype) first, then h * up
        self.taps_host = np.concatenate((np.zeros(self.n_pre_pad, np.float32), h))
        self.taps = torch.from_numpy(self.taps_host).to(device)
        self.taps64 = torch.from_numpy(np.concatenate((np.zeros(self.n_pre_pad), h64 * self.up))).to(device)

    @classmethod
    def get(cls, up, down, device):
        g = math.gcd(int(up), int(down))
        key = (int(up) // g, int(down) // g, str(device))
        p = cls._cache.get(key)
        if p is None:
            p = cls._cache[key] = cls(up, down, device)
        return p

    def n_out(self, n_in):
        return -(-int(n_in) * self.up // self.down)


class ResampleBatch:
    """A ragged batch + cached output descriptors / buffer for repeated ssr_resample_poly calls (K7)."""

    def __init__(self, ragged, up, down, exact=True, alloc=True):
        dev = ragged.device
        self.r, self.rp = ragged, ResamplePlan.get(up, down, dev)
        self.exact = exact
        self.f64 = ragged.data.dtype == torch.float64
        if self.rp.identity:
            self.out_len = ragged.lens_host.copy()
        else:
            self.out_len = np.array([self.rp.n_out(n) for n in ragged.lens_host], dtype=np.int64)
        self.out_off = np.concatenate(([0], np.cumsum(self.out_len)[:-1])).astype(np.int64) if ragged.n else np.zeros(0, np.int64)
        self.out_off_d = _h2d(self.out_off, dev)
        self.out_len_d = _h2d(self.out_len.astype(np.int32), dev)
        self.out = torch.empty(int(self.out_len.sum()), dtype=ragged.data.dtype, device=dev) if alloc else None

    def run(self):
        r, rp = self.r, self.rp
        if self.out is None:
            self.out = torch.empty(int(self.out_len.sum()), dtype=r.data.dtype, device=r.device)
        if rp.identity:
            self.out.copy_(r.data)             # scipy returns x.copy() before designing any filter
        elif r.n and self.out_len.max() > 0:
            taps = rp.taps64 if self.f64 else rp.taps
            lib = _lib.load()
            args = (_vp(r.data), _vp(r.off), _vp(r.len), _vp(self.out_off_d), _vp(self.out_len_d), r.n,
                    int(self.out_len.max()), rp.up, rp.down, _vp(taps), int(taps.numel()), rp.n_pre_remove, _vp(self.out), _stream())
            # exact=False: the matrix-core kernel (float32 fused multiply-adds: within ~1 ulp per tap of SciPy's sums, not its
            # bits) where the plan fits it; float64 signals and plans it does not hold run the bit-exact kernel
            rc = _lib.ERR_UNSUPPORTED if (self.f64 or self.exact) else lib.ssr_resample_poly_mfma(*args)
            if rc == _lib.ERR_UNSUPPORTED:
                rc = (lib.ssr_resample_poly_f64 if self.f64 else lib.ssr_resample_poly)(*args)
            _lib.check(rc)
        return self.out

    def out_ragged(self):
        return Ragged(self.out, self.out_off_d, self.out_len_d, self.out_len)


class ResampleChainBatch:
    """resample_poly(resample_poly(x, mid, orig), new, mid) for a ragged float32 (or float64: two calls) batch - BASELINE cfg-5's 16 kHz -> 44.1 kHz -> 48 kHz.
    Where the two plans fit ssr_resample_poly_chain (21-tap phases, 8 up1 = 24 down2: 441/160 then 160/147) ONE kernel runs both
    stages and the intermediate signal never leaves LDS; otherwise the two stages run through ssr_resample_poly.  Same bits either
    way (SciPy's).  `fused`: None = try the fused kernel, False = always two calls."""

    def __init__(self, ragged, sr_orig, sr_mid, sr_new, fused=None):
        self.r = ragged
        self.s1 = ResampleBatch(ragged, sr_mid, sr_orig, alloc=False)
        mid = Ragged(torch.empty(0, dtype=ragged.data.dtype, device=ragged.device), self.s1.out_off_d, self.s1.out_len_d,
                     self.s1.out_len)                                          # geometry only: the buffer may never exist
        self.s2 = ResampleBatch(mid, sr_new, sr_mid, alloc=False)
        self.out_len, self.out_off, self.out_off_d, self.out_len_d = self.s2.out_len, self.s2.out_off, self.s2.out_off_d, self.s2.out_len_d
        # (a float64 batch runs the two stages in float64 - what SciPy does per dtype - and so is its output: ADVICE r4)
        self.out = torch.empty(int(self.out_len.sum()), dtype=ragged.data.dtype, device=ragged.device)
        self.s2.out = self.out
        self.fused = fused
        self.ran_fused = None
        if ragged.data.dtype != torch.float32 or self.s1.rp.identity or self.s2.rp.identity:
            self.fused = False

    def run(self):
        r, p1, p2 = self.r, self.s1.rp, self.s2.rp
        if self.fused is not False and r.n and self.out_len.max() > 0:
            rc = _lib.load().ssr_resample_poly_chain(
                _vp(r.data), _vp(r.off), _vp(r.len), _vp(self.s1.out_len_d), _vp(self.out_off_d), _vp(self.out_len_d), r.n,
                int(self.out_len.max()), p1.up, p1.down, _vp(p1.taps), int(p1.taps.numel()), p1.n_pre_remove,
                p2.up, p2.down, _vp(p2.taps), int(p2.taps.numel()), p2.n_pre_remove, _vp(self.out), _stream())
            if rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc)
                self.ran_fused = True
                return self.out
            if self.fused:
                _lib.check(rc)
        self.ran_fused = False
        self.run_stage1()
        return self.run_stage2()

    def run_stage1(self):
        """Stage 1 alone, into the intermediate buffer (allocated on first use)."""
        return self.s1.run()

    def run_stage2(self):
        """Stage 2 alone, from the intermediate buffer run_stage1() filled."""
        self.s2.r = self.s1.out_ragged()
        return self.s2.run()

    def out_ragged(self):
        return Ragged(self.out, self.out_off_d, self.out_len_d, self.out_len)


def resample_poly_chain(wavs, sr_orig, sr_mid, sr_new, device=None, fused=None):
    """scipy.signal.resample_poly twice (sr_orig -> sr_mid -> sr_new) for a list of float32 waveforms; see ResampleChainBatch."""
    dev = torch.device(device) if device is not None else default_device()
    with torch.cuda.device(dev):
        r = wavs if isinstance(wavs, Ragged) else Ragged.from_list(wavs, dev)
        b = ResampleChainBatch(r, sr_orig, sr_mid, sr_new, fused=fused)
        out = b.run()
        return [out[b.out_off[i]:b.out_off[i] + b.out_len[i]] for i in range(r.n)]


def resample_poly(wavs, up, down, device=None, exact=True):
    """Polyphase resampling (K7) of a list of waveforms; bit-identical to scipy.signal.resample_poly.  A batch
    holding float64 signals is resampled in float64 (float64 taps and accumulation), everything else in float32.
    exact=False: float32 signals go through the matrix-core kernel (ssr_resample_poly_mfma: the same sums with fused
    multiply-adds, ~1 ulp per tap from SciPy's values, 1.2-1.5x the rate)."""
    dev = torch.device(device) if device is not None else default_device()
    with torch.cuda.device(dev):
        # views into ONE device buffer in increasing address order are read where they lie (the kernels take an offset per signal):
        # the IIR keys of a batch - [design][file] slices of ssr_sosfiltfilt_multi's output - used to be concatenated again here
        # (11 GB and 13 k per-signal calls per evaluate() batch of 256 files x 36 keys)
        r = wavs if isinstance(wavs, Ragged) else Ragged.from_list_keep64(wavs, dev, allow_gaps=True)
        if not r.packed and ResamplePlan.get(up, down, dev).identity:
            r = Ragged.from_list_keep64(wavs, dev)              # (the identity plan copies the packed buffer)
        b = ResampleBatch(r, up, down, exact=exact)
        out = b.run()
        return [out[b.out_off[i]:b.out_off[i] + b.out_len[i]] for i in range(r.n)]


# ------------------------------------------------------------------------------------------------------
_SINC_FILTERS = {  # resampy filter name -> (num_zeros, precision, Kaiser beta, roll-off)   (resampy/filters.py)
    "kaiser_best": (64, 9, 14.769656459379492, 0.9475937167399596),
    "kaiser_fast": (16, 9, 8.555504641634386, 0.85),
}


class SincPlan:
    """Interpolation tables of resampy.resample(x, sr_orig, sr_new, filter=name) on the device (N2): the right half of
    the Kaiser-windowed sinc, regenerated from the filter's documented parameters on the host in float64 (a plan, like
    the polyphase taps), its forward differences, and the running time register t_k = fl(t_{k-1} + 1/ratio)."""

    _cache = {}

    def __init__(self, sr_orig, sr_new, name, device):
        from scipy.signal.windows import kaiser
        if name not in _SINC_FILTERS:
            raise NotImplementedError("resampling filter %r (supported: %s)" % (name, sorted(_SINC_FILTERS)))
        nz, prec, beta, roll = _SINC_FILTERS[name]
        self.ratio = float(sr_new) / float(sr_orig)
        # filter phases repeat every `a` outputs when sr_new / sr_orig = a / b in lowest terms (a work-mapping hint)
        self.phase_period = 0
        if float(sr_new).is_integer() and float(sr_orig).is_integer():
            self.phase_period = int(sr_new) // math.gcd(int(sr_new), int(sr_orig))
        self.num_table = 2 ** prec
        n = self.num_table * nz
        win = kaiser(2 * n + 1, beta)[n:] * (roll * np.sinc(roll * np.linspace(0, nz, num=n + 1, endpoint=True)))
        if self.ratio < 1:
            win = win * self.ratio
        delta = np.zeros_like(win)
        delta[:-1] = np.diff(win)
        self.scale = min(1.0, self.ratio)
        self.index_step = int(self.scale * self.num_table)
        self.n_win = int(win.shape[0])
        self.device = device
        self.win = torch.from_numpy(win).to(device)
        self.delta = torch.from_numpy(delta).to(device)
        self._tr, self._tr_len = None, 0

    @classmethod
    def get(cls, sr_orig, sr_new, name, device):
        key = (float(sr_orig), float(sr_new), name, str(device))
        p = cls._cache.get(key)
        if p is None:
            p = cls._cache[key] = cls(sr_orig, sr_new, name, device)
        return p

    def n_out(self, n_in):
        return int(int(n_in) * self.ratio)                      # resampy: int(shape * sample_ratio)

    def time_register(self, n):
        """Device float64 [>= n]: resampy's `time_register += time_increment`, a SEQUENTIAL float64 sum (np.cumsum)."""
        if n > self._tr_len:
            m = max(n, 2 * self._tr_len, 1 << 16)
            tr = np.empty(m)
            tr[0] = 0.0
            tr[1:] = np.cumsum(np.full(m - 1, 1.0 / self.ratio))
            self._tr, self._tr_len = torch.from_numpy(tr).to(self.device), m
        return self._tr


def resample_sinc(wavs, sr_orig, sr_new, res_type="kaiser_best", device=None, fix=True):
    """librosa.resample(y, sr_orig, sr_new, res_type="kaiser_best" | "kaiser_fast") for a list of float32 waveforms (N2):
    resampy's band-limited interpolation on the GPU, then librosa's fix_length to ceil(n * ratio).  Device tensors out."""
    dev = torch.device(device) if device is not None else default_device()
    with torch.cuda.device(dev):
        r = wavs if isinstance(wavs, Ragged) else Ragged.from_list(wavs, dev)
        if float(sr_orig) == float(sr_new):
            return [w.clone() for w in r.split()]
        sp = SincPlan.get(sr_orig, sr_new, res_type, dev)
        out_len = np.array([sp.n_out(n) for n in r.lens_host], dtype=np.int64)
        want = np.array([int(np.ceil(n * sp.ratio)) for n in r.lens_host], dtype=np.int64) if fix else out_len
        out_off = np.concatenate(([0], np.cumsum(want)[:-1])).astype(np.int64) if r.n else np.zeros(0, np.int64)
        out = torch.zeros(int(want.sum()), dtype=torch.float32, device=dev)       # fix_length pads with zeros
        if r.n and out_len.max() > 0:
            out_off_d = _h2d(out_off, dev)
            out_len_d = _h2d(out_len.astype(np.int32), dev)
            tr = sp.time_register(int(out_len.max()))
            _lib.check(_lib.load().ssr_resample_sinc(_vp(r.data), _vp(r.off), _vp(r.len), _vp(out_off_d), _vp(out_len_d), r.n,
                                                     int(out_len.max()), _vp(tr), int(tr.numel()), _vp(sp.win), _vp(sp.delta),
                                                     sp.n_win, sp.num_table, sp.index_step, float(sp.scale), float(sp.ratio),
                                                     int(sp.phase_period), _vp(out), _stream()))
        return [out[out_off[i]:out_off[i] + want[i]] for i in range(r.n)]


class _Staging:
    """Two page-locked int16 arenas per device, used alternately: a batch's PCM is packed into one of them and crosses PCIe in
    ONE asynchronous copy while the host packs the next batch into the other.  An arena is reused only after the copy that
    last read it has completed (event)."""
    _by_dev = {}
    _by_dev_lock = threading.Lock()

    def __init__(self):
        self.buf = [None, None]
        self.ev = [None, None]
        self.twin = [None, None]                # the arenas' device-side twins and the events behind their last readers (_h2d_arena)
        self.done = [None, None]
        self.k = 0
        self.lock = threading.Lock()

    @classmethod
    def get(cls, dev, role="packed"):
        """One arena pair per (resolved device index, role).  role "packed": io.PackedBatch (the prefetching file path);
        "list": upload_decoded() of a list of RawAudio - its own pair, so that a list upload in the middle of an in-flight
        PackedBatch (its non-RIFF `others`) can never flip that batch's arenas."""
        d = torch.device(dev)
        idx = d.index if d.index is not None else torch.cuda.current_device()
        with cls._by_dev_lock:
            return cls._by_dev.setdefault((idx, role), cls())

    def arena(self, n):
        with self.lock:
            self.k ^= 1
            k = self.k
            ev = self.ev[k]
        if ev is not None:
            ev.synchronize()
        with self.lock:
            if self.buf[k] is None or self.buf[k].numel() < n:
                self.buf[k] = torch.empty(max(n, 1 << 22), dtype=torch.int16, pin_memory=True)
            return k, self.buf[k]

    def sent(self, k):
        ev = torch.cuda.Event()
        ev.record()
        with self.lock:
            self.ev[k] = ev


_upload_streams = {}


def _h2d_arena(st, k, arena, total, dev):
    """Page-locked int16 arena k of staging pair `st` -> its device twin, on the device's UPLOAD stream: the bus transfer of a batch
    runs under the kernels of the batch before it instead of queueing behind them (6 ms of an evaluate() pass over 367 files).  The
    twin is a persistent buffer (grown when a batch needs more), so no allocation crosses the two streams' pools.  Ordering, all on
    the GPU: the upload stream waits for the event recorded behind the last kernel that read the twin (consumed()), the current
    stream waits for the copy.  -> (the twin's first `total` elements, consumed)."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    side = _upload_streams.get(idx)
    if side is None:
        side = _upload_streams[idx] = torch.cuda.Stream(device=idx)
    cur = torch.cuda.current_stream(idx)
    with st.lock:
        twin, done = st.twin[k], st.done[k]
    if twin is None or twin.numel() < total or twin.device.index != idx:
        twin = torch.empty(max(total, 1 << 22), dtype=torch.int16, device=dev)       # (its old self is freed stream-ordered, on `cur`)
        side.wait_stream(cur)
    if done is not None:
        side.wait_event(done)
    with torch.cuda.stream(side):
        twin[:total].copy_(arena[:total], non_blocking=True)
        st.sent(k)
    cur.wait_stream(side)

    def consumed():
        ev = torch.cuda.Event()
        ev.record(cur)
        with st.lock:
            st.twin[k], st.done[k] = twin, ev
    return twin[:total], consumed


def _pcm_to_float(d16, in_off, frames, chans, dev):
    """int16 device buffer + host descriptors -> (flat float32 mono device buffer, host out offsets)."""
    n = len(frames)
    out_off = np.concatenate(([0], np.cumsum(frames)[:-1]))
    desc = _h2d(np.concatenate((in_off, out_off)).astype(np.int64), dev)
    desc32 = _h2d(np.concatenate((frames.astype(np.int32), chans.astype(np.int32))), dev)
    flat = torch.empty(int(frames.sum()), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().ssr_pcm16_to_float(_vp(d16), _vp(desc[:n]), _vp(desc32[:n]), _vp(desc32[n:]), n, int(frames.max()),
                                              _vp(flat), _vp(desc[n:]), _stream()))
    return flat, out_off


def upload_decoded(raw, device=None):
    """Decoded files -> float32 mono device tensors, one per file.  `raw`: an io.PackedBatch (the files' PCM already sits in a
    page-locked arena) or a list of io.RawAudio.  16-bit PCM crosses the bus as int16 - half the bytes, no float pass on the
    host - in one pinned asynchronous copy per batch and is converted / mixed to mono on the GPU (ssr_pcm16_to_float);
    anything else is uploaded as the float32 mono array it already is."""
    from .io import PackedBatch
    if isinstance(raw, PackedBatch):
        dev = torch.device(raw.device)
        out = [None] * len(raw.paths)
        with torch.cuda.device(dev):
            if raw.total:                      # the arena's copy (and its event) first: nothing else touches this batch's arena
                d16, consumed = _h2d_arena(raw.staging, raw.k, raw.arena, raw.total, dev)
                flat, out_off = _pcm_to_float(d16, raw.in_off, raw.frames, raw.chans, dev)
                consumed()
                for j, i in enumerate(raw.pcm_idx):
                    out[i] = flat[out_off[j]:out_off[j] + raw.frames[j]]
            for i, r in zip(raw.other_idx, raw.others):           # (these go through the "list" arenas, a separate pair)
                out[i] = upload_decoded([r], dev)[0]
        return out
    dev = torch.device(device) if device is not None else default_device()
    out = [None] * len(raw)
    with torch.cuda.device(dev):
        pcm = [i for i, r in enumerate(raw) if r.pcm is not None and r.pcm.shape[0] > 0]
        is_pcm = set(pcm)
        for i, r in enumerate(raw):
            if i not in is_pcm:
                out[i] = torch.from_numpy(np.ascontiguousarray(r.to_float(), dtype=np.float32)).to(dev, non_blocking=True)
        if pcm:
            sizes = np.array([raw[i].pcm.shape[0] for i in pcm], dtype=np.int64)
            frames = np.array([raw[i].n_frames for i in pcm], dtype=np.int64)
            chans = np.array([raw[i].nch for i in pcm], dtype=np.int32)
            total = int(sizes.sum())
            st = _Staging.get(dev, "list")
            k, arena = st.arena(total)
            host = arena.numpy()
            in_off = np.concatenate(([0], np.cumsum(sizes)[:-1]))
            for i, o in zip(pcm, in_off):
                host[o:o + raw[i].pcm.shape[0]] = raw[i].pcm
            d16, consumed = _h2d_arena(st, k, arena, total, dev)
            flat, out_off = _pcm_to_float(d16, in_off, frames, chans, dev)
            consumed()
            for j, i in enumerate(pcm):
                out[i] = flat[out_off[j]:out_off[j] + frames[j]]
    return out


def sosfiltfilt(sos, wavs, device=None):
    """scipy.signal.sosfiltfilt(sos, x) for a list of float32 (or float64) waveforms on the GPU (N1); float64 tensors out.
    The section coefficients and sosfilt_zi come from SciPy on the host (filter design, as in the reference)."""
    from scipy.signal import sosfilt_zi
    dev = torch.device(device) if device is not None else default_device()
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    if sos.ndim != 2 or sos.shape[1] != 6:
        raise ValueError("sos must have shape (n_sections, 6)")
    n_sections = sos.shape[0]
    edge = _sos_edge(sos)
    with torch.cuda.device(dev):
        r = wavs if isinstance(wavs, Ragged) else Ragged.from_list_keep64(wavs, dev)
        if r.n == 0:
            return []
        if int(r.lens_host.min()) <= edge:
            raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % edge)
        lib = _lib.load()
        total = int(r.lens_host.sum())
        sos_d = _h2d(sos, dev)
        zi_d = _h2d(np.ascontiguousarray(sosfilt_zi(sos), dtype=np.float64), dev)
        ws_bytes = int(lib.ssr_sosfiltfilt_workspace_bytes(total, r.n, edge))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        y = torch.empty(total, dtype=torch.float64, device=dev)
        fn = lib.ssr_sosfiltfilt_f64 if r.data.dtype == torch.float64 else lib.ssr_sosfiltfilt
        _lib.check(fn(_vp(r.data), _vp(r.off), _vp(r.len), r.n, total, _vp(sos_d), _vp(zi_d), n_sections, edge, _vp(y), _vp(ws),
                      ws_bytes, _stream()))
        return r.split(y)


def _sos_edge(sos):
    n_sections = sos.shape[0]
    ntaps = 2 * n_sections + 1 - min(int((sos[:, 2] == 0).sum()), int((sos[:, 5] == 0).sum()))
    return 3 * ntaps


SOS_MULTI_MAX_DESIGNS = 48                                       # designs per launch (the C ABI's limit)
# Output doubles per launch.  A launch lasts as long as its LONGEST utterance (a serial recurrence, ~195 cycles per sample step) whatever
# the number of (design, utterance) recurrences beside it, until the chip's wave slots are full (4 recurrences per wave, ~2.5 waves per
# SIMD by LDS): 64 files x 36 designs are 576 waves on 1024 SIMDs.  16 GiB of output per launch (round 6; 4 GiB before) lets a batch of
# 256 files x 36 designs go in ONE launch - evaluate() with 36 IIR keys 0.94 -> 0.74 s per 367 files; sosfiltfilt_multi also keeps
# the launch inside a quarter of the device's free memory.
SOS_MULTI_MAX_DOUBLES = int(os.environ.get("SSR_SOS_MULTI_MAX_DOUBLES", 1 << 31))


def sosfiltfilt_multi(sos_list, wavs, device=None):
    """scipy.signal.sosfiltfilt(sos, x) for EVERY design of sos_list over one list of float32 waveforms: ssr_sosfiltfilt_multi, the
    designs side by side in one launch (a launch is latency-bound - the recurrence is serial in time - and fills an eighth of a wave
    per utterance: one design after the other costs the same latency each time).  -> [design][signal] float64 device tensors, every one
    bit-identical to sosfiltfilt(sos, ...).  Designs of more than 8 sections and float64 signals go through sosfiltfilt()."""
    from scipy.signal import sosfilt_zi
    dev = torch.device(device) if device is not None else default_device()
    sos_list = [np.ascontiguousarray(s_, dtype=np.float64) for s_ in sos_list]
    for s_ in sos_list:
        if s_.ndim != 2 or s_.shape[1] != 6:
            raise ValueError("sos must have shape (n_sections, 6)")
    if not sos_list:                   # no design: no keys (the reference's nested loops run zero times, ssr_eval/eval.py:243-258)
        return []
    with torch.cuda.device(dev):
        r = wavs if isinstance(wavs, Ragged) else Ragged.from_list_keep64(wavs, dev)
        if r.n == 0:
            return [[] for _ in sos_list]
        if r.data.dtype != torch.float32 or any(s_.shape[0] > 8 for s_ in sos_list) or not r.packed:
            return [sosfiltfilt(s_, r if r.packed else wavs, dev) for s_ in sos_list]
        edges = [_sos_edge(s_) for s_ in sos_list]
        if int(r.lens_host.min()) <= max(edges):
            raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % max(edges))
        lib = _lib.load()
        total = int(r.lens_host.sum())
        try:
            free_doubles = int(torch.cuda.mem_get_info(dev)[0]) // 8 // 4
        except Exception:
            free_doubles = SOS_MULTI_MAX_DOUBLES
        per_launch = max(1, min(SOS_MULTI_MAX_DESIGNS, min(SOS_MULTI_MAX_DOUBLES, free_doubles) // max(total, 1)))
        out = []
        for d0 in range(0, len(sos_list), per_launch):
            chunk = sos_list[d0:d0 + per_launch]
            D = len(chunk)
            sos_h, zi_h = np.zeros((D, 8, 6)), np.zeros((D, 8, 2))
            for d, s_ in enumerate(chunk):
                sos_h[d, :s_.shape[0]] = s_
                zi_h[d, :s_.shape[0]] = sosfilt_zi(s_)
            ns = np.array([s_.shape[0] for s_ in chunk], dtype=np.int32)
            eg = np.array(edges[d0:d0 + D], dtype=np.int32)
            sos_d, zi_d = _h2d(sos_h, dev), _h2d(zi_h, dev)
            ws_bytes = int(lib.ssr_sosfiltfilt_multi_workspace_bytes(total, r.n, eg.ctypes.data_as(C.c_void_p), D))
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
            y = torch.empty((D, total), dtype=torch.float64, device=dev)
            _lib.check(lib.ssr_sosfiltfilt_multi(_vp(r.data), _vp(r.off), _vp(r.len), r.n, total, _vp(sos_d), _vp(zi_d),
                                                 ns.ctypes.data_as(C.c_void_p), eg.ctypes.data_as(C.c_void_p), D, _vp(y), total, _vp(ws),
                                                 ws_bytes, _stream()))
            out += [r.split(y[d]) for d in range(D)]
        return out


def xcorr_argmax(a_list, b_list, device=None):
    """numpy.argmax(scipy.signal.correlate(a, b, "full")) for lists of equal-length float32 signal pairs (N4)."""
    dev = torch.device(device) if device is not None else default_device()
    with torch.cuda.device(dev):
        ra, rb = Ragged.from_list(a_list, dev), Ragged.from_list(b_list, dev)
        if ra.n != rb.n or not np.array_equal(ra.lens_host, rb.lens_host):
            raise ValueError("cross-correlation alignment needs pairs of equal length (unify_length first)")
        if ra.n == 0:
            return np.zeros(0, np.int64)
        if int(ra.lens_host.min()) < 1:
            raise ValueError("empty signal")
        lib = _lib.load()
        ws_bytes = int(lib.ssr_xcorr_workspace_bytes(ra.n, ra.max_len))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        out = torch.empty(ra.n, dtype=torch.int64, device=dev)
        _lib.check(lib.ssr_xcorr_argmax(_vp(ra.data), _vp(ra.off), _vp(rb.data), _vp(rb.off), _vp(ra.len), ra.n, ra.max_len,
                                        _vp(out), _vp(ws), ws_bytes, _stream()))
        return out.cpu().numpy()
