"""Audio file ingest.

The reference decodes with librosa.load - soundfile decode, mono mix, then resampy's "kaiser_best" band-limited
interpolation when the file's rate differs from the requested one (ssr_eval/eval.py:242, ssr_eval/metrics.py:21-24) - and
shells out to ``sox -r`` for the evaluation-rate target (eval.py:133-134).  Here:

* decode: FLAC - the format of the VCTK test set - through the native decoder behind the C ABI (``ssr_flac_*``, csrc/ssr_flac.h:
  every file is checked against the MD5 of its PCM that STREAMINFO carries); PCM .wav through the arena reader / the
  standard-library ``wave`` module; anything else through ``soundfile`` when that package is importable - host work, fanned out
  over a thread pool by ``load_audio_batch``;
* rate change: ``ssr_resample_sinc`` on the GPU - resampy's kaiser_best algorithm and filter (bit-identical to the NumPy
  restatement in oracle/resampy.py; resampy itself is not in the image, so parity with the package is unpinned), one
  ragged launch per (file rate -> requested rate) group.  It also stands in for sox's ``rate`` effect, whose filter is not
  published in the reference tree: targets produced by sox on real VCTK files are comparable, not bit-identical.
"""
import os
import struct
import wave

import numpy as np


try:
    import soundfile as _sf  # optional (WAV / FLAC / OGG)
except ImportError:
    _sf = None


def _mono(x, nch):
    """Channel mean of interleaved samples [n * nch] (librosa.load's mono=True); a mono file passes through."""
    return x if nch == 1 else np.ascontiguousarray(x.reshape(-1, nch).mean(axis=1), dtype=np.float32)


FLAC_VERIFY_MD5 = True      # compare every decoded file with the MD5 signature in its STREAMINFO (a mismatch raises)


def _is_flac(path):
    return path.lower().endswith(".flac")


def flac_info(path):
    """(sample_rate, channels, bits, frames per channel or 0 if unknown) of a FLAC file (ssr_flac_info: host code, no GPU)."""
    import ctypes as C
    from . import _lib
    sr, nch, bits, has = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    total = C.c_int64()
    _lib.check(_lib.load().ssr_flac_info(os.fsencode(path), C.byref(sr), C.byref(nch), C.byref(bits), C.byref(total), C.byref(has)))
    return sr.value, nch.value, bits.value, total.value


def read_flac_int(path):
    """-> (interleaved integer frames [n * nch] - int16 for streams of <= 16 bits, int32 otherwise -, channels, rate, bits)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    sr, nch, bits, total = flac_info(path)
    n = C.c_int64()
    if total == 0:                                     # a streamed file without a sample count: count first
        _lib.check(lib.ssr_flac_decode_i32(os.fsencode(path), None, 0, 0, C.byref(n)))
        total = n.value
    out = np.empty(total * nch, dtype=np.int16 if bits <= 16 else np.int32)
    fn = lib.ssr_flac_decode_pcm16 if bits <= 16 else lib.ssr_flac_decode_i32
    _lib.check(fn(os.fsencode(path), out.ctypes.data_as(C.c_void_p), out.size, 1 if FLAC_VERIFY_MD5 else 0, C.byref(n)))
    return out[:n.value * nch], nch, sr, bits


def read_audio(path):
    """-> (float32 mono [n], sample_rate)."""
    if _is_flac(path):
        v, nch, sr, bits = read_flac_int(path)
        x = v.astype(np.float32)
        x *= np.float32(1.0 / (1 << (bits - 1)))      # libsndfile's normalisation for float reads (a power of two: exact)
        return _mono(x, nch), int(sr)
    if _sf is not None:
        x, sr = _sf.read(path, dtype="float32", always_2d=True)
        return _mono(np.ascontiguousarray(x, dtype=np.float32).reshape(-1), x.shape[1]), int(sr)
    if not path.lower().endswith(".wav"):
        raise RuntimeError("decoding %s needs the `soundfile` package (only PCM .wav is readable without it)"
                           % os.path.basename(path))
    with wave.open(path, "rb") as f:
        sr, nch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, "<i2").astype(np.float32)
        x *= np.float32(1.0 / 32768.0)                             # a power of two: the same values as the division
    elif sw == 4:
        x = np.frombuffer(raw, "<i4").astype(np.float32) / 2147483648.0
    elif sw == 3:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
    elif sw == 1:
        x = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise RuntimeError("unsupported sample width %d" % sw)
    return _mono(x, nch), int(sr)


class RawAudio:
    """A decoded file as the decoder hands it over: `pcm` = interleaved int16 frames [n * nch] of a PCM_16 file (the float32
    conversion and the mono mix then run on the GPU, backend.upload_decoded), or `x` = float32 mono [n] for anything else."""
    __slots__ = ("pcm", "x", "nch", "sr")

    def __init__(self, pcm, x, nch, sr):
        self.pcm, self.x, self.nch, self.sr = pcm, x, nch, sr

    @property
    def n_frames(self):
        return self.pcm.shape[0] // self.nch if self.pcm is not None else self.x.shape[0]

    def to_float(self):
        """float32 mono on the host: exactly what read_audio returns."""
        if self.x is None:
            x = self.pcm.astype(np.float32)
            x *= np.float32(1.0 / 32768.0)
            self.x = _mono(x, self.nch)
        return self.x


def read_audio_raw(path):
    """-> RawAudio: the int16 frames of a 16-bit PCM file untouched (up to 8 channels), float32 mono otherwise."""
    if _is_flac(path):
        sr, nch, bits, _ = flac_info(path)
        if bits == 16 and nch <= 8:
            v, nch, sr, bits = read_flac_int(path)
            return RawAudio(v, None, int(nch), int(sr))
        x, sr = read_audio(path)
        return RawAudio(None, x, 1, sr)
    if _sf is not None:
        info = _sf.info(path)
        if info.subtype == "PCM_16" and info.channels <= 8:
            pcm, sr = _sf.read(path, dtype="int16", always_2d=True)
            return RawAudio(np.ascontiguousarray(pcm).reshape(-1), None, int(info.channels), int(sr))
        x, sr = read_audio(path)
        return RawAudio(None, x, 1, sr)
    if path.lower().endswith(".wav"):
        with wave.open(path, "rb") as f:
            sr, nch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
            if sw == 2 and nch <= 8:
                return RawAudio(np.frombuffer(f.readframes(n), "<i2"), None, int(nch), int(sr))
    x, sr = read_audio(path)
    return RawAudio(None, x, 1, sr)


def write_wav(path, x, sr):
    """``soundfile.write(path, x, sr)`` of a mono signal (ssr_eval/eval.py:153-154).  For a ``.wav`` name soundfile's default
    subtype is PCM_16, so the reference's artefact files are 16-bit as well: libsndfile scales by 32768, rounds to nearest
    (ties to even) and - python-soundfile switches clipping on - saturates at -32768 / 32767.  Without the package the same
    conversion is done here through the standard-library writer."""
    x = np.asarray(x, np.float32).reshape(-1)
    if _sf is not None:
        _sf.write(path, x, int(sr))
        return
    q = np.clip(np.rint(x.astype(np.float64) * 32768.0), -32768.0, 32767.0).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(sr))
        f.writeframes(q.tobytes())


def load_audio(path, sr=None, res_type="kaiser_best"):
    """librosa.load(path, sr=sr): decode, mono, and - if sr is given and differs - kaiser_best resampling on the GPU."""
    return load_audio_batch([path], sr, res_type)[0]


_POOL = {}


def usable_cores():
    """Host cores this PROCESS may use: the scheduler affinity mask (not os.cpu_count(), which ignores it) capped by the cgroup
    CPU quota (a container that sees 64 logical CPUs and is granted 16 runs a 64-thread pool slower than a 16-thread one)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def local_world_size():
    """Ranks of this job on THIS node (torchrun exports LOCAL_WORLD_SIZE; a hand-rolled launcher may only export WORLD_SIZE
    on a single node): they share the node's host cores."""
    for key in ("LOCAL_WORLD_SIZE", "SSR_LOCAL_WORLD_SIZE"):
        v = os.environ.get(key)
        if v and v.isdigit() and int(v) > 0:
            return int(v)
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return max(1, dist.get_world_size())          # no local size announced: assume one node
    except Exception:
        pass
    return 1


def decode_threads():
    """Reader / decoder threads per process: this rank's share of the cores the job may use, at most 16 (eight ranks on a
    16-core grant start 8 x 2 readers, not 8 x 16; SSR_DECODE_THREADS overrides)."""
    v = os.environ.get("SSR_DECODE_THREADS")
    if v and v.isdigit() and int(v) > 0:
        return int(v)
    return max(1, min(16, usable_cores() // local_world_size()))


def _decode_pool(threads):
    """One pool per worker count for the life of the process (starting 16 threads per batch cost more than decoding it)."""
    from concurrent.futures import ThreadPoolExecutor
    if threads not in _POOL:
        _POOL[threads] = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="ssr-decode")
    return _POOL[threads]


def decode_batch(paths, threads=None):
    """[(float32 mono waveform, file rate)] for a list of files; decoding fans out over host threads."""
    paths = list(paths)
    if len(paths) <= 1:
        return [read_audio(p) for p in paths]
    return list(_decode_pool(threads or decode_threads()).map(read_audio, paths))


def decode_async(paths, threads=None, raw=False):
    """Start decoding `paths` on the pool; -> a function that waits for and returns the decode_batch result
    (raw: RawAudio items - 16-bit PCM stays int16 for the GPU to convert)."""
    fn = read_audio_raw if raw else read_audio
    futures = [_decode_pool(threads or decode_threads()).submit(fn, p) for p in paths]
    return lambda: [f.result() for f in futures]


def _wav_pcm16_layout(path):
    """(data offset, data bytes, channels, rate) of a 16-bit PCM RIFF / WAVE file - enough to read its frames straight into a
    staging arena - or None for anything else (other sample formats, other containers: those go through read_audio_raw).
    A 16-bit FLAC file with a known length answers ("flac", bytes of its decoded PCM, channels, rate): the decoder threads then
    DECODE it straight into the arena (ssr_flac_decode_pcm16)."""
    if _is_flac(path):
        try:
            sr, nch, bits, total = flac_info(path)
        except Exception:
            return None                                # (read_audio_raw will raise the real error)
        return ("flac", 2 * total * nch, nch, sr) if bits == 16 and 1 <= nch <= 8 and total > 0 else None
    try:
        with open(path, "rb") as f:
            head = f.read(12)
            if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
                return None
            fmt = None
            while True:
                h = f.read(8)
                if len(h) < 8:
                    return None
                cid, size = h[:4], struct.unpack("<I", h[4:])[0]
                if cid == b"fmt ":
                    d = f.read(size + (size & 1))
                    if len(d) < 16:
                        return None
                    tag, nch, sr, _, _, bits = struct.unpack("<HHIIHH", d[:16])
                    if tag == 0xFFFE and len(d) >= 26:                # WAVE_FORMAT_EXTENSIBLE: the sub-format's first two bytes
                        tag = struct.unpack("<H", d[24:26])[0]
                    fmt = (tag, nch, sr, bits)
                elif cid == b"data":
                    if fmt is None or fmt[0] != 1 or fmt[3] != 16 or not 1 <= fmt[1] <= 8:
                        return None
                    off = f.tell()
                    n = min(size, os.fstat(f.fileno()).st_size - off)
                    n -= n % (2 * fmt[1])
                    return off, n, fmt[1], fmt[2]
                else:
                    f.seek(size + (size & 1), 1)
    except OSError:
        return None


def duration_hint(path):
    """Seconds of audio in a file from its header alone (16-bit WAV / any FLAC with a known length), the file size in units of
    16-bit 44.1 kHz mono otherwise - the weight length-balanced sharding deals by (ssr_eval_amd.dist.shard_indices_balanced); the
    same value on every rank that sees the same tree."""
    lay = _wav_pcm16_layout(path)
    if lay is not None:
        return lay[1] / (2.0 * lay[2] * max(lay[3], 1))
    if _is_flac(path):
        try:
            sr, nch, bits, total = flac_info(path)
            if total > 0 and sr > 0:
                return total / float(sr)
        except Exception:
            pass
    try:
        return os.path.getsize(path) / (2.0 * 44100.0)
    except OSError:
        return 0.0


class PackedBatch:
    """A batch of decoded files whose 16-bit PCM frames sit back to back in one page-locked arena (backend._Staging): the
    decoder threads read the files' data chunks STRAIGHT into it, so the batch crosses PCIe in one asynchronous copy with no
    host-side float pass and no host-side packing.  Files that are not 16-bit PCM WAVE ride along as RawAudio items."""

    def __init__(self, paths, device):
        from . import backend as B
        self.paths = list(paths)
        lay = [_wav_pcm16_layout(p) for p in self.paths]
        self.pcm_idx = [i for i, l in enumerate(lay) if l is not None and l[1] > 0]
        self.other_idx = [i for i in range(len(lay)) if i not in set(self.pcm_idx)]
        self.sizes = np.array([lay[i][1] // 2 for i in self.pcm_idx], dtype=np.int64)          # int16 elements
        self.chans = np.array([lay[i][2] for i in self.pcm_idx], dtype=np.int32)
        self.frames = self.sizes // np.maximum(self.chans, 1)
        self.in_off = np.concatenate(([0], np.cumsum(self.sizes)[:-1])) if len(self.sizes) else np.zeros(0, np.int64)
        self.total = int(self.sizes.sum())
        self.srs = [None] * len(lay)
        for i in self.pcm_idx:
            self.srs[i] = int(lay[i][3])
        self.device = B.default_device() if device is None else device
        self.staging = B._Staging.get(self.device)
        self.k, self.arena = self.staging.arena(self.total) if self.total else (None, None)
        host = self.arena.numpy() if self.total else None
        pool = _decode_pool(decode_threads())

        def read_into(j):
            i = self.pcm_idx[j]
            o, n = int(self.in_off[j]), int(self.sizes[j])
            if lay[i][0] == "flac":                    # decode into the arena (the C call releases the GIL)
                import ctypes as C
                from . import _lib
                got = C.c_int64()
                _lib.check(_lib.load().ssr_flac_decode_pcm16(os.fsencode(self.paths[i]), C.c_void_p(host[o:o + n].ctypes.data), n,
                                                             1 if FLAC_VERIFY_MD5 else 0, C.byref(got)))
                if got.value * int(lay[i][2]) != n:
                    raise OSError("%s: decoded %d frames, STREAMINFO announces %d" % (self.paths[i], got.value, n // int(lay[i][2])))
                return
            with open(self.paths[i], "rb") as f:
                f.seek(lay[i][0])
                got = f.readinto(memoryview(host[o:o + n]).cast("B"))
            if got != 2 * n:
                raise OSError("short read on %s" % self.paths[i])
        self._fut = [pool.submit(read_into, j) for j in range(len(self.pcm_idx))]
        self._fut_other = [pool.submit(read_audio_raw, self.paths[i]) for i in self.other_idx]
        self.others = None

    def wait(self):
        for f in self._fut:
            f.result()
        self.others = [f.result() for f in self._fut_other]
        for i, r in zip(self.other_idx, self.others):
            self.srs[i] = r.sr
        return self


def decode_packed_async(paths, device=None):
    """Start reading `paths` into a staging arena; -> a function that waits for the reads and returns the PackedBatch."""
    pb = PackedBatch(paths, device)
    return pb.wait


def to_rate_resident(on_dev, file_srs, sr, res_type="kaiser_best"):
    """to_rate for waveforms that already live in HBM (device tensors) -> device tensors; an item already at `sr` is returned
    as it is (the same tensor)."""
    out = list(on_dev)
    groups = {}
    for i, file_sr in enumerate(file_srs):
        if sr is not None and int(sr) != file_sr:
            groups.setdefault(file_sr, []).append(i)
    if groups:
        from . import backend as B
        for file_sr, idx in groups.items():
            for i, y in zip(idx, B.resample_sinc([on_dev[i] for i in idx], file_sr, int(sr), res_type)):
                out[i] = y
    return out


def to_rate(decoded, sr, res_type="kaiser_best", keep_on_device=False, resident=None):
    """The rate change of librosa.load for decoded [(x, file_sr)] items: one ragged GPU launch per distinct file rate.
    keep_on_device: resampled items are returned as device tensors (views of the launch's output) instead of ndarrays -
    for a consumer that is another GPU stage; items already at `sr` stay the host arrays they are.
    resident: the same waveforms as device tensors, if the caller has uploaded them already (no second transfer)."""
    out = [None] * len(decoded)
    groups = {}
    for i, (x, file_sr) in enumerate(decoded):
        if sr is None or int(sr) == file_sr:
            out[i] = x
        else:
            groups.setdefault(file_sr, []).append(i)
    if groups:
        from . import backend as B
        for file_sr, idx in groups.items():
            ys = B.resample_sinc([decoded[i][0] if resident is None else resident[i] for i in idx], file_sr, int(sr), res_type)
            for i, y in zip(idx, ys):
                out[i] = y if keep_on_device else y.cpu().numpy()
    return out


def load_audio_batch(paths, sr=None, res_type="kaiser_best", threads=None):
    """librosa.load for a LIST of files -> list of float32 ndarrays."""
    return to_rate(decode_batch(paths, threads), sr, res_type)
