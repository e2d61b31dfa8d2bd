"""ctypes driver for the host emulation of the kernel bodies (tests/emu/ssr_emu.cpp).  Test infrastructure."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "ssr_emu.cpp")
SO = os.path.join(ROOT, "tests", "emu", "libssr_emu.so")

M_LSD, M_LOG_SISPEC, M_SISPEC, M_SSIM = 1, 2, 4, 8
M_ALL = 15


def build(force=False):
    deps = [SRC] + glob.glob(os.path.join(ROOT, "ssr_eval_amd", "csrc", "*.h"))
    if force or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", SO, SRC])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def num_frames(n, n_fft, hop):
    return 1 + (n + 2 * (n_fft // 2) - n_fft) // hop


def ragged(arrs):
    lens = np.array([len(a) for a in arrs], np.int32)
    off = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
    return np.concatenate(arrs).astype(np.float32), off, lens


def stft(sigs_a, sigs_b, n_fft, hop, precision=1, mode=0, out_kind=1, mask=0, units_per_chunk=8, est64=False,
         tgt64=False, wave=None, interleave=1):
    """mode 0 (pair): returns (mag_a list, mag_b list, part); mode 1 (single): (out_a list, out_b list, None).
    est64 / tgt64 (pair mode): the signals are float64 and run through the IN64 kernel variants."""
    a, a_off, lens = ragged(sigs_a)
    if est64:
        assert mode == 0
        a = np.concatenate(sigs_a).astype(np.float64)
    assert est64 or not tgt64
    if mode == 0:
        b, b_off, lens_b = ragged(sigs_b)
        assert (lens == lens_b).all()
    else:
        b, b_off = a, a_off
    T = np.array([num_frames(int(n), n_fft, hop) for n in lens])
    F = n_fft // 2 + 1
    frame_off = np.concatenate(([0], np.cumsum(T)[:-1])).astype(np.int64)
    units = T if mode == 0 else (T + 1) // 2
    n_chunks = int(-(-units.max() // units_per_chunk))
    if interleave > 1:                               # whole groups of `interleave` chunks (wave engine)
        n_chunks = -(-n_chunks // interleave) * interleave
    out_a = np.full((int(T.sum()), F), np.nan, np.float32)
    out_b = np.full((int(T.sum()), F), np.nan, np.float32)
    part = np.full((len(lens), n_chunks, 8), np.nan, np.float64) if mode == 0 else None
    tail = (_p(a_off, C.c_int64), _p(b_off, C.c_int64), _p(lens, C.c_int32), _p(frame_off, C.c_int64), len(lens),
            units_per_chunk, n_chunks, _p(out_a, C.c_float), _p(out_b, C.c_float), _p(part, C.c_double))
    if wave == "r3" and est64:                  # the rotating engine's float64-estimate variant (round 5)
        assert mode == 0 and not tgt64 and precision == 1
        rc = lib().emu_stft_r3_rot_est64(n_fft, hop, out_kind, mask, _p(a, C.c_double), _p(b, C.c_float), *tail)
        assert rc == 24, rc
        rc = 0
    elif wave in ("r3", "r3_2048", "r3_three"):   # radix-R x Bluestein on R autonomous waves (pair mode, n_fft = R q; R = 1, 2, 3): "r3" = the
        assert mode == 0 and not est64     # product's choice (M = 1536 where q <= 768), "r3_2048" = 2048-point transforms forced
        rc = lib().emu_stft_r3_wave(precision, n_fft, hop, out_kind, mask, {"r3": 1, "r3_2048": 0, "r3_three": 2}[wave], _p(a, C.c_float), _p(b, C.c_float), *tail)
        assert rc in (24, 32), rc
        rc = 0
    elif wave is not None:          # wave-autonomous engine: "full" or "split" exchange (pair mode, n_fft 2048, float32 signals)
        assert mode == 0 and n_fft == 2048 and not est64
        rc = lib().emu_stft_wave(precision, hop, out_kind, mask, 1 if wave == "split" else 0, interleave, _p(a, C.c_float),
                                 _p(b, C.c_float), *tail)
    elif est64:
        b64 = np.concatenate(sigs_b).astype(np.float64) if tgt64 else None
        rc = lib().emu_stft_in64(precision, n_fft, hop, out_kind, mask, _p(a, C.c_double),
                                 None if tgt64 else _p(b, C.c_float), _p(b64, C.c_double) if tgt64 else None, *tail)
    else:
        rc = lib().emu_stft(precision, n_fft, hop, mode, out_kind, mask, _p(a, C.c_float), _p(b, C.c_float), *tail)
    assert rc == 0, rc
    split = lambda o: [o[frame_off[i]:frame_off[i] + T[i]] for i in range(len(lens))]
    return split(out_a), split(out_b), part


def stft_r3_rot_est64x2(sigs_a, sigs_b, n_fft, hop, units_per_chunk=8):
    """Two float64 estimates per complex transform on the rotating engine, images only (k_stft_r3_rot<double, false, 3, 24, 7>):
    -> (float32 magnitude rows of a, of b)."""
    a64 = np.concatenate(sigs_a).astype(np.float64)
    b64 = np.concatenate(sigs_b).astype(np.float64)
    _, a_off, lens = ragged(sigs_a)
    _, b_off, lens_b = ragged(sigs_b)
    assert (lens == lens_b).all()
    T = np.array([num_frames(int(n), n_fft, hop) for n in lens])
    F = n_fft // 2 + 1
    frame_off = np.concatenate(([0], np.cumsum(T)[:-1])).astype(np.int64)
    n_chunks = int(-(-T.max() // units_per_chunk))
    out_a = np.full((int(T.sum()), F), np.nan, np.float32)
    out_b = np.full((int(T.sum()), F), np.nan, np.float32)
    rc = lib().emu_stft_r3_rot_est64x2(n_fft, hop, _p(a64, C.c_double), _p(b64, C.c_double), _p(a_off, C.c_int64), _p(b_off, C.c_int64),
                                       _p(lens, C.c_int32), _p(frame_off, C.c_int64), len(lens), units_per_chunk, n_chunks,
                                       _p(out_a, C.c_float), _p(out_b, C.c_float))
    assert rc == 24, rc
    split = lambda o: [o[frame_off[i]:frame_off[i] + T[i]] for i in range(len(lens))]
    return split(out_a), split(out_b)


def spectro_desc(sps):
    T = np.array([s.shape[0] for s in sps], np.int32)
    frame_off = np.concatenate(([0], np.cumsum(T)[:-1])).astype(np.int64)
    return np.ascontiguousarray(np.concatenate(sps).astype(np.float32)), frame_off, T


def ssim_parts(xs, ys, rows_per_tile=16, cpt=None, contig=False):
    """contig: the pair pipeline's layout - rows padded to a multiple of 4 floats (NaN in the padding) - and the CPT = 4
    kernel variant that reads them with aligned 16-byte loads."""
    x, frame_off, T = spectro_desc(xs)
    y, _, _ = spectro_desc(ys)
    F = xs[0].shape[1]
    pitch = 0
    if contig:
        pitch = (F + 3) & ~3
        x = np.ascontiguousarray(np.pad(x, ((0, 0), (0, pitch - F)), constant_values=np.nan))
        y = np.ascontiguousarray(np.pad(y, ((0, 0), (0, pitch - F)), constant_values=np.nan))
    n_row_tiles = int(-(-(T.max() - 6) // rows_per_tile))
    c, ns = C.c_int(), C.c_int()
    lib().emu_ssim_geom(F, C.byref(c), C.byref(ns))
    if cpt is None:
        cpt, n_strips = c.value, ns.value
    else:                                            # force a smaller CPT to exercise the multi-strip path
        n_strips = int(-(-(F - 6) // (64 * cpt)))
    part = np.full((len(xs), n_row_tiles * n_strips), np.nan)
    rc = lib().emu_ssim(_p(x, C.c_float), _p(y, C.c_float), _p(frame_off, C.c_int64), _p(T, C.c_int32), len(xs), F,
                        pitch, 1 if contig else 0, rows_per_tile, n_row_tiles, n_strips, cpt, _p(part, C.c_double))
    assert rc == 0
    return part, T


def specred_parts(xs, ys, mask=7, rows_per_chunk=8):
    x, frame_off, T = spectro_desc(xs)
    y, _, _ = spectro_desc(ys)
    F = xs[0].shape[1]
    n_chunks = int(-(-T.max() // rows_per_chunk))
    part = np.full((len(xs), n_chunks, 8), np.nan)
    rc = lib().emu_specred(_p(x, C.c_float), _p(y, C.c_float), _p(frame_off, C.c_int64), _p(T, C.c_int32), len(xs), F,
                           mask, rows_per_chunk, n_chunks, _p(part, C.c_double))
    assert rc == 0
    return part, T


def finalize(part, ssim_part, T, F, mask):
    T = np.asarray(T, np.int32)
    n = len(T)
    out = np.zeros((n, 4))
    rc = lib().emu_finalize(_p(part, C.c_double), part.shape[1] if part is not None else 0,
                            _p(ssim_part, C.c_double), ssim_part.shape[1] if ssim_part is not None else 0,
                            _p(T, C.c_int32), F, mask, n, _p(out, C.c_double))
    assert rc == 0
    return out


def pair_metrics(ests, tgts, n_fft, hop, precision=1, mask=M_ALL, units_per_chunk=8, rows_per_tile=16, est64=False,
                 tgt64=False, wave=None, interleave=1):
    ea, tb, part = stft(ests, tgts, n_fft, hop, precision, 0, 1, mask, units_per_chunk, est64=est64, tgt64=tgt64, wave=wave,
                        interleave=interleave)
    sp, T = ssim_parts(ea, tb, rows_per_tile) if mask & M_SSIM else (None, np.array([e.shape[0] for e in ea]))
    return finalize(part, sp, T, n_fft // 2 + 1, mask)


def lowpass(sigs, cuts, n_fft=2048, hop=441, precision=1, pairs_per_chunk=4, wave=None, interleave=1):
    a, off, lens = ragged(sigs)
    T = np.array([num_frames(int(n), n_fft, hop) for n in lens])
    frame_off = np.concatenate(([0], np.cumsum(T)[:-1])).astype(np.int64)
    cuts = np.asarray(cuts, np.int32)
    frames = np.full((int(T.sum()), n_fft), np.nan, np.float32)
    n_chunks = int(-(-((T.max() + 1) // 2) // pairs_per_chunk))
    if interleave > 1:
        n_chunks = -(-n_chunks // interleave) * interleave
    if wave is not None:
        assert n_fft == 2048
        rc = lib().emu_lowpass_wave(precision, hop, 0 if wave == "full" else 1, 1 if wave == "paired" else 0, interleave, _p(a, C.c_float), _p(off, C.c_int64),
                                    _p(lens, C.c_int32), _p(cuts, C.c_int32), _p(frame_off, C.c_int64), len(lens), pairs_per_chunk,
                                    n_chunks, None, None, _p(frames, C.c_float))
    else:
        rc = lib().emu_lowpass_frames(precision, n_fft, hop, _p(a, C.c_float), _p(off, C.c_int64), _p(lens, C.c_int32),
                                      _p(cuts, C.c_int32), _p(frame_off, C.c_int64), len(lens), pairs_per_chunk, n_chunks,
                                      None, None, _p(frames, C.c_float))
    assert rc == 0
    out = np.full(int(lens.sum()), np.nan, np.float32)
    rc = lib().emu_ola(n_fft, hop, 1 if wave == "paired" else 0, _p(frames, C.c_float), _p(frame_off, C.c_int64), _p(lens, C.c_int32),
                       _p(off, C.c_int64), len(lens), int(lens.max()), _p(out, C.c_float))
    assert rc == 0
    return [out[off[i]:off[i] + lens[i]] for i in range(len(lens))]


def lowpass_group(sigs, cuts, hop=441, rounds_per_chunk=1000):
    """ssr_lowpass_group.h (fused overlap-add, float64 2048-point plans): -> list of float32 outputs."""
    a, off, lens = ragged(sigs)
    cuts = np.asarray(cuts, np.int32)
    max_rounds = -(-((num_frames(int(lens.max()), 2048, hop) + 1) // 2) // 4)
    n_chunks = -(-max_rounds // rounds_per_chunk)
    out = np.full(int(lens.sum()), np.nan, np.float32)
    rc = lib().emu_lowpass_group(hop, _p(a, C.c_float), _p(off, C.c_int64), _p(lens, C.c_int32), _p(cuts, C.c_int32), None,
                                 _p(off, C.c_int64), len(lens), rounds_per_chunk, n_chunks, None, None, _p(out, C.c_float))
    assert rc == 0, rc
    return [out[off[i]:off[i] + lens[i]] for i in range(len(lens))]


def istft_group(res, ims, lengths, hop=441, rounds_per_chunk=1000):
    re, frame_off, T = spectro_desc(res)
    im, _, _ = spectro_desc(ims)
    lens = np.asarray(lengths, np.int32)
    assert all(num_frames(int(n), 2048, hop) == t for n, t in zip(lens, T))
    off = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
    max_rounds = -(-((int(T.max()) + 1) // 2) // 4)
    n_chunks = -(-max_rounds // rounds_per_chunk)
    out = np.full(int(lens.sum()), np.nan, np.float32)
    rc = lib().emu_lowpass_group(hop, None, None, _p(lens, C.c_int32), None, _p(frame_off, C.c_int64), _p(off, C.c_int64), len(lens),
                                 rounds_per_chunk, n_chunks, _p(re, C.c_float), _p(im, C.c_float), _p(out, C.c_float))
    assert rc == 0, rc
    return [out[off[i]:off[i] + lens[i]] for i in range(len(lens))]


def istft(res, ims, lengths, n_fft=2048, hop=441, precision=1, pairs_per_chunk=4, wave=None, interleave=1):
    re, frame_off, T = spectro_desc(res)
    im, _, _ = spectro_desc(ims)
    lens = np.asarray(lengths, np.int32)
    assert all(num_frames(int(n), n_fft, hop) == t for n, t in zip(lens, T))
    off = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
    frames = np.full((int(T.sum()), n_fft), np.nan, np.float32)
    n_chunks = int(-(-((T.max() + 1) // 2) // pairs_per_chunk))
    if interleave > 1:
        n_chunks = -(-n_chunks // interleave) * interleave
    if wave is not None:
        assert n_fft == 2048
        rc = lib().emu_lowpass_wave(precision, hop, 0 if wave == "full" else 1, 1 if wave == "paired" else 0, interleave, None, None, _p(lens, C.c_int32), None,
                                    _p(frame_off, C.c_int64), len(lens), pairs_per_chunk, n_chunks, _p(re, C.c_float),
                                    _p(im, C.c_float), _p(frames, C.c_float))
    else:
        rc = lib().emu_lowpass_frames(precision, n_fft, hop, None, None, _p(lens, C.c_int32), None, _p(frame_off, C.c_int64),
                                      len(lens), pairs_per_chunk, n_chunks, _p(re, C.c_float), _p(im, C.c_float),
                                      _p(frames, C.c_float))
    assert rc == 0
    out = np.full(int(lens.sum()), np.nan, np.float32)
    rc = lib().emu_ola(n_fft, hop, 1 if wave == "paired" else 0, _p(frames, C.c_float), _p(frame_off, C.c_int64), _p(lens, C.c_int32),
                       _p(off, C.c_int64), len(lens), int(lens.max()), _p(out, C.c_float))
    assert rc == 0
    return [out[off[i]:off[i] + lens[i]] for i in range(len(lens))]


def resample(sigs, up, down, taps_full, n_pre_remove, groups=0, taps_in_lds=1, dtype=np.float32):
    ct = C.c_float if dtype == np.float32 else C.c_double
    lens = np.array([len(a) for a in sigs], np.int32)
    off = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
    a = np.concatenate(sigs).astype(dtype)
    out_len = np.array([-(-int(n) * up // down) for n in lens], np.int32)
    out_off = np.concatenate(([0], np.cumsum(out_len)[:-1])).astype(np.int64)
    out = np.full(int(out_len.sum()), np.nan, dtype)
    taps = np.ascontiguousarray(taps_full, dtype)
    fn = lib().emu_resample if dtype == np.float32 else lib().emu_resample_f64
    rc = fn(_p(a, ct), _p(off, C.c_int64), _p(lens, C.c_int32), _p(out_off, C.c_int64), _p(out_len, C.c_int32), len(lens),
            int(out_len.max()), up, down, _p(taps, ct), len(taps), n_pre_remove, groups, taps_in_lds, _p(out, ct))
    assert rc == 0
    return [out[out_off[i]:out_off[i] + out_len[i]] for i in range(len(lens))]


def resample_mfma(sigs, up, down, taps_full, n_pre_remove, n_wg=3, geometry=None):
    """ssr_resample_mfma.h on the host (the matrix core's fused-multiply-add chains restated per element)."""
    lens = np.array([len(a) for a in sigs], np.int32)
    off = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
    a = np.concatenate(sigs).astype(np.float32)
    out_len = np.array([-(-int(n) * up // down) for n in lens], np.int32)
    out_off = np.concatenate(([0], np.cumsum(out_len)[:-1])).astype(np.int64)
    out = np.full(int(out_len.sum()), np.nan, np.float32)
    taps = np.ascontiguousarray(taps_full, np.float32)
    rc = lib().emu_resample_mfma(_p(a, C.c_float), _p(off, C.c_int64), _p(lens, C.c_int32), _p(out_off, C.c_int64),
                                 _p(out_len, C.c_int32), len(lens), int(out_len.max()), up, down, _p(taps, C.c_float), len(taps),
                                 n_pre_remove, n_wg, _p(out, C.c_float))
    if rc < 0:
        return None
    if geometry is not None:
        geometry.append((rc // 1000, rc % 1000))
    return [out[out_off[i]:out_off[i] + out_len[i]] for i in range(len(lens))]


def sosfiltfilt(sos, sigs, dtype=np.float32):
    from scipy.signal import sosfilt_zi
    a, off, lens = ragged(sigs)
    if dtype == np.float64:
        a = np.concatenate(sigs).astype(np.float64)
    sos = np.ascontiguousarray(sos, np.float64)
    zi = np.ascontiguousarray(sosfilt_zi(sos), np.float64)
    S = sos.shape[0]
    edge = 3 * (2 * S + 1 - min(int((sos[:, 2] == 0).sum()), int((sos[:, 5] == 0).sum())))
    fwd = np.full(int(lens.sum()) + 2 * edge * len(lens), np.nan)
    y = np.full(int(lens.sum()), np.nan)
    fn, ct = (lib().emu_sosfiltfilt, C.c_float) if dtype == np.float32 else (lib().emu_sosfiltfilt_f64, C.c_double)
    rc = fn(_p(a, ct), _p(off, C.c_int64), _p(lens, C.c_int32), len(lens), _p(sos, C.c_double), _p(zi, C.c_double), S, edge,
            _p(fwd, C.c_double), _p(y, C.c_double))
    assert rc == 0
    return [y[off[i]:off[i] + lens[i]] for i in range(len(lens))]


def xcorr_argmax(a_list, b_list):
    a, off, lens = ragged(a_list)
    b, _, lens_b = ragged(b_list)
    assert (lens == lens_b).all()
    out = np.full(len(lens), -1, np.int64)
    rc = lib().emu_xcorr_argmax(_p(a, C.c_float), _p(off, C.c_int64), _p(b, C.c_float), _p(off, C.c_int64),
                                _p(lens, C.c_int32), len(lens), int(lens.max()), _p(out, C.c_int64))
    assert rc == 0
    return out


def resample_sinc_tab(sigs, sr_orig, sr_new, name="kaiser_best"):
    """The device's round-5 tap loop on the host: phase-major table built by ssr_sinc_table_body, every output from its own row
    with the loop run to the longest wing (ssr_sinc_one_tab_host)."""
    from oracle import resampy as orsy
    ratio = float(sr_new) / sr_orig
    win, delta, num_table, step, scale = orsy.filter_tables(ratio, name)
    a, off, lens = ragged(sigs)
    out_len = np.array([int(int(n) * ratio) for n in lens], np.int32)
    out_off = np.concatenate(([0], np.cumsum(out_len)[:-1])).astype(np.int64)
    tr = orsy.time_register(int(out_len.max()), ratio)
    out = np.full(int(out_len.sum()), np.nan, np.float32)
    rc = lib().emu_resample_sinc_tab(_p(a, C.c_float), _p(off, C.c_int64), _p(lens, C.c_int32), _p(out_off, C.c_int64),
                                     _p(out_len, C.c_int32), len(lens), _p(tr, C.c_double), _p(win, C.c_double), _p(delta, C.c_double),
                                     len(win), num_table, step, C.c_double(scale), _p(out, C.c_float))
    assert rc >= 8 and rc % 4 == 0
    return [out[out_off[i]:out_off[i] + out_len[i]] for i in range(len(lens))]


def resample_sinc(sigs, sr_orig, sr_new, name="kaiser_best", phase_period=None, lds_cap_floats=12288, geometry=None):
    """ssr_sinc.h on the host: tables / time register from the oracle module's published-parameter restatement.
    phase_period: None = what the product passes (a of sr_new / sr_orig = a / b); geometry: a list that receives (period, m)."""
    import math
    from oracle import resampy as orsy
    ratio = float(sr_new) / sr_orig
    if phase_period is None:
        phase_period = int(sr_new) // math.gcd(int(sr_new), int(sr_orig))
    win, delta, num_table, step, scale = orsy.filter_tables(ratio, name)
    a, off, lens = ragged(sigs)
    out_len = np.array([int(int(n) * ratio) for n in lens], np.int32)
    out_off = np.concatenate(([0], np.cumsum(out_len)[:-1])).astype(np.int64)
    tr = orsy.time_register(int(out_len.max()), ratio)
    out = np.full(int(out_len.sum()), np.nan, np.float32)
    rc = lib().emu_resample_sinc(_p(a, C.c_float), _p(off, C.c_int64), _p(lens, C.c_int32), _p(out_off, C.c_int64),
                                 _p(out_len, C.c_int32), len(lens), int(out_len.max()), _p(tr, C.c_double), _p(win, C.c_double),
                                 _p(delta, C.c_double), len(win), num_table, step, C.c_double(scale), C.c_double(ratio),
                                 int(phase_period), int(lds_cap_floats), _p(out, C.c_float))
    assert rc > 0
    if geometry is not None:          # (period, phases per wave, padded window, blocks' m)
        geometry.append((rc // 1000000, rc % 1000000 // 10000, rc % 10000 // 1000, rc % 1000))
    return [out[out_off[i]:out_off[i] + out_len[i]] for i in range(len(lens))]
