"""CPU tests of the FLAC decoder behind the C ABI (ssr_flac_info / ssr_flac_decode_pcm16 / ssr_flac_decode_i32: host code, no GPU)
against streams produced by the specification-written test encoder (tests/flac_fixture.py).  The MD5 signature in STREAMINFO -
hashlib's in the fixtures - is verified by the decoder's own RFC 1321 implementation on every decode."""
import ctypes as C
import os

import numpy as np
import pytest

import flac_fixture as FF
from ssr_eval_amd import _lib, io as sio


def _signal(n, nch, bits, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    full = (1 << (bits - 1)) - 1
    x = np.stack([0.3 * np.sin(2 * np.pi * (110 + 40 * c) * t / 16000 * (1 + 0.1 * np.sin(t / 3000))) + 0.02 * rng.standard_normal(n)
                  + 0.2 * np.sin(2 * np.pi * 1900 * t / 16000) for c in range(nch)], axis=1)
    x[n // 3:n // 3 + 700] = 0.0                                  # digital silence: CONSTANT subframes
    if nch == 2:
        x[:, 1] = 0.7 * x[:, 0] + 0.3 * x[:, 1]                  # correlated channels: the side signal is small
    return np.clip(np.rint(x * full), -full - 1, full).astype(np.int64)


def _write(tmp_path, name, data):
    p = tmp_path / name
    p.write_bytes(data)
    return str(p)


@pytest.mark.parametrize("nch,bits,block", [(1, 16, 1152), (2, 16, 4096), (2, 16, 192), (1, 8, 576), (1, 12, 256), (2, 20, 1000),
                                            (1, 24, 4608), (2, 24, 300), (3, 16, 512)])
def test_decoder_returns_the_encoders_input_bit_for_bit(tmp_path, nch, bits, block):
    x = _signal(9000 + 37 * bits, nch, bits, seed=bits + nch)
    path = _write(tmp_path, "a.flac", FF.encode(x, 44100 if bits != 12 else 37123, bits, block))
    sr, c, b, total = sio.flac_info(path)
    assert (sr, c, b, total) == (44100 if bits != 12 else 37123, nch, bits, len(x))
    v, c2, sr2, b2 = sio.read_flac_int(path)                     # MD5 verification on (FLAC_VERIFY_MD5)
    assert v.dtype == (np.int16 if bits <= 16 else np.int32) and (c2, sr2, b2) == (nch, sr, bits)
    np.testing.assert_array_equal(v.reshape(-1, nch), x)
    y, sr3 = sio.read_audio(path)                                 # librosa.load(file, sr=None): float32 mono
    want = (x.astype(np.float32) * np.float32(1.0 / (1 << (bits - 1))))
    want = want[:, 0] if nch == 1 else want.mean(axis=1).astype(np.float32)
    np.testing.assert_array_equal(y, np.ascontiguousarray(want, np.float32))


def test_every_subframe_kind_and_residual_option(tmp_path):
    x = _signal(24 * 512, 2, 16, seed=3)
    kinds = ["verbatim"] + [("fixed", o) for o in range(5)] + [("lpc", o) for o in (1, 2, 3, 5, 8, 12, 16, 24, 32)]

    def plan(fi, nch):
        k = kinds[fi % len(kinds)]
        return {"kinds": [k, kinds[(fi + 5) % len(kinds)]], "stereo": [0, 8, 9, 10][(fi // 3) % 4], "rice_method": fi % 2,
                "partition_order": [0, 1, 2, 3, 4][fi % 5], "escape_partition": [-1, 0, 1][fi % 3],
                "lpc_precision": [12, 15, 7][fi % 3], "lpc_shift": [9, 12, 4][fi % 3], "explicit_block_size": fi % 2 == 1,
                "explicit_bits": fi % 4 != 2}
    path = _write(tmp_path, "k.flac", FF.encode(x, 48000, 16, 512, plan=plan))
    v, nch, sr, bits = sio.read_flac_int(path)
    np.testing.assert_array_equal(v.reshape(-1, 2), x)


def test_wasted_bits_variable_blocking_and_long_frame_numbers(tmp_path):
    x = _signal(200 * 192 + 77, 1, 16, seed=5)
    x = (x >> 3) << 3                                             # three wasted bits in every sample
    path = _write(tmp_path, "w.flac", FF.encode(x, 16000, 16, 192, plan=lambda fi, nch: {"wasted": 3}))
    np.testing.assert_array_equal(sio.read_flac_int(path)[0], x[:, 0])
    path = _write(tmp_path, "v.flac", FF.encode(x, 16000, 16, 192, variable=True))      # sample numbers up to 38,400: 3-byte codes
    np.testing.assert_array_equal(sio.read_flac_int(path)[0], x[:, 0])
    path = _write(tmp_path, "r.flac", FF.encode(x[:5000], 12345, 16, 1024, plan=lambda fi, nch: {"explicit_sample_rate": True}))
    assert sio.flac_info(path)[0] == 12345
    np.testing.assert_array_equal(sio.read_flac_int(path)[0], x[:5000, 0])


def test_corruption_is_refused(tmp_path):
    x = _signal(6000, 2, 16, seed=9)
    good = FF.encode(x, 44100, 16, 1024)
    lib = _lib.load()
    n = C.c_int64()

    def rc(data, verify=1):
        p = _write(tmp_path, "c.flac", data)
        out = np.empty(len(x) * 2 + 64, np.int16)
        return lib.ssr_flac_decode_pcm16(os.fsencode(p), out.ctypes.data_as(C.c_void_p), out.size, verify, C.byref(n))
    assert rc(good) == 0 and n.value == len(x)
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x10                                    # one bit inside a frame: CRC-16 (or the bit stream itself) catches it
    assert rc(bytes(bad)) == _lib.ERR_INVALID_ARG
    md5 = bytearray(good)
    md5[8 + 18 + 3] ^= 0xff                                       # the signature in STREAMINFO
    assert rc(bytes(md5)) == _lib.ERR_INVALID_ARG and b"MD5" in lib.ssr_last_error()
    assert rc(bytes(md5), verify=0) == 0                          # ... which only verification looks at
    assert rc(good[:len(good) - 40]) == _lib.ERR_INVALID_ARG      # truncated
    assert rc(b"RIFF" + good[4:]) == _lib.ERR_INVALID_ARG         # not a FLAC stream
    with pytest.raises(_lib.SsrHipError):
        sio.read_flac_int(_write(tmp_path, "d.flac", bytes(bad)))
    small = np.empty(10, np.int16)
    p = _write(tmp_path, "e.flac", good)
    assert lib.ssr_flac_decode_pcm16(os.fsencode(p), small.ctypes.data_as(C.c_void_p), small.size, 1, C.byref(n)) == _lib.ERR_WORKSPACE
    p24 = _write(tmp_path, "f.flac", FF.encode(_signal(3000, 1, 24, 1), 48000, 24, 1024))
    assert lib.ssr_flac_decode_pcm16(os.fsencode(p24), None, 0, 1, C.byref(n)) == _lib.ERR_UNSUPPORTED


def test_streams_without_length_or_signature_and_with_an_id3_tag(tmp_path):
    x = _signal(7000, 1, 16, seed=11)
    data = bytearray(FF.encode(x, 22050, 16, 1152, md5=False))
    # zero the 36-bit sample count of STREAMINFO (a streamed encode): bytes 13 (low nibble) .. 17 of the block body
    body = 8
    data[body + 13] &= 0xf0
    for i in range(14, 18):
        data[body + i] = 0
    tag = b"ID3\x04\x00\x00" + bytes([0, 0, 0, 20]) + bytes(20)
    path = _write(tmp_path, "s.flac", tag + bytes(data))
    assert sio.flac_info(path) == (22050, 1, 16, 0)
    v, nch, sr, bits = sio.read_flac_int(path)
    np.testing.assert_array_equal(v, x[:, 0])
    r = sio.read_audio_raw(path)
    assert r.pcm is not None and r.sr == 22050 and r.n_frames == len(x)


def test_md5_implementation_matches_hashlib_through_the_signature_check(tmp_path):
    """Lengths around the 56- and 64-byte padding boundaries of RFC 1321."""
    for n in (1, 27, 28, 31, 32, 33, 63, 64, 65, 4097):
        x = _signal(n + 16, 1, 16, seed=n)[:n]
        path = _write(tmp_path, "m%d.flac" % n, FF.encode(x, 8000, 16, 16 if n < 100 else 1024))
        np.testing.assert_array_equal(sio.read_flac_int(path)[0], x[:, 0])


def test_decoder_survives_corrupted_streams(tmp_path):
    """Robustness of the file-ingest path: byte / bit flips, random 4-byte splices and truncations of valid streams of every sample
    width - the decoder returns an error or (where the damage missed the audio: padding, frame-size hints) the unchanged samples,
    never crashes, hangs or writes past `capacity`.  A truncated stream never passes (STREAMINFO's sample count)."""
    import ctypes as C
    from ssr_eval_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(1)
    streams, pcm = [], []
    for bits, nch, n in [(16, 1, 3000), (16, 2, 2000), (24, 1, 1500), (8, 2, 1200), (16, 1, 17)]:
        x = np.clip((rng.standard_normal((n, nch)) * (1 << (bits - 3))).astype(np.int64), -(1 << (bits - 1)), (1 << (bits - 1)) - 1)
        streams.append(FF.encode(x, 44100, bits=bits, block_size=576))
        pcm.append(x)
    path = str(tmp_path / "f.flac").encode()
    cap = 1 << 13
    out = np.full(cap + 64, 0x5a5a5a5a, np.int32)
    passed = {0: 0, 1: 0, 2: 0, 3: 0}
    for it in range(800):
        s = bytearray(streams[it % len(streams)])
        mode = it % 4
        if mode == 0:
            for _ in range(int(rng.integers(1, 6))):
                s[int(rng.integers(4, len(s)))] = int(rng.integers(0, 256))
        elif mode == 1:
            for _ in range(int(rng.integers(1, 6))):
                s[int(rng.integers(42, len(s)))] ^= 1 << int(rng.integers(0, 8))
        elif mode == 2:
            s = s[:int(rng.integers(8, len(s)))]
        else:
            p = int(rng.integers(4, len(s) - 8))
            s[p:p + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
        with open(path, "wb") as f:
            f.write(bytes(s))
        sr, ch, b, tot, md5 = C.c_int(), C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        rc = lib.ssr_flac_info(path, C.byref(sr), C.byref(ch), C.byref(b), C.byref(tot), C.byref(md5))
        if rc != 0:
            continue
        fr = C.c_int64()
        rc = lib.ssr_flac_decode_i32(path, out.ctypes.data_as(C.c_void_p), C.c_int64(cap), 1, C.byref(fr))
        assert np.all(out[cap:] == 0x5a5a5a5a), "wrote past capacity"
        if rc == 0:
            passed[mode] += 1
            x = pcm[it % len(streams)]
            np.testing.assert_array_equal(out[:x.size].reshape(x.shape), x)       # MD5 on: a pass means the very samples
    assert passed[2] == 0


def test_verify_flac_tree_tool(tmp_path, capsys):
    """tools/verify_flac_tree.py: the one command a holder of the VCTK test set runs before evaluate() - every stream decoded with
    MD5 verification, per-file format lines, exit status 0 on a clean tree, 1 when a stream's samples do not hash to its STREAMINFO
    digest (a flipped residual bit), 2 on a tree without FLAC files."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("verify_flac_tree", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                   "tools", "verify_flac_tree.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    (tmp_path / "p360").mkdir()
    (tmp_path / "p361").mkdir()
    a = FF.encode(_signal(6000, 1, 16, 1), 48000, 16, 1152)
    b = FF.encode(_signal(5000, 2, 24, 2), 44100, 24, 4096)
    (tmp_path / "p360" / "a.flac").write_bytes(a)
    (tmp_path / "p361" / "b.flac").write_bytes(b)
    assert tool.main([str(tmp_path)]) == 0
    out = capsys.readouterr().out
    assert "a.flac: 48000 Hz, 1 ch, 16 bit, 6000 frames" in out and "b.flac: 44100 Hz, 2 ch, 24 bit, 5000 frames" in out
    assert "2 file(s), 0 failed, 11000 frames" in out
    bad = bytearray(a)
    bad[len(bad) // 2] ^= 0x10                                   # inside a frame: CRC-16 / MD5 catch it
    (tmp_path / "p361" / "c.flac").write_bytes(bytes(bad))
    assert tool.main([str(tmp_path), "--quiet"]) == 1
    out = capsys.readouterr().out
    assert "FAIL" in out and "c.flac" in out and "3 file(s), 1 failed" in out
    (tmp_path / "empty").mkdir()
    assert tool.main([str(tmp_path / "empty")]) == 2


# ---- independent streams: the example files of RFC 9639, Appendix D (VERDICT r5 item 7 i; ssr_eval/eval.py:158-169 lists .flac) -------
def _crc(data, width, poly):
    c, top, mask = 0, 1 << (width - 1), (1 << width) - 1
    for x in data:
        c ^= x << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


def _rfc_examples():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rfc9639_examples.json")))["examples"]


@pytest.mark.parametrize("ex", _rfc_examples(), ids=lambda e: e["name"][:3])
def test_rfc9639_appendix_d_example_files(tmp_path, ex):
    """The decoder against streams it shares no author with: libFLAC's, as RFC 9639 prints them.  First the transcription is pinned
    WITHOUT the decoder (frame-header CRC-8, poly 0x07; frame CRC-16, poly 0x8005; STREAMINFO's MD5 of the expected little-endian
    PCM - all computed here), then ssr_flac_info / ssr_flac_decode_* must return exactly that PCM."""
    import hashlib
    data = bytes.fromhex(ex["hex"])
    pcm = np.array(ex["pcm"], dtype=np.int64)
    assert data[:4] == b"fLaC" and (data[4] & 0x7F) == 0 and int.from_bytes(data[5:8], "big") == 34
    si = data[8:8 + 34]
    bytes_per = (ex["bits"] + 7) // 8
    raw = b"".join(int(v).to_bytes(bytes_per, "little", signed=True) for v in pcm.reshape(-1))
    assert hashlib.md5(raw).digest() == si[18:34]                       # the RFC's own statement of the decoded audio
    ends = ex["frames_at"][1:] + [len(data)]
    for a, b in zip(ex["frames_at"], ends):
        fr = data[a:b]
        assert fr[0] == 0xFF and (fr[1] & 0xFE) == 0xF8
        hdr = 4 + 1 + (1 if (fr[2] >> 4) == 6 else 2 if (fr[2] >> 4) == 7 else 0)      # (frame numbers < 128: one UTF-8 byte)
        assert _crc(fr[:hdr], 8, 0x07) == fr[hdr]
        assert _crc(fr[:-2], 16, 0x8005) == int.from_bytes(fr[-2:], "big")
    path = _write(tmp_path, "rfc.flac", data)
    assert sio.flac_info(path) == (ex["rate"], ex["channels"], ex["bits"], len(pcm))
    v, nch, sr, bits = sio.read_flac_int(path)                             # MD5 verification on
    assert (nch, sr, bits) == (ex["channels"], ex["rate"], ex["bits"])
    np.testing.assert_array_equal(v.reshape(-1, nch), pcm)
    y, sr2 = sio.read_audio(path)
    want = pcm.astype(np.float32) * np.float32(1.0 / (1 << (bits - 1)))
    np.testing.assert_array_equal(y, want[:, 0] if nch == 1 else want.mean(axis=1).astype(np.float32))
    # a flipped bit anywhere in a frame is refused (CRC-16 / MD5), not decoded into something else
    bad = bytearray(data)
    bad[ex["frames_at"][0] + 9] ^= 0x10
    with pytest.raises(Exception):
        sio.read_flac_int(_write(tmp_path, "bad.flac", bytes(bad)))
