"""CPU tests: C-ABI surface, host-side integer logic and the reference-shaped Python API (no GPU compute)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ssr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ssr_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from ssr_eval_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 17
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    for n in names:
        assert hasattr(lib, n), n
    assert lib.ssr_version() >= 100


def test_no_kernel_a_reference_size_reaches_uses_scratch_memory():
    """the kernel descriptors inside the SHIPPED libssrhip.so (AMDGPU metadata notes of its code objects,
    tools/code_objects.py) - every wave-engine transform kernel (k_stft_wave: 2048/512; k_stft_rn_wave: 743, 1114, 1486;
    k_stft_r3_rot: 2229 = AudioMetrics(48000), float32 and float64-estimate variants) runs with private_segment_fixed_size == 0 and
    no spilled vector register, and so does every other kernel except the 8192-point Bluestein instances (n_fft 2049 .. 4096: no
    AudioMetrics(rate), FDomainHelper or BasicTestee size)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import code_objects
    from ssr_eval_amd import _lib
    ks = code_objects.kernels(_lib.LIB_PATH)
    names = code_objects.demangle([k["name"] for k in ks])
    assert len(ks) > 250                                     # every translation unit's bundle was found
    seen = {"k_stft_wave<double": 0, "k_stft_wave<float": 0, "k_stft_r3_rot<": 0, "k_stft_rn_wave<": 0, "k_ssim<": 0, "k_resample": 0,
            "k_tl_": 0, "k_sosfiltfilt": 0}
    for k, name in zip(ks, names):
        for key in seen:
            seen[key] += key in name
        if re.match(r"void k_stft<(double|float), 13, ", name):
            continue                                          # the 8192-point block engine: scratch by design, no reference size
        assert k["scratch"] == 0 and k["vgpr_spill"] == 0, (name, k)
        assert k["vgpr"] + k["agpr"] <= 512
    assert all(v > 0 for v in seen.values()), seen
    # the product instances of the API-true path, by their rocprofv3 names
    for want in ("k_stft_r3_rot<double, false, 3, 24, 0>", "k_stft_r3_rot<double, true, 3, 24, 0>", "k_stft_r3_rot<double, false, 3, 24, 1>",
                 "k_stft_r3_rot<double, true, 3, 24, 1>", "k_stft_wave<double, false, true, true>", "k_stft_wave<double, true, true, true>"):
        assert any(want in n for n in names), want


def test_product_does_not_import_oracle_or_fall_back():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ssr_eval_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
    import torch
    if not torch.cuda.is_available():
        from ssr_eval_amd import AudioMetrics
        from ssr_eval_amd._lib import SsrHipError
        with pytest.raises(SsrHipError):
            AudioMetrics(48000).evaluation(np.zeros(4000, np.float32), np.zeros(4000, np.float32), "")


def test_resample_plan_integers_match_scipy(golden):
    from ssr_eval_amd import _lib
    from oracle import resample as ors
    lib = _lib.load()
    for n, up, down in [(176400, 160, 147), (64000, 441, 160), (8000, 14553, 44100), (8000, 44100, 14553), (100, 4, 2),
                        (12345, 3, 7)]:
        v = [C.c_int(), C.c_int(), C.c_int64(), C.c_int(), C.c_int(), C.c_int()]
        assert lib.ssr_resample_plan(n, up, down, *[C.byref(x) for x in v]) == 0
        p = ors.poly_plan(n, up, down)
        assert (v[0].value, v[1].value, v[2].value, v[3].value, v[4].value, v[5].value) == (
            p["up"], p["down"], p["n_out"], p["half_len"], p["n_pre_pad"], p["n_pre_remove"])
    assert lib.ssr_resample_plan(10, 0, 1, *[None] * 6) != 0
    assert b"up and down" in lib.ssr_last_error()


def test_audio_metrics_integer_table(golden):
    from ssr_eval_amd import AudioMetrics
    for rate, (n_fft, hop) in zip(golden["a1_rates"], golden["a1_nfft_hop"]):
        am = AudioMetrics(int(rate))
        assert (am.n_fft, am.hop_length) == (int(n_fft), int(hop))


def test_cut_bins_and_keys(golden, golden_manifest):
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee
    from ssr_eval_amd.lowpass import cut_bin
    for hc, fs, cut in golden["lp_cut_table"]:
        assert cut_bin(int(hc) / int(int(fs) / 2)) == int(cut)
    user = {"cutoff_freq": [1000, 4000, 22050]}
    h = SSR_Eval_Helper(BasicTestee(), 44100, 44100, test_data_root=None, setting_fft=user)
    assert user["cutoff_freq"] == [2000, 8000, 44100]                 # caller's dict mutated, as in the reference
    keys, ratios = h._fft_plan_keys(44100)
    assert keys == golden_manifest["fft_keys"]
    assert [cut_bin(r) for r in ratios] == [46, 185, 1024]
    assert h.cache_file_name("proc_x", "/a/b/c.wav") == golden_manifest["cache_file_name"]
    np.testing.assert_array_equal(h.shift(np.arange(8.0), 3), golden["helper_shift_p3"])
    np.testing.assert_array_equal(h.shift(np.arange(8.0), -3), golden["helper_shift_m3"])
    a, b = h.pad(np.ones(3), np.ones(5))
    assert a.shape == b.shape == (5,) and a[3:].sum() == 0
    a, b = h.unify_length(np.ones(7), np.ones(5))
    assert a.shape == (5,)


def test_find_cutoff_matches_reference(golden):
    from ssr_eval_amd import BasicTestee
    bt = BasicTestee()
    got = [bt._find_cutoff(golden["bt_energy"], th) for th in (0.5, 0.9, 0.95, 0.97, 0.999)]
    np.testing.assert_array_equal(got, golden["bt_find_cutoff"])
    assert bt._find_cutoff(np.ones(5), 0.5) == 0
    assert bt.infer("x") == "x"


def test_aggregation_and_json_schema(tmp_path, monkeypatch):
    from datetime import datetime
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "aggregate.json")))
    h = SSR_Eval_Helper(BasicTestee(), 44100, 44100, test_data_root=None)
    work = [(s, f) for s in g["speakers"] for f in g["files"][s]]
    local = [g["per_file"][os.path.join(s, f)] for s, f in work]
    monkeypatch.chdir(tmp_path)
    final = h._assemble(work, g["speakers"], np.arange(len(work)), local, True, datetime(2022, 3, 28, 18, 7, 54, 109221))
    assert final["each_speaker"] == g["each_speaker"]
    assert final["averaged"] == g["averaged"]
    for s, f in work:
        assert final[s][f] == g["per_file"][os.path.join(s, f)]
    saved = json.load(open(tmp_path / "results" / "2022-03-28-18:07:54.109221-test.json"))
    assert saved["averaged"] == g["averaged"]
    flat = np.array([[v[k][m] for k in sorted(v) for m in ("lsd", "log_sispec", "sispec", "ssim")] for v in local])
    order = [k for k in local[0]]
    want = np.array([g["averaged"][k][m] for k in order for m in ("lsd", "log_sispec", "sispec", "ssim")])
    np.testing.assert_allclose(h.last_allreduce_average, want, rtol=1e-13)


def test_lowpass_dispatch_semantics():
    from ssr_eval_amd.lowpass import lowpass, bandpass, limit, align_length
    with pytest.raises(ValueError):
        lowpass(np.zeros((10, 1)), 1000, 44100, _type="stft_hard")
    with pytest.raises(ValueError):
        lowpass(np.zeros(10), 1000, 44100, _type="nope")
    with pytest.raises(ValueError):
        bandpass(np.zeros(10), 10, 1000, 44100, _type="stft_hard")
    assert (limit(12, 10, 2), limit(1, 10, 2), limit(5.7, 10, 2)) == (10, 2, 5)
    assert len(align_length(np.zeros(7), np.zeros(4))) == 7 and len(align_length(np.zeros(4), np.zeros(7))) == 4


def test_cabi_argument_validation_needs_no_gpu():
    """Every entry point rejects null / nonsensical arguments with SSR_ERR_INVALID_ARG before touching the device."""
    from ssr_eval_amd import _lib
    lib = _lib.load()
    assert lib.ssr_stft(None, None, None, None, None, 1, 4096, 1, None, None, None) == -1
    assert b"null" in lib.ssr_last_error()
    assert lib.ssr_pair_metrics(None, None, None, None, None, None, None, 1, 4096, 9, 15, None, None, 0, None) == -1
    assert lib.ssr_spectrogram_metrics(None, None, None, None, 1, 10, 10, 15, None, None, 0, None) == -1
    assert lib.ssr_fft_lowpass(None, None, None, None, None, None, 1, 4096, 9, None, None, 0, None) == -1
    assert lib.ssr_istft(None, None, None, None, None, None, 1, 4096, 9, None, None, 0, None) == -1
    assert lib.ssr_resample_poly(None, None, None, None, None, 1, 10, 2, 1, None, 5, 0, None, None) == -1
    assert lib.ssr_magphase(None, None, 10, 0.0, None, None, None, None) == -1
    h = C.c_void_p()
    assert lib.ssr_plan_create(1, 512, 1, C.byref(h)) == -1          # n_fft < 2
    assert lib.ssr_plan_create(2048, 0, 1, C.byref(h)) == -1         # hop < 1
    assert lib.ssr_plan_create(2048, 512, 7, C.byref(h)) == -1       # unknown precision
    assert lib.ssr_plan_create(5000, 512, 1, C.byref(h)) == -2       # Bluestein length would exceed 8192
    assert lib.ssr_plan_destroy(None) == 0
    assert lib.ssr_num_frames(None, 100) == -1
    assert lib.ssr_pair_metrics_workspace_bytes(None, 4, 4096, 36) == 0
    comm = C.c_void_p()
    assert lib.ssr_comm_unique_id(None) == -1 and lib.ssr_allreduce_sums(None, 4, None, None) == -1
    assert lib.ssr_comm_init_rank(None, 1, 0, C.byref(comm)) == -1
    uid = (C.c_char * 128)()
    assert lib.ssr_comm_init_rank(uid, 2, 2, C.byref(comm)) == -1 and b"rank" in lib.ssr_last_error()
    assert lib.ssr_comm_destroy(None) == 0


def test_cabi_round3_entry_points_validate_before_the_device():
    """ssr_resample_sinc (time register length, table description), ssr_pcm16_to_float, ssr_sispec_multichannel and
    ssr_plan_set_lowpass_engine reject bad arguments on the host (fake non-null pointers are never dereferenced)."""
    from ssr_eval_amd import _lib
    lib = _lib.load()
    p = C.c_void_p(0x1000)
    # null argument
    assert lib.ssr_resample_sinc(None, p, p, p, p, 1, 100, p, 100, p, p, 8193, 512, 512, 1.0, 1.0, 1, p, None) == -1
    # time register shorter than the longest output
    assert lib.ssr_resample_sinc(p, p, p, p, p, 1, 100, p, 99, p, p, 8193, 512, 512, 1.0, 1.0, 1, p, None) == -1
    assert b"time_register" in lib.ssr_last_error()
    # nonsensical filter description (no table entries / zero step)
    assert lib.ssr_resample_sinc(p, p, p, p, p, 1, 100, p, 100, p, p, 0, 512, 512, 1.0, 1.0, 1, p, None) == -1
    assert lib.ssr_resample_sinc(p, p, p, p, p, 1, 100, p, 100, p, p, 8193, 512, 0, 1.0, 1.0, 1, p, None) == -1
    assert lib.ssr_pcm16_to_float(None, p, p, p, 1, 10, p, p, None) == -1
    assert lib.ssr_pcm16_to_float(p, p, p, p, 0, 10, p, p, None) == 0           # empty batch: nothing to do
    assert lib.ssr_sispec_multichannel(None, p, 1, 2, 10, 0, p, p, 1024, None) == -1
    assert lib.ssr_sispec_multichannel(p, p, 0, 2, 10, 0, p, p, 1024, None) == -1
    assert lib.ssr_sispec_multichannel(p, p, 4, 2, 10, 0, p, p, 8, None) == -4  # workspace too small
    assert b"workspace" in lib.ssr_last_error()
    assert lib.ssr_plan_set_lowpass_engine(None, 0) == -1
    # matrix-core resampler: null argument; a plan whose tap table does not fit the kernel is refused before any device call
    assert lib.ssr_resample_poly_mfma(None, p, p, p, p, 1, 100, 441, 160, p, 8821, 50, p, None) == -1
    assert lib.ssr_resample_poly_mfma(p, p, p, p, p, 1, 100, 7349, 7350, p, 147001, 10, p, None) == -2
    assert b"ssr_resample_poly" in lib.ssr_last_error()


def test_cabi_round5_entry_points_validate_before_the_device():
    """ssr_fft_lowpass_multi and ssr_plan_set_tl_weights reject null arguments on the host, an empty batch / key list is a no-op, and the
    Python mirror's weight construction (backend.tl_conv_weights = torchlibrosa's numpy expressions) is the oracle's, bit for bit."""
    from ssr_eval_amd import _lib, backend as B
    from oracle import stft as ostft
    lib = _lib.load()
    p = C.c_void_p(0x1000)
    cuts = (C.c_int32 * 2)(10, 20)
    assert lib.ssr_fft_lowpass_multi(None, p, p, p, cuts, 2, p, 1, 100, 10, p, 100, p, 1 << 20, None) == -1
    assert lib.ssr_fft_lowpass_multi(p, p, p, p, None, 2, p, 1, 100, 10, p, 100, p, 1 << 20, None) == -1
    assert lib.ssr_fft_lowpass_multi(p, p, p, p, cuts, 0, p, 1, 100, 10, p, 100, p, 1 << 20, None) == 0     # no key: nothing to do
    assert lib.ssr_fft_lowpass_multi(p, p, p, p, cuts, 2, p, 0, 100, 10, p, 100, p, 1 << 20, None) == 0     # empty batch
    assert lib.ssr_plan_set_tl_weights(None, p, p, p, p, p) == -1
    assert lib.ssr_plan_set_tl_weights(p, None, p, p, p, p) == -1
    # ssr_sosfiltfilt_multi: null arguments, design count and section count limits, workspace size - all before any launch
    ns, eg = (C.c_int32 * 3)(1, 2, 4), (C.c_int32 * 3)(9, 15, 27)
    need = lib.ssr_sosfiltfilt_multi_workspace_bytes(1000, 2, eg, 3)
    assert need >= 8 * (3 * 1000 + 2 * 2 * (9 + 15 + 27)) and lib.ssr_sosfiltfilt_multi_workspace_bytes(1000, 2, None, 3) == 0
    assert lib.ssr_sosfiltfilt_multi(None, p, p, 2, 1000, p, p, ns, eg, 3, p, 1000, p, need, None) == -1
    assert lib.ssr_sosfiltfilt_multi(p, p, p, 2, 1000, p, p, None, eg, 3, p, 1000, p, need, None) == -1
    assert lib.ssr_sosfiltfilt_multi(p, p, p, 2, 1000, p, p, ns, eg, 0, p, 1000, p, need, None) == -2        # 1 .. 48 designs
    assert lib.ssr_sosfiltfilt_multi(p, p, p, 2, 1000, p, p, ns, eg, 49, p, 1000, p, need, None) == -2
    assert lib.ssr_sosfiltfilt_multi(p, p, p, 2, 1000, p, p, (C.c_int32 * 3)(1, 9, 4), eg, 3, p, 1000, p, need, None) == -2   # > 8 sections
    assert lib.ssr_sosfiltfilt_multi(p, p, p, 2, 1000, p, p, ns, eg, 3, p, 999, p, need, None) == -1         # y_stride < the batch
    assert lib.ssr_sosfiltfilt_multi(p, p, p, 2, 1000, p, p, ns, eg, 3, p, 1000, p, need - 1, None) == -4    # workspace too small
    assert lib.ssr_sosfiltfilt_multi(p, p, p, 0, 1000, p, p, ns, eg, 3, p, 1000, p, need, None) == 0         # empty batch
    got = B.tl_conv_weights(256)
    for a, b in zip(got[:4], ostft.tl_weights(256)):
        assert a.dtype == np.float32 and np.array_equal(a, b)
    assert got[0].shape == (129, 256) and got[2].shape == (256, 256) and got[4].shape == (256,)


def test_wav_decode_mono_stereo_and_batch(tmp_path):
    """io.read_audio on 16-bit mono / stereo PCM and the pooled decode_batch; Ragged.from_list packing on the host device."""
    import torch
    from ssr_eval_amd import backend as B
    from ssr_eval_amd.io import read_audio, write_wav, decode_batch
    r = B.Ragged.from_list([np.arange(4, dtype=np.float64), torch.arange(3, dtype=torch.float32), np.zeros(0, np.float32)], "cpu")
    assert r.data.dtype == torch.float32 and list(r.lens_host) == [4, 3, 0] and r.off.tolist() == [0, 4, 7]
    assert r.data.tolist() == [0, 1, 2, 3, 0, 1, 2]
    with pytest.raises(ValueError):
        B.Ragged.from_list([np.zeros((2, 2), np.float32)], "cpu")
    x = (0.25 * np.sin(np.arange(1000) / 7.0)).astype(np.float32)
    write_wav(str(tmp_path / "m.wav"), x, 16000)
    y, sr = read_audio(str(tmp_path / "m.wav"))
    assert sr == 16000 and y.dtype == np.float32 and np.abs(y - x).max() <= 0.5 / 32768   # rounded, not truncated
    # libsndfile's float -> PCM_16 with clipping on (what soundfile.write does for a .wav name): x 32768, nearest-even, saturate
    edge = np.array([1.5, 1.0, 32766.5 / 32768, 0.5 / 32768, 1.5 / 32768, -0.5 / 32768, -1.0, -2.0], np.float32)
    write_wav(str(tmp_path / "e.wav"), edge, 8000)
    q = np.rint(read_audio(str(tmp_path / "e.wav"))[0].astype(np.float64) * 32768).astype(int)
    assert q.tolist() == [32767, 32767, 32766, 0, 2, 0, -32768, -32768]
    import wave
    with wave.open(str(tmp_path / "s.wav"), "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(8000)
        f.writeframes(np.stack([np.full(10, 1000), np.full(10, 3000)], 1).astype("<i2").tobytes())
    ys, sr = read_audio(str(tmp_path / "s.wav"))
    assert sr == 8000 and np.allclose(ys, 2000 / 32768.0)
    both = decode_batch([str(tmp_path / "m.wav"), str(tmp_path / "s.wav")])
    assert np.array_equal(both[0][0], y) and both[1][1] == 8000


def test_length_balanced_sharding_of_the_vctk_shaped_set():
    """VERDICT r3 item 6 / SURVEY 8(e): dealing the 2,937 utterances of cfg-4 longest-first to the least-loaded rank keeps the
    heaviest shard within 1 % of the mean for 2, 4 and 8 ranks (round-robin: up to 3 %); every utterance has exactly one owner and
    the assignment is the same whichever rank computes it."""
    import bench
    from ssr_eval_amd import dist as D
    lens, _ = bench.Cfg4.layout()
    assert len(lens) == 2937
    for world in (1, 2, 3, 4, 8):
        shards = [D.shard_indices_balanced(lens, r, world) for r in range(world)]
        allidx = np.sort(np.concatenate(shards))
        np.testing.assert_array_equal(allidx, np.arange(len(lens)))
        loads = np.array([lens[s].sum() for s in shards], dtype=np.float64)
        assert loads.max() / loads.mean() <= 1.01, (world, loads.max() / loads.mean())
        assert all((np.diff(s) > 0).all() for s in shards if len(s) > 1)
    rr = np.array([lens[np.arange(r, len(lens), 8)].sum() for r in range(8)], dtype=np.float64)
    bal = np.array([lens[D.shard_indices_balanced(lens, r, 8)].sum() for r in range(8)], dtype=np.float64)
    assert bal.max() / bal.mean() < rr.max() / rr.mean()


def test_duration_hint_and_balanced_deal_of_a_file_tree(tmp_path):
    """evaluate(shard="balanced") deals by the duration in the file headers: WAV (RIFF data size) and FLAC (STREAMINFO) agree on
    what a second is, and the deal is a partition that every rank computes alike."""
    import flac_fixture as FF
    from ssr_eval_amd import dist as D
    from ssr_eval_amd.io import duration_hint, write_wav
    rng = np.random.default_rng(3)
    paths, secs = [], []
    for i, n in enumerate([44100, 22050, 132300, 8000, 66150, 99225]):
        x = (0.1 * rng.standard_normal(n)).astype(np.float32)
        p = str(tmp_path / ("f%d.%s" % (i, "flac" if i % 2 else "wav")))
        if i % 2:
            with open(p, "wb") as f:
                f.write(FF.encode(np.round(x * 32767).astype(np.int64), 44100, bits=16))
        else:
            write_wav(p, x, 44100)
        paths.append(p)
        secs.append(n / 44100.0)
    got = [duration_hint(p) for p in paths]
    np.testing.assert_allclose(got, secs, rtol=1e-9)
    w = [int(round(1000 * g)) for g in got]
    parts = [D.shard_indices_balanced(w, r, 2) for r in range(2)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(6))
    loads = [sum(w[i] for i in p) for p in parts]
    assert max(loads) / (sum(loads) / 2.0) < 1.1
    assert duration_hint(str(tmp_path / "missing.wav")) == 0.0


def test_cabi_round4_entry_points_validate_before_the_device():
    """ssr_resample_poly_chain (null arguments, bad plans, a geometry the fused kernel does not hold -> SSR_ERR_UNSUPPORTED with the
    reason), ssr_pair_metrics_multi, ssr_plan_create_ex and the FLAC entry points reject bad arguments on the host (fake non-null
    pointers are never dereferenced; no GPU needed)."""
    from ssr_eval_amd import _lib
    lib = _lib.load()
    p = C.c_void_p(0x1000)
    ok_chain = (p, p, p, p, p, p, 4, 1000, 441, 160, p, 8821, 28, 160, 147, p, 3201, 11, p, None)
    bad = list(ok_chain); bad[3] = None                                                   # mid_len missing
    assert lib.ssr_resample_poly_chain(*bad) == _lib.ERR_INVALID_ARG
    bad = list(ok_chain); bad[8] = 0                                                      # up1 = 0
    assert lib.ssr_resample_poly_chain(*bad) == _lib.ERR_INVALID_ARG
    bad = list(ok_chain); bad[6] = 0                                                      # empty batch: nothing to do
    assert lib.ssr_resample_poly_chain(*bad) == 0
    # 48 -> 44.1 -> 16 kHz (147/160 then 160/441): down-sampling plans have 221 and 441 taps per phase, not 21
    assert lib.ssr_resample_poly_chain(p, p, p, p, p, p, 4, 1000, 147, 160, p, 3201, 10, 160, 441, p, 8821, 10, p, None) == _lib.ERR_UNSUPPORTED
    assert b"21-tap" in lib.ssr_last_error()
    # 21-tap phases but the wrong block geometry (3/2 then 160/147: 8 x 3 != 24 x 147)
    assert lib.ssr_resample_poly_chain(p, p, p, p, p, p, 4, 1000, 441, 160, p, 8821, 28, 80, 147, p, 3201, 11, p, None) == _lib.ERR_UNSUPPORTED
    assert lib.ssr_pair_metrics_multi(None, p, p, p, p, p, p, 2, 3, 4096, 18, 15, p, p, 1 << 20, None) == _lib.ERR_INVALID_ARG
    assert lib.ssr_pair_metrics_multi_workspace_bytes(None, 2, 3, 4096, 18, 15) == 0
    # ... and its float64-estimate twin (round 6)
    assert lib.ssr_pair_metrics_multi_est64(None, p, p, p, p, p, p, 2, 3, 4096, 18, 15, p, p, 1 << 20, None) == _lib.ERR_INVALID_ARG
    assert lib.ssr_pair_metrics_multi_est64(p, None, p, p, p, p, p, 2, 3, 4096, 18, 15, p, p, 1 << 20, None) == _lib.ERR_INVALID_ARG
    assert lib.ssr_pair_metrics_multi_est64_workspace_bytes(None, 2, 3, 4096, 18, 15) == 0
    h = C.c_void_p()
    assert lib.ssr_plan_create_ex(1, 441, None, 1, 0, C.byref(h)) == _lib.ERR_INVALID_ARG        # n_fft < 2
    assert lib.ssr_plan_create_ex(2048, 441, None, 1, 7, C.byref(h)) == _lib.ERR_INVALID_ARG     # unknown pad mode
    assert lib.ssr_plan_create_ex(2048, 441, None, 1, 0, None) == _lib.ERR_INVALID_ARG
    sr, ch, b, md5, tot = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int64()
    assert lib.ssr_flac_info(b"/nonexistent/x.flac", C.byref(sr), C.byref(ch), C.byref(b), C.byref(tot), C.byref(md5)) != 0
    assert lib.ssr_flac_info(None, C.byref(sr), C.byref(ch), C.byref(b), C.byref(tot), C.byref(md5)) == _lib.ERR_INVALID_ARG


def test_pipeline_batches_cover_every_file_once_in_order():
    """evaluate()'s batching (round 5: two short pipeline-fill batches, then `step` files each): every file exactly once, in the
    reference's order (eval.py:171-199 walks speakers and files in sorted order), no empty batch, no batch above `step`."""
    from ssr_eval_amd.eval import pipeline_batches
    for n in (0, 1, 5, 63, 64, 65, 129, 200, 367, 2937):
        for step in (1, 5, 64, 128, 512):
            paths = list(range(n))
            batches = pipeline_batches(paths, step)
            assert [p for b in batches for p in b] == paths
            assert all(0 < len(b) <= step for b in batches)
            if n > step and step >= 4:
                assert len(batches[0]) == step // 4                 # the GPU starts after a quarter of a batch's reads
    assert [len(b) for b in pipeline_batches(list(range(367)), 64)] == [16, 32, 64, 64, 64, 64, 63]
