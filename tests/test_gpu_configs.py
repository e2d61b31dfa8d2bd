"""GPU parity tests (-m gpu) at BASELINE.json's configurations (cfg-2 at the full 1024-pair batch, the cfg-3 cutoff sweep
at 4 s @ 48 kHz, cfg-5's chain), the round-2 reference vectors, and an RCCL world-size-1 smoke of the collectives.
Everything goes ctypes -> libssrhip.so; nothing here reads /root/reference."""
import json
import os
import subprocess
import sys

import numpy as np
import conftest
import pytest
import torch
from scipy import signal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("lsd", "log_sispec", "sispec", "ssim")
CUTOFFS = [1000, 2000, 4000, 6000, 8000, 12000, 16000]
CUT_BINS = [42, 85, 170, 256, 341, 512, 683]                      # int(1025 * c / 24000)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    from ssr_eval_amd import _lib
    _lib.load()


def _vec(d):
    return np.array([d[k] for k in KEYS])


def _oracle_pair(args):
    """(est, target) -> (reference float32 metrics, float64 evaluation of the two SISpec terms); runs in a worker."""
    torch.set_num_threads(1)
    from oracle import metrics as om
    est, tgt = args
    want, exact = om.evaluation_with_exact(est, tgt, n_fft=2048, hop=512)
    return _vec(want), np.array([exact["log_sispec"], exact["sispec"]])


def _oracle_many(pairs):
    import multiprocessing as mp
    cores = min(os.cpu_count() or 1, 32, len(pairs))
    if cores >= 4:
        try:
            with mp.get_context("fork").Pool(cores) as pool:
                return pool.map(_oracle_pair, pairs, chunksize=1)
        except Exception:
            pass
    return [_oracle_pair(p) for p in pairs]


def _check_rows(got, oracle_rows, what):
    from test_gpu_parity import assert_sispec_parity
    for i, (g, (want, exact)) in enumerate(zip(got, oracle_rows)):
        np.testing.assert_allclose(g[[0, 3]], want[[0, 3]], rtol=1e-5, err_msg="%s pair %d (lsd, ssim)" % (what, i))
        assert_sispec_parity(g[1], want[1], exact[0], "%s pair %d log_sispec" % (what, i))
        assert_sispec_parity(g[2], want[2], exact[1], "%s pair %d sispec" % (what, i))


# ---- cfg-2 at the real batch: 1024 pairs (launch geometry: 94 frames per workgroup, 4 chunks per pair) ------------------
def test_cfg2_full_batch_1024_pairs():
    from ssr_eval_amd import backend as B
    N, n = 1024, 192000
    g = torch.Generator(device="cuda").manual_seed(20220328)
    tgt = (0.1 * torch.randn((N, n), generator=g, device="cuda", dtype=torch.float32)).contiguous()
    est = (tgt + 0.01 * torch.randn((N, n), generator=g, device="cuda", dtype=torch.float32)).contiguous()
    plan = B.get_plan(2048, 512, "f64")
    assert plan.frames(n) == 376 and plan.n_bins == 1025
    batch = B.PairBatch(plan, B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
    full = batch.run(B.M_ALL).cpu().numpy().copy()
    assert full.shape == (N, 4) and np.isfinite(full).all()
    spots = [0, 1, 255, 256, 511, 512, 777, 1023]                   # spread over the whole grid
    if (os.cpu_count() or 1) >= 16:                                 # a 16-core box checks 32 of the 1024 pairs
        spots = sorted(set(spots) | set(range(17, 1024, 43)))[:32]
    _check_rows(full[spots], _oracle_many([(est[i].cpu().numpy(), tgt[i].cpu().numpy()) for i in spots]), "cfg2")
    # the bench's mask (LSD + SSIM) gives the same two numbers, the other two stay NaN
    two = batch.run(B.M_LSD | B.M_SSIM).cpu().numpy()
    np.testing.assert_allclose(two[:, [0, 3]], full[:, [0, 3]], rtol=1e-12)      # (two separately compiled kernel variants)
    assert np.isnan(two[:, [1, 2]]).all()
    # batch-composition invariance: a pair alone (6 frames per workgroup) equals the pair inside the 1024 batch
    for i in (0, 511, 1023):
        alone = B.pair_metrics(plan, [est[i]], [tgt[i]])[0]
        np.testing.assert_allclose(alone, full[i], rtol=1e-12)
    # statistics of the synthetic workload: i.i.d. pairs -> tight spread (a wrong chunk anywhere would stick out)
    assert full[:, 0].std() / full[:, 0].mean() < 0.01 and full[:, 3].std() < 0.01


# ---- cfg-3: the cutoff sweep at 4 s @ 48 kHz, full metric set -------------------------------------------------------------
def test_cfg3_cutoff_sweep_full_metric_set():
    from ssr_eval_amd import backend as B
    import importlib
    L = importlib.import_module("ssr_eval_amd.lowpass")        # (the package attribute `lowpass` is the function)
    from oracle import lowpass as olp
    n_t, n = 16, 192000
    # cut bins: the dispatcher's integer arithmetic (lowpass.py:193-194 -> :24), bit-exact
    assert [L.cut_bin(c / int(48000 / 2)) for c in CUTOFFS] == CUT_BINS
    assert [olp.cut_bin(c, 48000) for c in CUTOFFS] == CUT_BINS
    g = torch.Generator(device="cuda").manual_seed(20220328)
    tgt = (0.1 * torch.randn((n_t, n), generator=g, device="cuda", dtype=torch.float32)).contiguous()
    # one ragged batch of 16 x 7 (utterance, cutoff) items
    rep = tgt.repeat_interleave(len(CUT_BINS), dim=0).contiguous()                  # item = t * 7 + c
    cuts = CUT_BINS * n_t
    assert L.DEFAULT_ENGINE == "conv"                           # the product's default: the reference's arithmetic class
    lp = B.LowpassBatch(B.get_plan(2048, 441, "f64", lowpass_engine=L.DEFAULT_ENGINE), B.Ragged.from_uniform(rep), cuts)
    est = lp.run().view(n_t * 7, n)
    torch.cuda.synchronize()
    est_h, tgt_h = est.cpu().numpy(), tgt.cpu().numpy()
    # the degradation itself against the oracle's torchlibrosa restatement (the published float32 dense-DFT convolutions on torch-CPU)
    for t, c in ((0, 0), (3, 3), (7, 5), (15, 6)):
        ref = olp.lowpass(tgt_h[t], CUTOFFS[c], 48000, 1, "stft_hard")
        np.testing.assert_allclose(est_h[t * 7 + c], ref, atol=4e-7)                # 0.1-amplitude noise: |y| up to ~0.5
    # metrics of every (degraded, target) pair: HIP vs the oracle on the SAME degraded signal
    plan = B.get_plan(2048, 512, "f64")
    got = B.PairBatch(plan, lp.out_ragged(), B.Ragged.from_uniform(rep)).run(B.M_ALL).cpu().numpy()
    assert np.isfinite(got).all()
    n_check = n_t if (os.cpu_count() or 1) >= 16 else 4                           # all 112 pairs where the host has the cores
    idx = [t * 7 + c for t in range(n_check) for c in range(7)]
    _check_rows(got[idx], _oracle_many([(est_h[i], tgt_h[i // 7]) for i in idx]), "cfg3")
    # LSD falls monotonically as the cutoff rises, for every target
    lsd = got[:, 0].reshape(n_t, 7)
    assert (np.diff(lsd, axis=1) < 0).all()
    # the helper's own route gives the same degraded signals (keys as the reference names them)
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee
    h = SSR_Eval_Helper(BasicTestee(), input_sr=48000, output_sr=48000, evaluation_sr=48000, test_data_root=None,
                        setting_fft={"cutoff_freq": list(CUTOFFS)})
    d = h.lowpass_stft_hard("", tgt_h[2], 48000)
    assert list(d.keys()) == ["proc_fft_%d_48000" % (2 * c) for c in CUTOFFS]
    for c, y in enumerate(d.values()):
        np.testing.assert_array_equal(y, est_h[2 * 7 + c])


MEMBER_BAR_DB = 1e-4        # a HIP value outside [min, max] of the reference's real members may be at most this far outside (dB)


def test_cfg3_sispec_inside_the_references_member_spread():
    """VERDICT r5 item 6 (ssr_eval/metrics.py:114-121, utils.py:68-92): cfg-3 shaped pairs - 4 s @ 48 kHz noise targets, estimate = the
    STFT-domain low-pass at the seven cutoffs - against the REFERENCE'S OWN sispec evaluated at 1 / 2 / 4 / 8 / 16 torch threads x
    {the transposed layout the reference builds, contiguous} (tools/exp_sispec_members.py imports /root/reference in the build
    container; tests/golden/sispec_members.json holds the members' values, data only).  Measured there: the members differ from each
    other by 2e-6 .. 4.5e-5 dB and share their float32 ELEMENTWISE roundings (scaled = dot * t / norm, noise = est - scaled), so all of
    them sit on one side of the float64 evaluation of the same formula in 38 of 56 values, up to 4.3e-5 dB away.  The HIP value is that
    float64 evaluation (to 1e-6): it must lie inside [min, max] of the members or at most MEMBER_BAR_DB = 1e-4 dB (2.3e-5 relative on
    the energy ratio) outside, and within 1e-4 dB of the reference at its default 8 threads.  The relative figures quoted for cfg-3
    (up to 5e-2) are these same absolute differences divided by SISpec values near 0 dB (cut 12 kHz: +0.03 dB)."""
    import json
    from ssr_eval_amd import backend as B
    import importlib
    L = importlib.import_module("ssr_eval_amd.lowpass")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "sispec_members.json")))["cases"]
    seeds = sorted({c["target_seed"] for c in gold})
    n = 192000
    tg = np.stack([(0.1 * np.random.default_rng(sd).standard_normal(n)).astype(np.float32) for sd in seeds])
    tgt = torch.from_numpy(tg).cuda()
    rep = tgt.repeat_interleave(len(CUT_BINS), dim=0).contiguous()
    lp = B.LowpassBatch(B.get_plan(2048, 441, "f64", lowpass_engine=L.DEFAULT_ENGINE), B.Ragged.from_uniform(rep), CUT_BINS * len(seeds))
    est = lp.run().view(len(seeds) * 7, n)
    got = B.PairBatch(B.get_plan(2048, 512, "f64"), lp.out_ragged(), B.Ragged.from_uniform(rep)).run(B.M_ALL).cpu().numpy()
    est_h = est.cpu().numpy()
    for c in gold:
        row = seeds.index(c["target_seed"]) * 7 + CUTOFFS.index(c["cutoff_hz"])
        # the SAME degraded signal the members were evaluated on (the conv engine is the multi-threaded conv1d member bit for bit)
        assert int(np.abs(est_h[row]).sum(dtype=np.float64) * 1e6) % (1 << 31) == c["est_crc"], ("low-passed signal differs", c["cutoff_hz"])
        for col, name in ((2, "sispec"), (1, "log_sispec")):
            m, hip = c[name], float(got[row, col])
            dist = max(m["min"] - hip, hip - m["max"], 0.0)
            conftest.MEMBER_LOG.append({"what": "target %d cut %d Hz %s" % (c["target_index"], c["cutoff_hz"], name), "hip_db": hip,
                                        "min_db": m["min"], "max_db": m["max"], "spread_db": m["spread_db"], "distance_db": dist,
                                        "exact_db": m["exact"], "err_vs_reference_db": abs(hip - m["reference_t8"])})
            assert abs(hip - m["exact"]) <= 1e-6 * abs(m["exact"]) + 1e-6, (name, c["cutoff_hz"], hip, m["exact"])
            assert dist <= MEMBER_BAR_DB, (name, c["cutoff_hz"], hip, m["min"], m["max"])
            assert abs(hip - m["reference_t8"]) <= MEMBER_BAR_DB, (name, c["cutoff_hz"], hip, m["reference_t8"])


def _oracle_conv_pipeline(args):
    """(target, cutoff Hz) -> metrics of (published-torchlibrosa low-pass of the target, target) at 2048/512; runs in a forked
    worker, ONE thread (an OpenMP team in a forked child of a process that has run one deadlocks): torch's single-threaded
    convolution blocks the inverse product by 384 / 448 channels - a member of the class up to 0.5 % (LSD) from the multi-threaded
    one the HIP engine reproduces bit for bit (tests/test_oracle.py), so this leg is held to the class bar; the strict leg of the
    test runs in the parent."""
    torch.set_num_threads(1)
    from oracle import lowpass as olp, metrics as om
    tgt, hc = args
    est = olp.lowpass(tgt, hc, 48000, 1, "stft_hard")            # arithmetic="conv": the reference's class
    return _vec(om.evaluation(est, tgt, n_fft=2048, hop=512))


def test_cfg3_pipeline_against_reference_arithmetic():
    """the PIPELINE, not the stages - HIP low-pass -> HIP metrics against oracle low-pass in the published
    torchlibrosa arithmetic (float32 conv1d on torch-CPU) -> oracle metrics, cfg-3's 16 targets x 7 cutoffs of 4 s @ 48 kHz, at the
    tolerance the class itself defines (conftest.assert_metrics_in_lowpass_class; CPU measurement:
    tests/test_oracle.py::test_lowpass_arithmetic_class_sensitivity).  The conv engine (the default of lowpass(_type="stft_hard"))
    has to be inside; the float64 FFT engine and the float32 FFT engine are measured next to it and the float64 one has to be
    OUTSIDE (LSD > 1.5 % off somewhere): it is the exact low-pass, not the reference's.  Deviations -> gpurun_out/r05_cfg3_engines.json."""
    import multiprocessing as mp
    from ssr_eval_amd import backend as B
    n_t, n = 16, 192000
    g = torch.Generator(device="cuda").manual_seed(20220328)
    tgt = (0.1 * torch.randn((n_t, n), generator=g, device="cuda", dtype=torch.float32)).contiguous()
    tgt_h = tgt.cpu().numpy()
    n_check = n_t if (os.cpu_count() or 1) >= 16 else 3
    jobs = [(tgt_h[t], CUTOFFS[c]) for t in range(n_check) for c in range(7)]
    cores = min(os.cpu_count() or 1, 32, len(jobs))
    if cores >= 4:
        with mp.get_context("fork").Pool(cores) as pool:
            want = np.array(pool.map(_oracle_conv_pipeline, jobs, chunksize=1))
    else:
        want = np.array([_oracle_conv_pipeline(j) for j in jobs])
    rep = tgt[:n_check].repeat_interleave(7, dim=0).contiguous()
    cuts = CUT_BINS * n_check
    mplan = B.get_plan(2048, 512, "f64")
    dev = {}
    for name, plan in (("conv", B.get_plan(2048, 441, "f64", lowpass_engine="conv")), ("f64_fft", B.get_plan(2048, 441, "f64")),
                       ("f32_fft", B.get_plan(2048, 441, "f32"))):
        lp = B.LowpassBatch(plan, B.Ragged.from_uniform(rep), cuts)
        lp.run()
        got = B.PairBatch(mplan, lp.out_ragged(), B.Ragged.from_uniform(rep)).run(B.M_ALL).cpu().numpy()
        assert np.isfinite(got).all()
        dev[name] = {"lsd_rel_max": float(np.abs(got[:, 0] / want[:, 0] - 1).max()), "lsd_rel_mean": float((got[:, 0] / want[:, 0] - 1).mean()),
                     "log_sispec_abs_max_db": float(np.abs(got[:, 1] - want[:, 1]).max()),
                     "sispec_abs_max_db": float(np.abs(got[:, 2] - want[:, 2]).max()), "ssim_rel_max": float(np.abs(got[:, 3] / want[:, 3] - 1).max())}
        if name == "conv":
            for i, (gv, wv) in enumerate(zip(got, want)):
                conftest.assert_metrics_in_lowpass_class(gv, wv, "cfg3-pipeline target %d cutoff %d" % (i // 7, CUTOFFS[i % 7]))
    assert dev["conv"]["lsd_rel_max"] <= conftest.CLASS_LSD_RTOL
    assert dev["f64_fft"]["lsd_rel_max"] > conftest.CLASS_LSD_RTOL, dev      # the idealisation is NOT the reference's arithmetic
    # the STRICT leg (round 5): torch at >= 2 threads, in this process - the HIP low-pass of a 4 s target equals the published
    # torchlibrosa code's sample for sample, so the pipeline's four metrics meet the north_star bar, not a class bar
    from oracle import lowpass as olp, metrics as om
    strict = {"samples_differing": 0, "lsd_rel_max": 0.0, "items": 0}
    if torch.backends.cpu.get_cpu_capability() == "AVX512" and torch.__version__.startswith("2.10"):
        old = torch.get_num_threads()
        torch.set_num_threads(max(2, min(old, 16)))
        try:
            conv_plan = B.get_plan(2048, 441, "f64", lowpass_engine="conv")
            for t in range(2):
                ys = B.fft_lowpass_multi(conv_plan, [tgt[t]], CUT_BINS)
                for c, yk in enumerate(ys):
                    ref = olp.lowpass(tgt_h[t], CUTOFFS[c], 48000, 1, "stft_hard")
                    strict["samples_differing"] += int((yk[0].cpu().numpy() != ref).sum())
                    got = B.pair_metrics(mplan, [yk[0]], [tgt[t]])[0]
                    wv = _vec(om.evaluation(ref, tgt_h[t], n_fft=2048, hop=512))
                    strict["lsd_rel_max"] = max(strict["lsd_rel_max"], float(abs(got[0] / wv[0] - 1)))
                    strict["items"] += 1
                    np.testing.assert_allclose(got[[0, 3]], wv[[0, 3]], rtol=1e-5, err_msg="strict leg target %d cutoff %d" % (t, CUTOFFS[c]))
        finally:
            torch.set_num_threads(old)
        assert strict["samples_differing"] == 0, strict
    dev["conv_strict_leg(torch >= 2 threads, in process)"] = strict
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_cfg3_engines.json"), "w") as f:
        json.dump({"what": "cfg-3 pipeline (low-pass -> metrics 2048/512), %d targets x 7 cutoffs, deviation of each HIP low-pass engine from the "
                           "published torchlibrosa arithmetic on torch-CPU" % n_check, "engines": dev}, f, indent=1)
    print("cfg3 engine deviations:", json.dumps(dev))


@pytest.mark.parametrize("engine", ["segments", "conv"])
def test_cfg3_full_launch_1024_targets_spot_checks(engine):
    """cfg-3 at the bench's REAL launch geometry: 1024 targets of 4 s per ssr_fft_lowpass launch, every
    cutoff of the sweep, oracle spot checks on (target, cutoff) items spread over the grid - the degraded signal against the
    oracle's torchlibrosa restatement, the four metrics of the pair against the oracle on the same degraded signal.  Both the
    float64 FFT engine (what bench.py --config cfg3 times) and the conv engine (the product default; 15 GB of workspace here)."""
    from ssr_eval_amd import backend as B
    from oracle import lowpass as olp
    N, n = 1024, 192000
    g = torch.Generator(device="cuda").manual_seed(20220328)
    tgt = (0.1 * torch.randn((N, n), generator=g, device="cuda", dtype=torch.float32)).contiguous()
    tr = B.Ragged.from_uniform(tgt)
    lp = B.LowpassBatch(B.get_plan(2048, 441, "f64", lowpass_engine=engine), tr, [CUT_BINS[0]] * N)
    batch = B.PairBatch(B.get_plan(2048, 512, "f64"), lp.out_ragged(), tr)
    spots = {0: (0, 513), 1: (1023,), 2: (255, 768), 3: (511,), 4: (1, 1022), 5: (640,), 6: (127, 1023)}     # cutoff index -> targets
    pairs, rows = [], []
    for c, cut in enumerate(CUT_BINS):
        lp.set_cuts(cut)
        est = lp.run().view(N, n)
        got = batch.run(B.M_ALL).cpu().numpy().copy()
        assert np.isfinite(got).all()
        for t in spots[c]:
            e = est[t].cpu().numpy().copy()
            ref = olp.lowpass(tgt[t].cpu().numpy(), CUTOFFS[c], 48000, 1, "stft_hard")
            np.testing.assert_allclose(e, ref, atol=4e-7, err_msg="cfg3 target %d cutoff %d" % (t, CUTOFFS[c]))
            pairs.append((e, tgt[t].cpu().numpy()))
            rows.append(got[t])
        # every target is an i.i.d. draw: the batch statistics are tight at every cutoff (a wrong chunk would stick out)
        assert got[:, 0].std() / got[:, 0].mean() < 0.02 and got[:, 3].std() < 0.02
    assert len(pairs) == 11
    _check_rows(np.array(rows), _oracle_many(pairs), "cfg3-full-launch")


def test_cfg5_full_launch_2048_utterances_bit_exact():
    """cfg-5's chain at a launch that makes the persistent resampler ITERATE: 2048 utterances x 17 blocks
    = 34,816 work items over ~768 persistent workgroups - 45 rounds with the cross-round register prefetch for 441/160 and
    160/147 - against scipy.signal.resample_poly, bit for bit, after BOTH stages on utterances {0, mid, last}; then the
    LSD of those utterances against the oracle."""
    from ssr_eval_amd import backend as B
    from oracle import metrics as om
    N = 2048
    g = torch.Generator(device="cuda").manual_seed(20220328)
    x = (0.1 * torch.randn((N, 64000), generator=g, device="cuda", dtype=torch.float32)).contiguous()
    tgt = (0.1 * torch.randn((N, 192000), generator=g, device="cuda", dtype=torch.float32)).contiguous()
    s1 = B.ResampleBatch(B.Ragged.from_uniform(x), 44100, 16000)
    s2 = B.ResampleBatch(s1.out_ragged(), 48000, 44100)
    y1 = s1.run().view(N, 176400)
    y2 = s2.run().view(N, 192000)
    batch = B.PairBatch(B.get_plan(2048, 512, "f64"), s2.out_ragged(), B.Ragged.from_uniform(tgt))
    lsd = batch.run(B.M_LSD).cpu().numpy()[:, 0]
    for i in (0, 1, N // 2, N - 2, N - 1):
        r1 = signal.resample_poly(x[i].cpu().numpy(), 441, 160)
        np.testing.assert_array_equal(y1[i].cpu().numpy(), r1, err_msg="stage 1, utterance %d" % i)
        r2 = signal.resample_poly(r1, 160, 147)
        np.testing.assert_array_equal(y2[i].cpu().numpy(), r2, err_msg="stage 2, utterance %d" % i)
        want = float(om.lsd(om.wav_to_spectrogram(r2, 2048, 512), om.wav_to_spectrogram(tgt[i].cpu().numpy(), 2048, 512)))
        assert abs(lsd[i] - want) <= 1e-5 * want, (i, lsd[i], want)
    # the rest of the launch: every utterance is an i.i.d. draw, so one wrong block anywhere would show in the energies
    e1, e2 = (y1.double() ** 2).mean(dim=1), (y2.double() ** 2).mean(dim=1)
    assert float(e1.std() / e1.mean()) < 0.02 and float(e2.std() / e2.mean()) < 0.02
    assert np.isfinite(lsd).all() and lsd.std() / lsd.mean() < 0.01
    # the same launch through the fused kernel (ssr_resample_poly_chain: what bench.py --config cfg5 times): EVERY sample of
    # every utterance identical to the two-call result, compared on the device
    chain = B.ResampleChainBatch(B.Ragged.from_uniform(x), 16000, 44100, 48000, fused=True)
    yc = chain.run()
    assert chain.ran_fused is True and torch.equal(yc, s2.out)


def test_cfg5_bench_launch_12500_utterances_oracle_spot_checks():
    """cfg-5 at the bench's REAL launch: 12,500 utterances of 64,000 samples through both resampling
    stages (the residue-class kernel: one workgroup per utterance and stage, 50 / 150 blocks each) and the LSD against a 48 kHz
    target - the first, a middle and the last utterance bit-identical to scipy.signal.resample_poly after BOTH stages, their LSD at
    1e-5 against the oracle, and every utterance's energy in line (an i.i.d. draw: one wrong block anywhere would stick out)."""
    from ssr_eval_amd import backend as B
    from oracle import metrics as om
    N = 12500
    g = torch.Generator(device="cuda").manual_seed(20220329)
    x = (0.1 * torch.randn((N, 64000), generator=g, device="cuda", dtype=torch.float32)).contiguous()
    s1 = B.ResampleBatch(B.Ragged.from_uniform(x), 44100, 16000)
    s2 = B.ResampleBatch(s1.out_ragged(), 48000, 44100)
    y1 = s1.run().view(N, 176400)
    y2 = s2.run().view(N, 192000)
    spots = (0, 1, N // 2, N - 2, N - 1)
    tgt = (0.1 * torch.randn((len(spots), 192000), generator=g, device="cuda", dtype=torch.float32)).contiguous()
    lsd = B.PairBatch(B.get_plan(2048, 512, "f64"), B.Ragged.from_uniform(y2[list(spots)].contiguous()), B.Ragged.from_uniform(tgt)).run(B.M_LSD).cpu().numpy()[:, 0]
    for j, i in enumerate(spots):
        r1 = signal.resample_poly(x[i].cpu().numpy(), 441, 160)
        np.testing.assert_array_equal(y1[i].cpu().numpy(), r1, err_msg="stage 1, utterance %d" % i)
        r2 = signal.resample_poly(r1, 160, 147)
        np.testing.assert_array_equal(y2[i].cpu().numpy(), r2, err_msg="stage 2, utterance %d" % i)
        want = float(om.lsd(om.wav_to_spectrogram(r2, 2048, 512), om.wav_to_spectrogram(tgt[j].cpu().numpy(), 2048, 512)))
        assert abs(lsd[j] - want) <= 1e-5 * want, (i, lsd[j], want)
    e1, e2 = (y1.double() ** 2).mean(dim=1), (y2.double() ** 2).mean(dim=1)
    assert float(e1.std() / e1.mean()) < 0.02 and float(e2.std() / e2.mean()) < 0.02
    assert float((e1 / e1.mean() - 1).abs().max()) < 0.1 and float((e2 / e2.mean() - 1).abs().max()) < 0.1
    # the bench's launch of the FUSED chain (one workgroup of seven producer and five consumer waves per utterance, 52 iterations):
    # all 2.4 G output samples identical to the two-call result
    chain = B.ResampleChainBatch(B.Ragged.from_uniform(x), 16000, 44100, 48000, fused=True)
    yc = chain.run()
    assert chain.ran_fused is True and torch.equal(yc, s2.out)


@pytest.mark.parametrize("hop", [441, 512])
def test_lowpass_engines_fused_and_segments(hop, golden):
    """ssr_plan_set_lowpass_engine: the fused engine (k_lowpass_group: overlap-add inside the transform kernel) against the default
    one (paired segments + k_ola_paired) and the oracle, on a ragged batch - a signal of several rounds, one shorter than a round,
    one barely longer than the reflect pad - in low-pass and in ISTFT mode; its bits do not depend on the batch it runs in; and
    it refuses plans it does not cover."""
    from ssr_eval_amd import backend as B
    from ssr_eval_amd._lib import SsrHipError
    from oracle import lowpass as olp, stft as ostft
    rng = np.random.default_rng(hop)
    sigs = [(0.3 * rng.standard_normal(n)).astype(np.float32) for n in (150000, 5 * hop + 1030, 1025, 40000)]
    cuts = [300, 1025, 77, 512]
    seg, fus = B.Plan(2048, hop, "f64"), B.Plan(2048, hop, "f64").set_lowpass_engine("fused")
    ys = [y.cpu().numpy() for y in B.fft_lowpass(seg, sigs, cuts)]
    yf = [y.cpu().numpy() for y in B.fft_lowpass(fus, sigs, cuts)]
    for x, c, a, b in zip(sigs, cuts, ys, yf):
        want = olp.stft_hard_lowpass(x, (c + 0.5) / 1025, n_fft=2048, hop=hop, arithmetic="ideal")
        np.testing.assert_allclose(a, want, atol=5e-8)
        np.testing.assert_allclose(b, want, atol=5e-8)
        np.testing.assert_allclose(a, b, atol=1e-7)
    alone = B.fft_lowpass(fus, [sigs[0]], [cuts[0]])[0].cpu().numpy()          # another batch geometry: the same bits
    np.testing.assert_array_equal(alone, yf[0])
    re, im = B.stft(seg, [sigs[3]], kind="complex")
    for plan in (seg, fus):
        back = B.istft(plan, re, im, [len(sigs[3])])[0].cpu().numpy()
        np.testing.assert_allclose(back, sigs[3], atol=2e-6)
    # the golden low-pass vectors of the imported reference through the fused engine as well (FDomainHelper's 2048 / 441)
    if hop == 441:
        x = golden["lp_x"]
        for hc, fs in [(4000, 44100), (12000, 44100), (6000, 48000)]:
            y = B.fft_lowpass(fus, [x], [olp.cut_bin(hc, fs)])[0].cpu().numpy()
            np.testing.assert_allclose(y, olp.lowpass(x, hc, fs, 1, "stft_hard", arithmetic="ideal"), atol=3e-8)
            np.testing.assert_allclose(y, golden["lp_y_%d_%d" % (hc, fs)], atol=2e-7)      # the reference's own (float32 conv) vector
    with pytest.raises(SsrHipError):
        B.Plan(2048, 100, "f64").set_lowpass_engine("fused")
    with pytest.raises(SsrHipError):
        B.Plan(2048, 441, "f32").set_lowpass_engine("fused")


def test_cfg3_reference_vectors(golden_r2):
    """The sweep in small, against outputs of the imported reference (tests/golden/make_golden_r2.py)."""
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee, AudioMetrics
    x = golden_r2["c3_x"]
    h = SSR_Eval_Helper(BasicTestee(), input_sr=48000, output_sr=48000, evaluation_sr=48000, test_data_root=None,
                        setting_fft={"cutoff_freq": list(CUTOFFS)})
    d = h.lowpass_stft_hard("", x, 48000)
    assert list(d.keys()) == [str(k) for k in golden_r2["c3_keys"]]
    assert [int(c) for c in golden_r2["c3_cut_bins"]] == CUT_BINS
    from oracle import metrics as om
    from test_gpu_parity import assert_sispec_parity
    am_api, am_b = AudioMetrics(48000), AudioMetrics(48000, n_fft=2048, hop_length=512)
    for j, (k, y) in enumerate(d.items()):
        ref_y = golden_r2["c3_y_" + k]
        np.testing.assert_allclose(y, ref_y, atol=2e-7)
        # THE PIPELINE against the imported reference's own outputs: HIP conv low-pass -> HIP metrics vs reference low-pass (the
        # published torchlibrosa arithmetic) -> reference metrics, at the class tolerance (conftest.assert_metrics_in_lowpass_class)
        conftest.assert_metrics_in_lowpass_class(_vec(am_b.evaluation(y, x, "")), golden_r2["c3_metrics_2048_512"][j], "cfg3-vectors %s 2048/512" % k)
        conftest.assert_metrics_in_lowpass_class(_vec(am_api.evaluation(y, x, "")), golden_r2["c3_metrics_api2229"][j], "cfg3-vectors %s 2229/480" % k)
        # metrics on the REFERENCE's degraded signal (the stop band is round-off: it has to be the same round-off).
        # LSD / SSIM at 1e-5; the two SISpec terms against the reference's float32 value AND the float64 evaluation of the
        # same formula (log-SISpec sits near 0.5 dB here: the reference's own float32 sums move it by ~1e-5 relative)
        for am, key, nf, hp in ((am_b, "c3_metrics_2048_512", 2048, 512), (am_api, "c3_metrics_api2229", 2229, 480)):
            got, want = _vec(am.evaluation(ref_y, x, "")), golden_r2[key][j]
            np.testing.assert_allclose(got[[0, 3]], want[[0, 3]], rtol=1e-5)
            _, exact = om.evaluation_with_exact(ref_y, x, n_fft=nf, hop=hp)
            assert_sispec_parity(got[1], want[1], exact["log_sispec"], "cfg3-reference-vectors %s log_sispec" % k)
            assert_sispec_parity(got[2], want[2], exact["sispec"], "cfg3-reference-vectors %s sispec" % k)


# ---- multi-channel tensors on the metric API --------------------------------------------------------------------------------
def test_multichannel_tensor_reductions_match_reference_vectors(golden_r2):
    from ssr_eval_amd import AudioMetrics, utils as U
    am = AudioMetrics(44100)
    for dev in ("cpu", "cuda"):
        e, t = torch.tensor(golden_r2["mc_est"], device=dev), torch.tensor(golden_r2["mc_tgt"], device=dev)
        lsd = am.lsd(e, t)
        assert tuple(lsd.shape) == (2, 3, 1, 1) and lsd.dtype == torch.float32 and lsd.device.type == dev
        np.testing.assert_allclose(lsd.cpu().numpy(), golden_r2["mc_lsd"], rtol=1e-5)
        ss = am.ssim(e, t)
        assert tuple(ss.shape) == (2, 3, 1, 1) and ss.dtype == torch.float64
        np.testing.assert_allclose(ss.cpu().numpy(), golden_r2["mc_ssim"], rtol=2e-7)
        np.testing.assert_allclose(float(am.sispec(e, t)), float(golden_r2["mc_sispec"]), rtol=1e-5)
        np.testing.assert_allclose(float(am.sispec(U.to_log(e), U.to_log(t))), float(golden_r2["mc_log_sispec"]), rtol=1e-5)
        np.testing.assert_allclose(float(am.log_sispec(e, t)), float(golden_r2["mc_log_sispec"]), rtol=1e-5)
        _, ut = U.energy_unify(e, t)
        np.testing.assert_allclose(ut.cpu().numpy(), golden_r2["mc_energy_unify_tgt"], rtol=2e-6)


def test_multichannel_sispec_stays_accurate_at_very_high_snr():
    """C > 1 SISpec against the float64 evaluation of the reference formula from 40 dB to 140 dB: the kernel keeps its sums on
    est - target (the (See, Stt, Set) form lost tens of dB above 100 dB)."""
    from ssr_eval_amd import AudioMetrics
    from oracle import metrics as om
    am = AudioMetrics(44100)
    rng = np.random.default_rng(11)
    t = torch.tensor(np.abs(rng.standard_normal((2, 3, 40, 129))).astype(np.float32) + 0.05)
    n = torch.tensor(rng.standard_normal((2, 3, 40, 129)).astype(np.float32))
    for snr_db in (40, 80, 100, 120, 140):
        e64 = t.double() * (1.0 + 10.0 ** (-snr_db / 20.0) * n.double())
        e = e64.float()
        want = float(om.sispec_exact(e, t))
        got = float(B_sispec(am, e.cuda(), t.cuda()))
        assert abs(got - want) <= 1e-6 * abs(want), (snr_db, got, want)
        want_l = float(om.sispec_exact(om.to_log(e), om.to_log(t)))
        got_l = float(am.log_sispec(e.cuda(), t.cuda()).double())
        assert abs(got_l - want_l) <= 2e-6 * abs(want_l), (snr_db, got_l, want_l)


def B_sispec(am, e, t):
    from ssr_eval_amd import backend as B
    return B.sispec_multichannel(e, t, False)           # float64 (AudioMetrics.sispec rounds it to float32 like the reference)


# ---- subsampling at a rate pair whose reduced `up` is huge (7349 / 7350) ----------------------------------------------------
def test_subsampling_quirk_rate_pair_16k(golden_r2):
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee, lowpass
    from ssr_eval_amd import backend as B
    x = golden_r2["ssq_x"]
    h = SSR_Eval_Helper(BasicTestee(), input_sr=16000, output_sr=16000, evaluation_sr=16000, test_data_root=None,
                        setting_subsampling={"cutoff_freq": [8000]})
    d = h.lowpass_subsampling("", x, 16000)
    assert list(d.keys()) == [str(golden_r2["ssq_key"])] == ["proc_subsampling_15999_16000"]
    np.testing.assert_array_equal(d["proc_subsampling_15999_16000"], golden_r2["ssq_y"])       # bit-exact
    np.testing.assert_array_equal(lowpass(x, 7999, 16000, order=1, _type="subsampling"), golden_r2["ssq_y"])
    rng = np.random.default_rng(7349)
    for up, down, n in ((7349, 7350, 20000), (7350, 7349, 777), (11024, 11025, 5000)):
        s = (0.1 * rng.standard_normal(n)).astype(np.float32)
        np.testing.assert_array_equal(B.resample_poly([s], up, down)[0].cpu().numpy(), signal.resample_poly(s, up, down))
        s64 = s.astype(np.float64) * 1.0000001
        np.testing.assert_array_equal(B.resample_poly([s64], up, down)[0].cpu().numpy(), signal.resample_poly(s64, up, down))


# ---- FDomainHelper.spectrogram_to_wav on near-silent frames (torchlibrosa.magphase clamps the MAGNITUDE at 1e-10) -------------
def test_spectrogram_to_wav_near_silent_bins():
    from ssr_eval_amd.dsp import FDomainHelper
    from oracle import stft as ostft
    rng = np.random.default_rng(12)
    n = 6000
    x = (0.1 * rng.standard_normal(n)).astype(np.float32)
    x[2000:4200] *= 1e-9                                   # bins with magnitude between 1e-10 and 1e-5
    fh = FDomainHelper()
    wav = torch.tensor(x[None, None, :])
    sp = fh.wav_to_spectrogram(wav, eps=1e-8)
    y = fh.spectrogram_to_wav(wav, sp, length=n)
    assert tuple(y.shape) == (1, 1, n)
    re, im = ostft.tl_stft_conv(x[None, :])
    mag = np.sqrt(re ** 2 + im ** 2)
    den = np.clip(mag, np.float32(1e-10), np.inf)          # dsp.py:147-152 via torchlibrosa.magphase
    spn = np.clip(re ** 2 + im ** 2, np.float32(1e-8), np.inf) ** np.float32(0.5)
    want = ostft.tl_istft_conv(spn * (re / den), spn * (im / den), n)[0]
    quiet = slice(2600, 3600)
    assert np.abs(want[quiet]).max() > 1e-6                # the clamp at 1e-8 dominates there: phase errors would show
    # loud part: float32 dot-product round-off of the class; quiet part: every bin there is BELOW the float32 round-off of the
    # dense DFT against the loud neighbours (1e-10 signal under a ~3e-6 floor) - its phase is that noise's in the reference too
    np.testing.assert_allclose(y[0, 0].numpy()[:1500], want[:1500], atol=4e-7)
    np.testing.assert_allclose(y[0, 0].numpy()[4700:], want[4700:], atol=4e-7)
    assert np.abs(y[0, 0].numpy()[quiet]).max() < 3e-4
    y64 = FDomainHelper(engine="segments")
    sp64 = y64.wav_to_spectrogram(wav, eps=1e-8)
    w64 = y64.spectrogram_to_wav(wav, sp64, length=n)[0, 0].numpy()
    re_i, im_i = ostft.tl_stft_ideal(x[None, :])
    den_i = np.clip(np.sqrt(re_i ** 2 + im_i ** 2), np.float32(1e-10), np.inf)
    spn_i = np.clip(re_i ** 2 + im_i ** 2, np.float32(1e-8), np.inf) ** np.float32(0.5)
    want_i = ostft.tl_istft_ideal(spn_i * (re_i / den_i), spn_i * (im_i / den_i), n)[0]
    assert np.abs(want_i[quiet]).max() > 1e-6
    np.testing.assert_allclose(w64, want_i, atol=2e-7, rtol=1e-4)


# ---- RCCL smoke: the collectives of ssr_eval_amd.dist on a world of one GPU --------------------------------------------------
def test_nccl_world_size_one_collectives(tmp_path):
    code = r"""
import os, sys, json
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29%%03d" %% (os.getpid() %% 1000), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0)
from ssr_eval_amd import dist as D
assert D.rank_world() == (0, 1) and dist.get_backend() == "nccl"
red = D.allreduce_sums(np.arange(5.0))
t = torch.arange(6, dtype=torch.float64, device="cuda")
dist.all_reduce(t)                                   # RCCL kernel on the device
rows = np.arange(12.0).reshape(4, 3)
tab = D.allgather_rows(rows, [0, 1, 2, 3], 4)
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"red": red.tolist(), "t": t.cpu().tolist(), "tab": tab.tolist()}))
""" % ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["red"] == [0, 1, 2, 3, 4] and d["t"] == [0, 1, 2, 3, 4, 5]
    assert d["tab"] == np.arange(12.0).reshape(4, 3).tolist()


_TWO_RANK_EVAL = r"""
import os, sys, json
sys.path.insert(0, %r)
rank = int(sys.argv[1]); world = int(sys.argv[2]); root = sys.argv[3]; port = sys.argv[4]; shard = sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", LOCAL_WORLD_SIZE=str(world))
import torch
torch.cuda.set_device(0)                                  # both ranks share cuda:0; the exchange runs over gloo
from ssr_eval_amd import SSR_Eval_Helper, BasicTestee
from ssr_eval_amd import dist as D
if world > 1:
    D.init_from_env(backend="gloo")
h = SSR_Eval_Helper(BasicTestee(), input_sr=44100, output_sr=44100, evaluation_sr=48000, test_data_root=root,
                    setting_fft={"cutoff_freq": [4000, 12000]}, setting_lowpass_filtering={"filter": ["cheby"], "cutoff_freq": [6000], "filter_order": [6]})
res = h.evaluate(save_json=False, batch_files=5, shard=shard)
import threading
from ssr_eval_amd import io as sio
readers = sum(1 for t in threading.enumerate() if t.name.startswith("ssr-decode"))
print("RESULT" + json.dumps({"res": res, "allreduce_avg": h.last_allreduce_average.tolist(), "readers": readers, "limit": sio.decode_threads(),
                             "cores": sio.usable_cores()}))
if world > 1:
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()
"""


def test_evaluate_sharded_two_processes_equals_single_process(tmp_path):
    """The REAL SSR_Eval_Helper.evaluate() - file walk, round-robin shard, decode, degradations, kernels, all-gather of the
    per-utterance rows, float64 sums + counts all-reduce - under TWO processes that share cuda:0 and exchange over gloo,
    against the single-process run on the same wav tree: every per-file number, every per-speaker mean and the averaged
    block (both ranks bit-identical to each other; against one process to 1e-12: the batches differ) (ssr_eval/eval.py:180-216)."""
    from ssr_eval_amd.io import write_wav
    rng = np.random.default_rng(12)
    root = tmp_path / "vctk"
    for s, c in (("p360", 4), ("p361", 3), ("s5", 2)):
        (root / s).mkdir(parents=True)
        for i in range(c):
            n = int(rng.integers(int(0.6 * 44100), int(1.4 * 44100)))
            write_wav(str(root / s / ("u%02d.wav" % i)), 0.1 * rng.standard_normal(n), 44100)
    code = _TWO_RANK_EVAL % ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)

    def launch(rank, world, port, shard="round-robin"):
        return subprocess.Popen([sys.executable, "-c", code, str(rank), str(world), str(root), str(port), shard], stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True, env=env, cwd=str(tmp_path))

    def result(p):
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, err[-3000:]
        return json.loads([l for l in out.splitlines() if l.startswith("RESULT")][-1][6:])
    single = result(launch(0, 1, 0))
    port = 34500 + os.getpid() % 1000
    procs = [launch(r, 2, port) for r in range(2)]
    both = [result(p) for p in procs]
    procs = [launch(r, 2, port + 1, "balanced") for r in range(2)]            # dealt by header duration instead of round-robin
    balanced = [result(p) for p in procs]
    assert balanced[0]["res"] == balanced[1]["res"]
    assert len(single["res"]["p360"]) == 4 and set(single["res"]["averaged"]) == {"proc_ch_12000_6_44100", "proc_fft_8000_44100", "proc_fft_24000_44100"}
    # every rank returns the SAME assembled result, bit for bit (the exchange transports float64 rows unchanged) ...
    assert both[0]["res"] == both[1]["res"]

    def flat(d, pre=""):
        for k, v in d.items():
            if isinstance(v, dict):
                yield from flat(v, pre + k + "/")
            else:
                yield pre + k, v
    a, b = dict(flat(single["res"])), dict(flat(both[0]["res"]))
    assert list(a) == list(b)                                  # same schema, same key order
    # ... and it is the single-process result: identical where the reference holds float32 values (lsd / sispec of float32
    # pairs), within 1e-12 elsewhere - a batch of other files chunks the float64 partial sums differently (last-bit effects)
    worst = max(abs(a[k] - b[k]) / max(abs(a[k]), 1e-300) for k in a)
    assert worst <= 1e-12, worst
    assert sum(a[k] == b[k] for k in a) >= len(a) // 2
    np.testing.assert_allclose(both[0]["allreduce_avg"], single["allreduce_avg"], rtol=1e-12)
    c = dict(flat(balanced[0]["res"]))
    assert list(a) == list(c) and max(abs(a[k] - c[k]) / max(abs(a[k]), 1e-300) for k in a) <= 1e-12


def test_evaluate_sharded_eight_processes_share_the_host_cores(tmp_path):
    """the driver's 8-rank launch in small - EIGHT processes (gloo, sharing cuda:0) run the real evaluate() on one
    wav tree: every rank returns the single-process result (1e-12), and the job as a whole starts no more reader threads than
    max(ranks, the cores its affinity mask / cgroup grant) - each rank's pool is its share (ssr_eval_amd.io.decode_threads)."""
    from ssr_eval_amd.io import write_wav
    rng = np.random.default_rng(13)
    root = tmp_path / "vctk"
    for s, c in (("p360", 9), ("p361", 8), ("s5", 7)):
        (root / s).mkdir(parents=True)
        for i in range(c):
            write_wav(str(root / s / ("u%02d.wav" % i)), 0.1 * rng.standard_normal(int(rng.integers(int(0.5 * 44100), int(1.1 * 44100)))), 44100)
    code = _TWO_RANK_EVAL % ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "SSR_DECODE_THREADS"):
        env.pop(k, None)

    def launch(rank, world, port):
        return subprocess.Popen([sys.executable, "-c", code, str(rank), str(world), str(root), str(port), "balanced"], stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True, env=env, cwd=str(tmp_path))

    def result(p):
        out, err = p.communicate(timeout=1200)
        assert p.returncode == 0, err[-3000:]
        return json.loads([l for l in out.splitlines() if l.startswith("RESULT")][-1][6:])
    single = result(launch(0, 1, 0))
    port = 36500 + os.getpid() % 1000
    ranks = [result(p) for p in [launch(r, 8, port) for r in range(8)]]
    assert all(r["res"] == ranks[0]["res"] for r in ranks)

    def flat(d, pre=""):
        for k, v in d.items():
            if isinstance(v, dict):
                yield from flat(v, pre + k + "/")
            else:
                yield pre + k, v
    a, b = dict(flat(single["res"])), dict(flat(ranks[0]["res"]))
    assert list(a) == list(b) and max(abs(a[k] - b[k]) / max(abs(a[k]), 1e-300) for k in a) <= 1e-12
    assert all(1 <= r["readers"] <= r["limit"] for r in ranks)
    assert sum(r["readers"] for r in ranks) <= max(8, ranks[0]["cores"])
    assert single["limit"] == min(16, single["cores"])


def test_cabi_allreduce_sums_two_rank_communicator():
    """ssr_allreduce_sums over a TWO-rank RCCL communicator (one process per device) where the box has two devices."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one HIP device on this box: the two-rank RCCL communicator needs two (covered by the gloo tests on CPU)")
    code = r"""
import os, sys, ctypes as C
sys.path.insert(0, %r)
rank = int(sys.argv[1]); path = sys.argv[2]
import torch
torch.cuda.set_device(rank)
from ssr_eval_amd import _lib
lib = _lib.load()
uid = C.create_string_buffer(128)
if rank == 0:
    _lib.check(lib.ssr_comm_unique_id(uid)); open(path + ".tmp", "wb").write(uid.raw); os.rename(path + ".tmp", path)
else:
    import time
    while not os.path.exists(path): time.sleep(0.05)
    uid = C.create_string_buffer(open(path, "rb").read(), 128)
comm = C.c_void_p()
_lib.check(lib.ssr_comm_init_rank(uid, 2, rank, C.byref(comm)))
t = torch.arange(10, dtype=torch.float64, device="cuda") * (rank + 1)
_lib.check(lib.ssr_allreduce_sums(C.c_void_p(t.data_ptr()), 10, comm, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
assert t.cpu().tolist() == [3.0 * i for i in range(10)], t
_lib.check(lib.ssr_comm_destroy(comm))
print("OK")
""" % ROOT
    import tempfile
    path = os.path.join(tempfile.mkdtemp(), "uid")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0 and "OK" in out, err[-3000:]


def test_bench_cli_configs_smoke():
    """bench.py --config cfg3 / cfg5 at a small batch: one JSON line with the contract's fields."""
    for cfg, extra in (("cfg3", ["--pairs", "32"]), ("cfg5", ["--utterances", "64"]), ("cfg4", [])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "2", "--warmup", "1",
                            "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0 and d["config"]["workload"].startswith(cfg[:3] + "-" + cfg[3])
        assert d["scaling"] == ("strong" if cfg == "cfg4" else "weak")


def test_bench_under_two_ranks_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --steps K --warmup W` - the driver's command for its scaling
    runs - on the one-GPU box: both ranks on cuda:0, gloo for the collectives (--_share-gpu, a test hook: RCCL itself cannot run two ranks
    on one device).  What only exists under world_size > 1 runs for real here: the asynchronous all-reduce of the step's sums against the
    next step's kernels, the barrier + MAX-over-ranks timing, the cfg-4 strong-scaling side figure every rank takes part in, the rank-0
    report with each rank's own clock.  One JSON line from rank 0, the whole-job value = both ranks' pairs."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    port = 29700 + os.getpid() % 200
    for extra in ([], ["--config", "cfg4", "--collective", "allreduce"]):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                            "--_share-gpu"] + extra, capture_output=True, text=True, timeout=900, env=env)
        port += 1
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["cpu_baseline"] is None
        ranks = d["extra"]["ranks"]
        assert ranks["joined"] == 2 and len(ranks["ms_per_step_per_rank"]) == 2 and ranks["backend"] == "gloo"
        if not extra:
            assert d["scaling"] == "weak" and abs(d["value"] - 2 * 1024 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
            side = d["extra"]["cfg4_strong_scaling"]
            assert side["n_gpus"] == 2 and side["scaling"] == "strong" and side["value"] > 0
            assert d["extra"]["allreduce_payload_bytes_per_step"] == 24
        else:
            assert d["scaling"] == "strong" and abs(d["value"] - 2937 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]


# ---- N2 ingest: windowed-sinc (resampy kaiser_best) resampling and the batched file loader ---------------------------------------
@pytest.mark.parametrize("sr_orig,sr_new", [(44100, 48000), (48000, 44100), (48000, 16000), (16000, 44100), (44100, 16000), (22050, 48000)])
def test_sinc_resampler_bit_exact_vs_restatement_gpu(sr_orig, sr_new):
    from ssr_eval_amd import backend as B
    from oracle import resampy as orsy
    rng = np.random.default_rng(sr_orig + sr_new)
    sigs = [(0.2 * rng.standard_normal(n)).astype(np.float32) for n in (60000, 20000, 4321, 50, 1)]    # several blocks / one / a fraction
    got = B.resample_sinc(sigs, sr_orig, sr_new)
    for x, y in zip(sigs, got):
        want = orsy.librosa_resample_kaiser(x, sr_orig, sr_new)
        assert y.dtype == torch.float32 and tuple(y.shape) == want.shape          # ceil(n * ratio): fix_length applied
        np.testing.assert_array_equal(y.cpu().numpy(), want)                      # bit-identical to the NumPy restatement
    assert [tuple(y.shape) for y in B.resample_sinc(sigs, 48000, 48000)] == [x.shape for x in sigs]
    if (sr_orig, sr_new) in ((44100, 48000), (48000, 44100)):
        # a long signal: the accumulated time register drifts further, more waves find their lanes split between two neighbouring
        # table entries (round 5: those waves read both rows of the phase-major table instead of falling to the gather loop)
        x = (0.2 * rng.standard_normal(400000)).astype(np.float32)
        np.testing.assert_array_equal(B.resample_sinc([x], sr_orig, sr_new)[0].cpu().numpy(), orsy.librosa_resample_kaiser(x, sr_orig, sr_new))


def test_pcm16_upload_is_the_host_decode(tmp_path):
    """backend.upload_decoded: 16-bit PCM crosses the bus as int16 and is converted / mixed to mono by ssr_pcm16_to_float -
    bit-identical to the host decode (read_audio = soundfile's float32 read + librosa's channel mean), mono, stereo, 3 channels."""
    import wave
    from ssr_eval_amd import backend as B
    from ssr_eval_amd.io import read_audio, read_audio_raw
    rng = np.random.default_rng(8)
    paths = []
    for k, (nch, n) in enumerate([(1, 30001), (2, 12345), (3, 777), (1, 1), (2, 50000)]):
        p = str(tmp_path / ("f%d.wav" % k))
        with wave.open(p, "wb") as f:
            f.setnchannels(nch); f.setsampwidth(2); f.setframerate(44100)
            f.writeframes(rng.integers(-32768, 32768, n * nch).astype("<i2").tobytes())
        paths.append(p)
    raw = [read_audio_raw(p) for p in paths]
    assert all(r.pcm is not None for r in raw)
    for _ in range(3):                                   # (both staging arenas, and the reuse of the first)
        got = B.upload_decoded(raw)
        for p, g in zip(paths, got):
            x, sr = read_audio(p)
            assert sr == 44100 and g.dtype == torch.float32
            np.testing.assert_array_equal(g.cpu().numpy(), x)
    np.testing.assert_array_equal(raw[1].to_float(), read_audio(paths[1])[0])
    # the packed route of evaluate(): decoder threads read the data chunks straight into the page-locked arena; a 24-bit
    # file in the same batch rides along as a float32 item
    from ssr_eval_amd.io import decode_packed_async
    p24 = str(tmp_path / "f24.wav")
    with wave.open(p24, "wb") as f:
        f.setnchannels(1); f.setsampwidth(3); f.setframerate(48000)
        f.writeframes(rng.integers(0, 256, 3 * 999, dtype=np.uint8).tobytes())
    mixed = paths[:2] + [p24] + paths[2:]
    for _ in range(3):
        pb = decode_packed_async(mixed)()
        assert pb.other_idx == [2] and pb.srs == [44100, 44100, 48000, 44100, 44100, 44100]
        for p, g in zip(mixed, B.upload_decoded(pb)):
            np.testing.assert_array_equal(g.cpu().numpy(), read_audio(p)[0])


def test_resident_path_equals_ndarray_testee_path(tmp_path):
    """The identity testee's resident path (device tensors through degradation, infer, resampling, metrics) against a testee
    that insists on ndarrays (the reference's contract: D2H before infer, H2D after) and one that opts in to device tensors:
    the same numbers, bit for bit - with a float32 (FFT low-pass), a float64 (IIR) and a subsampling degradation."""
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee
    from ssr_eval_amd.io import write_wav
    rng = np.random.default_rng(21)
    root = tmp_path / "set"
    for s, c in (("p100", 3), ("p101", 2)):
        (root / s).mkdir(parents=True)
        for i in range(c):
            write_wav(str(root / s / ("u%d.wav" % i)), 0.1 * rng.standard_normal(int(rng.integers(30000, 70000))), 44100)
    seen = {"nd": 0, "dev": 0}

    class NdTestee(BasicTestee):
        def infer(self, x):
            assert isinstance(x, np.ndarray); seen["nd"] += 1
            return x.copy()

    class DevTestee(BasicTestee):
        accepts_device_tensors = True

        def infer(self, x):
            assert isinstance(x, torch.Tensor) and x.is_cuda; seen["dev"] += 1
            return x * 1.0, {"extra_metric": 1.5}

    def run(testee):
        h = SSR_Eval_Helper(testee, input_sr=44100, output_sr=44100, evaluation_sr=48000, test_data_root=str(root),
                            setting_fft={"cutoff_freq": [4000]}, setting_subsampling={"cutoff_freq": [8000]},
                            setting_lowpass_filtering={"filter": ["butter"], "cutoff_freq": [6000], "filter_order": [4]})
        assert h._testee_takes_device_tensors() == (not isinstance(testee, NdTestee))
        return h.evaluate(save_json=False)
    base, nd, dv = run(BasicTestee()), run(NdTestee()), run(DevTestee())
    assert seen["nd"] == 15 and seen["dev"] == 15                    # 5 files x 3 keys
    assert nd == base
    for spk in ("p100", "p101"):
        for f, keys in dv[spk].items():
            for k, v in keys.items():
                assert v.pop("extra_metric") == 1.5 and v == base[spk][f][k]


def test_pipelined_pass_equals_file_by_file_and_the_descriptor_ring_wraps(tmp_path):
    """Round 5's host pipeline: evaluate() queues a batch's launches and collects its metric values one batch later, uploads
    descriptors through a page-locked ring and the PCM on its own stream.  (1) Whatever the batching (1, 3, 64 files per launch
    sequence; file by file through evaluate_single, which waits for each file), every number is the same, bit for bit.  (2) The
    ring: thousands of uploads (several times its 4 MB) interleaved with kernels that read them arrive intact."""
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee, backend as B
    from ssr_eval_amd.io import write_wav
    rng = np.random.default_rng(77)
    root = tmp_path / "set"
    for s, c in (("p200", 6), ("p201", 5), ("s5", 3)):
        (root / s).mkdir(parents=True)
        for i in range(c):
            write_wav(str(root / s / ("u%d.wav" % i)), 0.1 * rng.standard_normal(int(rng.integers(30000, 90000))), 44100)
    h = SSR_Eval_Helper(BasicTestee(), input_sr=44100, output_sr=44100, evaluation_sr=48000, test_data_root=str(root),
                        setting_fft={"cutoff_freq": [4000, 12000]}, setting_subsampling={"cutoff_freq": [8000]})
    base = h.evaluate(save_json=False, batch_files=64)
    for bf in (1, 3, 5):
        assert h.evaluate(save_json=False, batch_files=bf) == base
    for spk in ("p200", "p201", "s5"):
        for f in sorted(os.listdir(root / spk)):
            assert h.evaluate_single(str(root / spk / f)) == base[spk][f]
    # (2)
    dev = B.default_device()
    ring = B._DescRing.get(torch.cuda.current_device())
    start_half, flips, kept = ring.half, 0, []
    for i in range(2600):                                           # 2600 x 4 KB = 2.5 rings
        a = rng.integers(-2 ** 31, 2 ** 31 - 1, size=512, dtype=np.int64)
        t = B._h2d(a, dev)
        if i % 2 == 0:
            t = t + 1                                               # a kernel behind the copy, on the same stream
            a = a + 1
        if i % 100 == 0 or i > 2590:
            kept.append((a, t))
        flips += ring.half != start_half; start_half = ring.half
    assert flips >= 4
    for a, t in kept:
        np.testing.assert_array_equal(t.cpu().numpy(), a)
    for dt in (np.int32, np.float64, np.float32):
        a = rng.standard_normal(37).astype(dt)
        np.testing.assert_array_equal(B._h2d(a, dev).cpu().numpy(), a)
    big = rng.integers(0, 100, size=B._DescRing.SIZE // 8 // 8 + 1, dtype=np.int64)   # above the ring's per-upload limit: the plain copy
    np.testing.assert_array_equal(B._h2d(big, dev).cpu().numpy(), big)
    # (3) the metric stage takes views with gaps where they lie (est[:m] / target[:m] cut samples off items of one buffer)
    plan = B.get_plan(2048, 512, "f64")
    lens = [30000, 41234, 25000, 52001]
    buf_e = 0.1 * torch.randn(sum(lens) + 64, device="cuda")
    buf_t = buf_e + 0.01 * torch.randn_like(buf_e)
    offs = np.concatenate(([0], np.cumsum(lens)[:-1]))
    cut = [0, 3, 1, 7]                                              # samples dropped at the end of each item
    ev = [buf_e[o:o + n - c] for o, n, c in zip(offs, lens, cut)]
    tv = [buf_t[o:o + n - c] for o, n, c in zip(offs, lens, cut)]
    r = B.Ragged.from_list(ev, dev, allow_gaps=True)
    assert not r.packed and r.data.data_ptr() == buf_e.data_ptr() and B.Ragged.from_list(ev, dev).data.data_ptr() != buf_e.data_ptr()
    with pytest.raises(ValueError):
        r.split()
    np.testing.assert_array_equal(B.pair_metrics(plan, ev, tv), B.pair_metrics(plan, [e.clone() for e in ev], [t.clone() for t in tv]))
    multi = B.pair_metrics_multi(plan, [ev, [e * 0.5 for e in ev]], tv)          # key 0: views with gaps, key 1: separate tensors -> copied
    np.testing.assert_array_equal(multi[:, 0], B.pair_metrics(plan, ev, tv))
    ev2 = torch.cat([torch.cat(ev), torch.cat([e * 0.5 for e in ev])])           # both keys in one buffer, key-major, then cut again
    o2 = np.concatenate(([0], np.cumsum([e.shape[0] for e in ev] * 2)[:-1]))
    views = [ev2[o:o + n - 2] for o, n in zip(o2, [e.shape[0] for e in ev] * 2)]
    tv2 = [t[:t.shape[0] - 2] for t in tv]
    got = B.pair_metrics_multi(plan, [views[:4], views[4:]], tv2)
    np.testing.assert_array_equal(got[:, 0], B.pair_metrics(plan, [v.clone() for v in views[:4]], [t.clone() for t in tv2]))
    np.testing.assert_array_equal(got[:, 1], B.pair_metrics(plan, [v.clone() for v in views[4:]], [t.clone() for t in tv2]))


def test_load_audio_is_librosa_load_shaped(tmp_path):
    """load_audio / load_audio_batch: decode + mono + kaiser_best to the requested rate (librosa.load semantics), batched."""
    from ssr_eval_amd.io import write_wav, load_audio, load_audio_batch, read_audio
    from oracle import resampy as orsy
    rng = np.random.default_rng(1)
    paths = []
    for i, (sr, n) in enumerate([(44100, 30000), (48000, 25000), (44100, 12345), (16000, 9000)]):
        p = str(tmp_path / ("f%d.wav" % i))
        write_wav(p, 0.3 * np.sin(2 * np.pi * 440 * np.arange(n) / sr) + 0.01 * rng.standard_normal(n), sr)
        paths.append(p)
    got = load_audio_batch(paths, 48000)
    for p, y in zip(paths, got):
        x, sr = read_audio(p)
        np.testing.assert_array_equal(y, orsy.librosa_resample_kaiser(x, sr, 48000))
        np.testing.assert_array_equal(load_audio(p, 48000), y)
    x, sr = read_audio(paths[0])
    np.testing.assert_array_equal(load_audio(paths[0]), x)                          # sr=None: native rate


@pytest.mark.gpu
def test_cabi_allreduce_sums_single_rank_communicator():
    """ssr_allreduce_sums (the path's one collective behind the C ABI, RCCL resolved with dlopen): a one-rank communicator
    on this GPU - unique id, init, in-place float64 sum (identity with one rank), destroy."""
    import ctypes as C
    import torch
    from ssr_eval_amd import _lib
    lib = _lib.load()
    torch.cuda.set_device(0)
    uid = (C.c_char * 128)()
    _lib.check(lib.ssr_comm_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(lib.ssr_comm_init_rank(uid, 1, 0, C.byref(comm)))
    assert comm.value
    buf = torch.tensor([1.5, -2.0, 1e-300, 3.0], dtype=torch.float64, device="cuda")
    want = buf.cpu().clone()
    _lib.check(lib.ssr_allreduce_sums(buf.data_ptr(), buf.numel(), comm, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), want)
    assert lib.ssr_allreduce_sums(buf.data_ptr(), 0, comm, None) == 0
    _lib.check(lib.ssr_comm_destroy(comm))


def test_full_size_properties_of_the_path():
    """Size-independent properties at BASELINE's full sizes (4 s @ 48 kHz; 512 / 1024 items per launch), where the oracle is
    too slow to cover every item: STFT -> ISTFT round trip and the low-pass with a cut beyond Nyquist are the identity,
    scaling by a power of two commutes with the low-pass bit for bit, a pair's metrics do not depend on where it sits in
    the batch, LSD(c t, t) = 2 |log10 c|, SISpec ignores the target's scale, SSIM(x, x) = 1."""
    from ssr_eval_amd import backend as B
    dev = torch.device("cuda", 0)
    n, n_s = 512, 192000
    g = torch.Generator(device=dev).manual_seed(99)
    x = (0.1 * torch.randn((n, n_s), generator=g, device=dev)).contiguous()
    # FDomainHelper sizes (2048 / 441): forward complex STFT, inverse STFT (the paired-segment overlap-add at full scale)
    plan_fd = B.get_plan(2048, 441, "f64", dev)
    sub = [x[i] for i in range(0, n, 4)]                              # 128 items: 0.9 GB of spectra
    re, im = B.stft(plan_fd, sub, kind="complex", torch_style_pad=True)
    back = B.istft(plan_fd, re, im, [n_s] * len(sub))
    worst = max(float((b - s).abs().max()) for b, s in zip(back, sub))
    assert worst <= 2e-6, worst
    del re, im, back
    rag = B.Ragged.from_uniform(x)
    ident = B.LowpassBatch(plan_fd, rag, [1025] * n).run().clone()
    assert float((ident - x.view(-1)).abs().max()) <= 2e-6
    lp = B.LowpassBatch(plan_fd, rag, [256] * n).run().clone()
    half = B.LowpassBatch(plan_fd, B.Ragged.from_uniform((0.5 * x).contiguous()), [256] * n).run()
    assert torch.equal(half, 0.5 * lp)                                 # exact: a power of two scales every intermediate
    assert float(lp.abs().max()) > 0.01 and float((lp - x.view(-1)).abs().max()) > 0.01
    # pair metrics (2048 / 512): position independence, closed forms
    plan = B.get_plan(2048, 512, "f64", dev)
    est = lp.view(n, n_s)
    fwd = B.PairBatch(plan, B.Ragged.from_uniform(est), rag).run(B.M_ALL).clone()
    rev = B.PairBatch(plan, B.Ragged.from_uniform(est.flip(0).contiguous()), B.Ragged.from_uniform(x.flip(0).contiguous())).run(B.M_ALL)
    assert torch.equal(fwd, rev.flip(0))
    scaled = B.PairBatch(plan, B.Ragged.from_uniform((0.25 * x).contiguous()), rag).run(B.M_ALL).cpu().numpy()
    np.testing.assert_allclose(scaled[:, 0], 2 * abs(np.log10(0.25)), rtol=1e-6)          # LSD(c t, t)
    assert (scaled[:, 2] > 100).all()                                 # est = c t: SISpec is round-off defined, but huge
    tgt2 = B.PairBatch(plan, B.Ragged.from_uniform(est), B.Ragged.from_uniform((3.0 * x).contiguous())).run(B.M_SISPEC).cpu().numpy()
    np.testing.assert_allclose(tgt2[:, 2], fwd[:, 2].cpu().numpy(), rtol=2e-6)            # target scale does not matter
    same = B.PairBatch(plan, rag, rag).run(B.M_SSIM | B.M_LSD).cpu().numpy()
    # (the two SSIM ratios are float32: 1 - 3e-8; LSD's 1e-12 guards leave ~1e-12 where a frame holds a tiny bin)
    assert (np.abs(same[:, 3] - 1.0) <= 1e-7).all() and (same[:, 0] <= 1e-9).all()


# ---- float32 transform precision: every pair engine with all four metrics ---------------------------------------------------
@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (2048, 441), (2229, 480), (1486, 320), (1114, 240), (743, 160), (1024, 256)])
def test_float32_precision_plans_all_metrics(n_fft, hop):
    """precision="f32" plans (float32 transform: not the reference's arithmetic, offered for speed) run the same engines with
    float32 exchange arrays; the running-sum accumulators behind those arrays are float64 and need their own alignment
    (a misaligned ds_add_f64 is a memory violation, which only this precision could hit).  Against the oracle (float64
    transform) the float32 transform stays within 5e-5 on these signals."""
    from ssr_eval_amd import backend as B
    from oracle import metrics as om
    rng = np.random.default_rng(n_fft * 7 + hop)
    plan = B.get_plan(n_fft, hop, "f32")
    ests, tgts = [], []
    for _ in range(3):
        n = int(rng.integers(20 * hop, 60 * hop))
        t = (0.1 * rng.standard_normal(n)).astype(np.float32)
        ests.append((t + 0.02 * rng.standard_normal(n)).astype(np.float32))
        tgts.append(t)
    got = B.pair_metrics(plan, ests, tgts)
    for i, (e, t) in enumerate(zip(ests, tgts)):
        want = _vec(om.evaluation(e, t, n_fft=n_fft, hop=hop))
        np.testing.assert_allclose(got[i], want, rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("hop", [441, 512])
def test_float32_precision_lowpass_and_istft(hop):
    """The wave low-pass / inverse-STFT kernels at precision="f32": the same signal as the float64 transform to float32
    round-off (the reference's FDomainHelper itself is a float32 torch pipeline, ssr_eval/dsp.py:107-119)."""
    from ssr_eval_amd import backend as B
    rng = np.random.default_rng(hop)
    dev = torch.device("cuda", 0)
    xs = [(0.1 * rng.standard_normal(int(rng.integers(30000, 90000)))).astype(np.float32) for _ in range(4)]
    out = {}
    for prec in ("f64", "f32"):
        plan = B.get_plan(2048, hop, prec, dev)
        out[prec] = B.LowpassBatch(plan, B.Ragged.from_list(xs, dev), [256] * len(xs)).run().clone()
        re, im = B.stft(plan, xs, kind="complex", torch_style_pad=True)
        back = B.istft(plan, re, im, [len(x) for x in xs])
        for b, x in zip(back, xs):
            assert float((b.cpu() - torch.from_numpy(x)).abs().max()) <= (2e-7 if prec == "f64" else 6e-7)
    assert float((out["f64"] - out["f32"]).abs().max()) <= 3e-7


# ---- ssr_pair_metrics_multi: one target, K estimates ------------------------------------------------------
def _multi_inputs(seed, lens, K):
    rng = np.random.default_rng(seed)
    tgts = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]
    if len(tgts) > 1:
        tgts[1][len(tgts[1]) // 3:len(tgts[1]) // 3 + 6000] = 0.0            # a silent stretch: zero-forced frames
    ests = [[(t * (0.5 + 0.1 * k) + (0.01 + 0.01 * k) * rng.standard_normal(len(t))).astype(np.float32) for t in tgts] for k in range(K)]
    if K > 2 and len(tgts) > 2:
        ests[2][2][:] = 0.0                                                     # an all-zero estimate
    return ests, tgts


@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (2229, 480), (1486, 320), (743, 160), (1000, 250), (4096, 1024)])
@pytest.mark.parametrize("K", [1, 2, 3, 4, 7])
def test_pair_metrics_multi_equals_k_calls_of_pair_metrics(n_fft, hop, K):
    """ssr_pair_metrics_multi against K calls of ssr_pair_metrics on the same inputs, every engine family (2048: k_stft_wave; 2229:
    rotating radix-3; 1486 / 743 / 1000: R waves; 4096: the block engine's plain-pass fallback), ragged items with a silent stretch
    and an all-zero estimate.  Key 0, an odd last key (both transformed WITH the target) and every key of the fallback reproduce
    ssr_pair_metrics' LSD / SISpec bit for bit; the estimates transformed two per complex transform agree to 1e-6
    (include/ssr_hip.h says why not bit for bit)."""
    from ssr_eval_amd import backend as B
    plan = B.get_plan(n_fft, hop, "f64")
    lens = (30000, 48000, 9000, 7 * hop + 100)                                # (the last one: seven frames, SSIM's minimum)
    ests, tgts = _multi_inputs(n_fft + K, lens, K)
    for mask in (B.M_ALL, B.M_LSD | B.M_SSIM, B.M_LSD, B.M_SISPEC | B.M_LOG_SISPEC):
        got = B.pair_metrics_multi(plan, ests, tgts, mask)
        assert got.shape == (len(lens), K, 4)
        for k in range(K):
            want = B.pair_metrics(plan, ests[k], tgts, mask)
            g = got[:, k]
            assert np.array_equal(np.isnan(g), np.isnan(want)), (k, mask)
            ok = ~np.isnan(want)
            if k == 0 or n_fft == 4096 or (k == K - 1 and K % 2 == 0):
                np.testing.assert_array_equal(g[:, :3][ok[:, :3]], want[:, :3][ok[:, :3]])
            # dB values: 1e-6 relative + 1e-6 dB; LSD / SSIM 1e-6 relative
            np.testing.assert_allclose(g[ok], want[ok], rtol=1e-6, atol=1e-6, err_msg="key %d mask %d" % (k, mask))


@pytest.mark.parametrize("n_fft,hop", [(2229, 480), (2048, 441), (1486, 320)])
@pytest.mark.parametrize("K", [2, 3, 6, 9])
def test_pair_metrics_multi_est64_equals_k_calls_of_est64(n_fft, hop, K):
    """ssr_pair_metrics_multi_est64 (K float64 IIR keys of a file against its one float32 target) against K calls of
    ssr_pair_metrics_est64: real zero-phase IIR estimates (stop bands at the float64 round-off floor - what makes the float64 path
    necessary), ragged items.  2229: the two-estimates-per-transform kernel of the rotating engine - key 0 and an odd last key bit for
    bit, the others to 1e-6 (include/ssr_hip.h says why); 2048 / 1486: K plain passes, every key bit for bit.  Also against the oracle."""
    from ssr_eval_amd import backend as B
    from oracle import lowpass as olp, metrics as om
    rng = np.random.default_rng(n_fft + K)
    lens = (30000, 41000, 9000, 7 * hop + 100)
    tgts = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]
    tgts[1][9000:15000] = 0.0                                                 # a silent stretch
    designs = [olp.iir_sos(hc, 44100, order, ft) for ft in ("butter", "cheby1", "ellip") for hc in (2000, 8000, 12000) for order in (2, 8)][:K]
    ests = [[signal.sosfiltfilt(sos, t) for t in tgts] for sos in designs]
    assert all(e.dtype == np.float64 for key in ests for e in key)
    plan = B.get_plan(n_fft, hop, "f64")
    for mask in (B.M_ALL, B.M_LSD | B.M_SSIM, B.M_LSD):
        got = B.pair_metrics_multi(plan, ests, tgts, mask)
        assert got.shape == (len(lens), K, 4)
        for k in range(K):
            want = B.pair_metrics(plan, ests[k], tgts, mask)
            g = got[:, k]
            assert np.array_equal(np.isnan(g), np.isnan(want)), (k, mask)
            ok = ~np.isnan(want)
            if k == 0 or n_fft != 2229 or (k == K - 1 and K % 2 == 0):
                np.testing.assert_array_equal(g[:, :3][ok[:, :3]], want[:, :3][ok[:, :3]])
            np.testing.assert_allclose(g[ok], want[ok], rtol=1e-6, atol=1e-6, err_msg="key %d mask %d" % (k, mask))
    got = B.pair_metrics_multi(plan, ests, tgts, B.M_ALL)
    for k in (0, 1, K - 1):
        for i in (0, 2):
            want = _vec(om.evaluation(ests[k][i], tgts[i], n_fft=n_fft, hop=hop))
            np.testing.assert_allclose(got[i, k][[0, 3]], want[[0, 3]], rtol=2e-6)
            # the reference's pow_p_norm(target) is a FLOAT32 torch.norm whose own summation error enters its float64 SISpec (here up to
            # 9e-6 dB on a log-SISpec of -0.5 dB: 1.8e-5 of the value, 2e-6 of the energy ratio; the kernels equal the float64 evaluation
            # of the formula): 1e-5 relative, or 5e-5 dB where the value is near 0 dB (SISpec accounting)
            np.testing.assert_allclose(got[i, k][[1, 2]], want[[1, 2]], rtol=1e-5, atol=5e-5)


def test_evaluate_arrays_scores_iir_keys_through_the_est64_multi_entry(monkeypatch):
    """SSR_Eval_Helper.evaluate_arrays with IIR keys (float64) next to FFT keys (float32): the keys of a file are split by dtype, each
    group goes through ONE multi launch sequence (ssr_pair_metrics_multi_est64 / ssr_pair_metrics_multi), and every value is what the
    per-pair path returns for that key."""
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee, backend as B
    seen = []
    orig = B.pair_metrics_multi

    def spy(plan, est_lists, tgt_list, *a, **k):
        seen.append((len(est_lists), bool(B._is_f64(est_lists[0][0]))))
        return orig(plan, est_lists, tgt_list, *a, **k)
    monkeypatch.setattr(B, "pair_metrics_multi", spy)
    h = SSR_Eval_Helper(BasicTestee(), input_sr=44100, output_sr=44100, evaluation_sr=48000, test_data_root=None,
                        setting_fft={"cutoff_freq": [2000, 8000]},
                        setting_lowpass_filtering={"filter": ["cheby", "butter", "bessel"], "cutoff_freq": [4000], "filter_order": [2, 6]})
    rng = np.random.default_rng(18)
    items = []
    for n in (20000, 31000, 26000):
        x = (0.1 * rng.standard_normal(n)).astype(np.float32)
        items.append((signal.resample_poly(x, 160, 147).astype(np.float32), x))
    res = h.evaluate_arrays(items)
    assert sorted(seen) == [(2, False), (6, True)] and all(len(r) == 8 for r in res)
    for (tgt, x), r in zip(items, res):
        d = h.preprocess_array(x, 44100)
        assert list(d.keys()) == list(r.keys())
        for key, y in d.items():
            y = np.asarray(y)
            assert (y.dtype == np.float64) == ("fft" not in key)
            y48 = signal.resample_poly(y, 160, 147)
            want = h.audio_metrics.evaluation(y48 if y.dtype == np.float64 else y48.astype(np.float32), tgt, "")
            np.testing.assert_allclose(_vec(r[key]), _vec(want), rtol=2e-6, atol=2e-6)


def test_cfg3_through_pair_metrics_multi_against_the_oracle():
    """cfg-3's sweep through the new entry: 16 targets x 7 cutoffs of 4 s @ 48 kHz, the seven low-passed estimates of a target written
    key-major into one buffer, ONE ssr_pair_metrics_multi launch sequence (8 real transforms per target instead of 14), every
    (degraded, target) pair against the oracle on the same degraded signal at the bars of the pair tests."""
    from ssr_eval_amd import backend as B
    n_t, n, K = 16, 192000, 7
    g = torch.Generator(device="cuda").manual_seed(20220328)
    tgt = (0.1 * torch.randn((n_t, n), generator=g, device="cuda", dtype=torch.float32)).contiguous()
    tr = B.Ragged.from_uniform(tgt)
    est = torch.empty((K, n_t, n), dtype=torch.float32, device="cuda")
    lplan = B.get_plan(2048, 441, "f64")
    for k, cut in enumerate(CUT_BINS):
        B.LowpassBatch(lplan, tr, [cut] * n_t, out=est[k].reshape(-1)).run()
    mb = B.MultiPairBatch(B.get_plan(2048, 512, "f64"), B.Ragged.from_uniform(est.view(K * n_t, n)), tr, K)
    got = mb.run(B.M_ALL).cpu().numpy()
    assert np.isfinite(got).all()
    est_h, tgt_h = est.cpu().numpy(), tgt.cpu().numpy()
    n_check = n_t if (os.cpu_count() or 1) >= 16 else 3
    pairs = [(est_h[k, t], tgt_h[t]) for t in range(n_check) for k in range(K)]
    _check_rows(got[:n_check].reshape(-1, 4), _oracle_many(pairs), "cfg3-multi")
    assert (np.diff(got[:, :, 0], axis=1) < 0).all()                       # LSD falls as the cutoff rises


def test_evaluate_arrays_scores_every_key_through_the_multi_entry(monkeypatch):
    """SSR_Eval_Helper.evaluate_arrays with four float32 degradation keys per file goes through ssr_pair_metrics_multi (one target
    transform per file) and returns what the per-pair path returns."""
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee, backend as B
    calls = {"multi": 0}
    orig = B.pair_metrics_multi

    def spy(*a, **k):
        calls["multi"] += 1
        return orig(*a, **k)
    monkeypatch.setattr(B, "pair_metrics_multi", spy)
    h = SSR_Eval_Helper(BasicTestee(), input_sr=48000, output_sr=48000, evaluation_sr=48000, test_data_root=None,
                        setting_fft={"cutoff_freq": [2000, 4000, 8000]}, setting_subsampling={"cutoff_freq": [6000]})
    rng = np.random.default_rng(8)
    items = [((0.1 * rng.standard_normal(n)).astype(np.float32),) * 2 for n in (20000, 31000, 26000)]
    res = h.evaluate_arrays(items)
    assert calls["multi"] == 1 and all(len(r) == 4 for r in res)
    for (tgt, x), r in zip(items, res):
        d = h.preprocess_array(x, 48000)
        assert list(d.keys()) == list(r.keys())
        for key, y in d.items():
            want = h.audio_metrics.evaluation(np.asarray(y, np.float32), tgt, "")
            np.testing.assert_allclose(_vec(r[key]), _vec(want), rtol=2e-6, atol=2e-6)
