#!/usr/bin/env python
"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

/root/reference/ssr_eval is imported as-is.  Four third-party packages it needs are absent from the
image (librosa, soundfile, scikit-image, torchlibrosa); name-only stub modules are injected into
sys.modules and bound to the restatements in oracle/ (see oracle/__init__.py: parity at those three
boundaries is unpinned; everything else executed below is the reference's own code).  SciPy is real.
Round 4: the torchlibrosa stub is bound to the package's arithmetic AS PUBLISHED (oracle.stft.tl_stft_conv / tl_istft_conv:
float32 F.conv1d with the float32 DFT x Hann weights, F.fold) - it used to be a float64-FFT idealisation, which made every
lp_* / c3_* / *fftlp* vector circular at that boundary and 2-7 % off in LSD.

Outputs are data only (inputs + the reference's outputs).  Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import scipy
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import stft as ostft, ssim as ossim, resample as oresample  # noqa: E402


# ---------------------------------------------------------------- stubs for the four absent packages
def _install_stubs():
    librosa = types.ModuleType("librosa")
    librosa.stft = lambda y, n_fft=2048, hop_length=None, **kw: ostft.librosa_stft(y, n_fft, hop_length)
    librosa.istft = lambda S, hop_length=None, length=None, **kw: ostft.librosa_istft(S, hop_length, length)
    librosa.resample = lambda y, orig_sr, target_sr, res_type="kaiser_best", **kw: \
        oresample.librosa_resample_polyphase(y, orig_sr, target_sr)

    def _no_io(*a, **k):
        raise RuntimeError("file I/O is out of scope for golden generation")
    librosa.load = _no_io
    sys.modules["librosa"] = librosa

    sf = types.ModuleType("soundfile")
    sf.write = _no_io
    sys.modules["soundfile"] = sf

    skimage = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.metrics")
    skm.structural_similarity = lambda a, b, win_size=7, **kw: ossim.structural_similarity(a, b, win_size)
    skimage.metrics = skm
    sys.modules["skimage"] = skimage
    sys.modules["skimage.metrics"] = skm

    tl = types.ModuleType("torchlibrosa")
    tls = types.ModuleType("torchlibrosa.stft")

    class STFT(torch.nn.Module):
        def __init__(self, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True,
                     pad_mode="reflect", freeze_parameters=True):
            super().__init__()
            assert window == "hann" and center and pad_mode == "reflect" and win_length in (None, n_fft)
            self.n_fft, self.hop = n_fft, hop_length

        def forward(self, x):
            re, im = ostft.tl_stft_conv(x.detach().numpy(), self.n_fft, self.hop)
            return torch.tensor(re), torch.tensor(im)

    class ISTFT(torch.nn.Module):
        def __init__(self, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True,
                     pad_mode="reflect", freeze_parameters=True):
            super().__init__()
            self.n_fft, self.hop = n_fft, hop_length

        def forward(self, real, imag, length):
            return torch.tensor(ostft.tl_istft_conv(real.detach().numpy(), imag.detach().numpy(), length,
                                                    self.n_fft, self.hop))

    def magphase(real, imag):
        mag = (real ** 2 + imag ** 2) ** 0.5
        return mag, real / torch.clamp(mag, 1e-10, np.inf), imag / torch.clamp(mag, 1e-10, np.inf)

    tls.STFT, tls.ISTFT, tls.magphase = STFT, ISTFT, magphase
    tl.stft = tls
    sys.modules["torchlibrosa"] = tl
    sys.modules["torchlibrosa.stft"] = tls


_install_stubs()
sys.path.insert(0, REF)
import ssr_eval  # noqa: E402  (the reference)
from ssr_eval.metrics import AudioMetrics  # noqa: E402
from ssr_eval import lowpass as ref_lowpass  # noqa: E402
from ssr_eval import utils as ref_utils  # noqa: E402
from ssr_eval.eval import SSR_Eval_Helper, BasicTestee  # noqa: E402

assert ssr_eval.__file__.startswith(REF)


def speechlike(seed, n, sr):
    """Deterministic band-rich test signal: harmonics with vibrato + decaying noise floor."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    f0 = 110.0 + 40.0 * rng.random()
    x = np.zeros(n)
    for h in range(1, 40):
        x += (1.0 / h) * np.sin(2 * np.pi * (h * f0 * t + 0.3 * np.sin(2 * np.pi * 3.0 * t)) + rng.random() * 6.28)
    x = 0.08 * x / np.max(np.abs(x)) + 0.002 * rng.standard_normal(n)
    return x.astype(np.float32)


def noise_pair(seed, n):
    """SURVEY 8(d) cfg-2 synthetic pair."""
    rng = np.random.default_rng(seed)
    tgt = (0.1 * rng.standard_normal(n)).astype(np.float32)
    est = (tgt + 0.01 * rng.standard_normal(n).astype(np.float32)).astype(np.float32)
    return est, tgt


def main():
    out = {}
    manifest = {"numpy": np.__version__, "scipy": scipy.__version__, "torch": torch.__version__,
                "reference": "haoheliu/ssr_eval v0.0.6 imported from /root/reference with stubs",
                "torchlibrosa_stub": "as published: float32 F.conv1d / F.fold on torch-CPU (oracle.stft.tl_*_conv), %d threads"
                                     % torch.get_num_threads()}

    # ---- A1: integer tables (metrics.py:16-19)
    rates = [16000, 24000, 32000, 44100, 48000]
    out["a1_rates"] = np.array(rates)
    out["a1_nfft_hop"] = np.array([[AudioMetrics(r).n_fft, AudioMetrics(r).hop_length] for r in rates])

    # ---- A3..A7: AudioMetrics.evaluation on waveform pairs
    cases = []
    # (name, rate, est, tgt)
    e, t = noise_pair(20220328, 12000)
    cases.append(("noise48k", 48000, e, t))
    e, t = noise_pair(20220329, 9000)
    cases.append(("noise44k", 44100, e, t))
    e, t = noise_pair(20220330, 6000)
    cases.append(("noise16k", 16000, e, t))
    sp = speechlike(7, 16000, 48000)
    ref_lowpass.f_helper = None
    lp = ref_lowpass.lowpass(sp, 6000, 48000, order=1, _type="stft_hard")
    cases.append(("speech48k_fftlp6k", 48000, lp.astype(np.float32), sp))
    sp2 = speechlike(8, 14000, 44100)
    lp2 = ref_lowpass.lowpass(sp2, 4000, 44100, order=1, _type="stft_hard")
    cases.append(("speech44k_fftlp4k_ragged", 44100, lp2.astype(np.float32)[:-37], sp2))
    sp3 = speechlike(9, 10000, 24000)
    cases.append(("speech24k_scaled", 24000, (0.5 * sp3).astype(np.float32), sp3))
    names = []
    for name, rate, e, t in cases:
        am = AudioMetrics(rate)
        res = am.evaluation(e, t, "")
        names.append(name)
        out["ev_%s_est" % name] = e
        out["ev_%s_tgt" % name] = t
        out["ev_%s_rate" % name] = np.array(rate)
        out["ev_%s_out" % name] = np.array([res["lsd"], res["log_sispec"], res["sispec"], res["ssim"]], np.float64)
        # the spectrogram the reference fed to its reductions (through the librosa stub)
        if name in ("noise44k", "speech48k_fftlp6k"):
            out["ev_%s_tgt_sp" % name] = am.wav_to_spectrogram(t[:min(len(e), len(t))]).numpy()[0, 0]
    manifest["evaluation_cases"] = names

    # bench STFT parameters (librosa defaults 2048/512) through the reference's reductions
    am = AudioMetrics(48000)
    am.n_fft, am.hop_length = 2048, 512
    e, t = noise_pair(20220331, 10000)
    res = am.evaluation(e, t, "")
    out["ev_bench2048_est"], out["ev_bench2048_tgt"] = e, t
    out["ev_bench2048_out"] = np.array([res["lsd"], res["log_sispec"], res["sispec"], res["ssim"]], np.float64)

    # ---- A4..A6 directly on spectrogram tensors (reference's own torch arithmetic, no stubs involved)
    rng = np.random.default_rng(5)
    tsp = np.abs(rng.standard_normal((2, 1, 23, 65))).astype(np.float32) * 3
    esp = (tsp * (1 + 0.2 * rng.standard_normal(tsp.shape))).astype(np.float32)
    esp[0, 0, 3, 10:20] = 0.0                      # exact zeros exercise the EPS paths
    esp = np.abs(esp)
    am = AudioMetrics(44100)
    out["sp_est"], out["sp_tgt"] = esp, tsp
    out["sp_lsd"] = am.lsd(torch.tensor(esp), torch.tensor(tsp)).numpy()
    out["sp_sispec"] = np.array(float(am.sispec(torch.tensor(esp), torch.tensor(tsp))))
    out["sp_sispec_each"] = np.array([float(am.sispec(torch.tensor(esp[i:i + 1]), torch.tensor(tsp[i:i + 1])))
                                      for i in range(2)])
    out["sp_log_sispec_each"] = np.array([float(am.sispec(ref_utils.to_log(torch.tensor(esp[i:i + 1])),
                                                            ref_utils.to_log(torch.tensor(tsp[i:i + 1]))))
                                          for i in range(2)])
    out["sp_to_log"] = ref_utils.to_log(torch.tensor(esp)).numpy()
    out["sp_ssim"] = am.ssim(torch.tensor(esp), torch.tensor(tsp)).numpy()
    # A6 helpers called directly (ssr_eval/utils.py:43-50,68-92)
    out["sp_from_log"] = ref_utils.from_log(torch.tensor(esp) * 2 - 3).numpy()         # covers the clip at 5
    out["sp_pow_p_norm"] = ref_utils.pow_p_norm(torch.tensor(tsp)).numpy()
    out["sp_pow_norm"] = ref_utils.pow_norm(torch.tensor(esp), torch.tensor(tsp)).numpy()
    eu_e, eu_t = ref_utils.energy_unify(torch.tensor(esp), torch.tensor(tsp))
    out["sp_energy_unify_est"], out["sp_energy_unify_tgt"] = eu_e.numpy(), eu_t.numpy()

    # ---- A8/A9: FFT low-pass (lowpass.py:17-28,156-196) + cut bins
    x = speechlike(11, 9000, 44100)
    out["lp_x"] = x
    cuts = []
    for hc, fs in [(1000, 44100), (2000, 44100), (4000, 44100), (6000, 44100), (8000, 44100), (12000, 44100),
                   (16000, 44100), (22049, 44100), (6000, 48000), (12000, 48000), (4000, 16000)]:
        captured = {}
        ref_lowpass.f_helper = None
        ref_lowpass.lowpass(x[:1200], 100, fs, _type="stft_hard")       # instantiate f_helper
        orig = ref_lowpass.f_helper.spectrogram_phase_to_wav

        def spy(sps, coss, sins, length, _orig=orig, _c=captured):
            nz = (sps[0, 0].abs().sum(dim=0) != 0).numpy()
            _c["cut"] = int(nz.sum())
            return _orig(sps, coss, sins, length)
        ref_lowpass.f_helper.spectrogram_phase_to_wav = spy
        y = ref_lowpass.lowpass(x, hc, fs, order=1, _type="stft_hard")
        cuts.append([hc, fs, captured["cut"]])
        if (hc, fs) in [(4000, 44100), (12000, 44100), (6000, 48000)]:
            out["lp_y_%d_%d" % (hc, fs)] = np.asarray(y, np.float32)
    out["lp_cut_table"] = np.array(cuts)
    ref_lowpass.f_helper = None
    fh = ssr_eval.dsp.FDomainHelper()
    mag, cos, sin = fh.wav_to_spectrogram_phase(torch.tensor(x[None, None, :4000]))
    out["fd_x"] = x[:4000]
    out["fd_mag"], out["fd_cos"], out["fd_sin"] = mag.numpy()[0, 0], cos.numpy()[0, 0], sin.numpy()[0, 0]
    out["fd_roundtrip"] = fh.spectrogram_phase_to_wav(mag, cos, sin, 4000).numpy()[0, 0]

    # ---- A10/A11: polyphase resampling (SciPy is the real third-party code) + subsampling/align_length
    xr = speechlike(12, 6400, 16000)
    out["rs_x16k"] = xr
    out["rs_16k_to_44k"] = sys.modules["librosa"].resample(xr, 16000, 44100, res_type="polyphase")
    out["rs_44k_to_48k"] = sys.modules["librosa"].resample(out["rs_16k_to_44k"], 44100, 48000, res_type="polyphase")
    xs = speechlike(13, 8000, 44100)
    out["ss_x"] = xs
    for hc in (2000, 4000, 12000):
        out["ss_y_%d" % hc] = np.asarray(ref_lowpass.lowpass(xs, hc, 44100, order=1, _type="subsampling"), np.float32)
    out["al_pad"] = ref_lowpass.align_length(np.arange(7.0), np.arange(4.0))
    out["al_cut"] = ref_lowpass.align_length(np.arange(4.0), np.arange(7.0))
    for ft in ("butter", "cheby1", "ellip", "bessel"):
        out["iir_%s" % ft] = np.asarray(ref_lowpass.lowpass(xs, 4000, 44100, order=6, _type=ft), np.float64)

    # ---- A13: BasicTestee integer helpers + postprocessing
    bt = BasicTestee()
    energy = np.cumsum(np.abs(np.random.default_rng(3).standard_normal(200)) * np.linspace(1, 0, 200) ** 4)
    out["bt_energy"] = energy
    out["bt_find_cutoff"] = np.array([bt._find_cutoff(energy, th) for th in (0.5, 0.9, 0.95, 0.97, 0.999)])
    xg = speechlike(14, 9000, 44100)
    ref_lowpass.f_helper = None
    xl = np.asarray(ref_lowpass.lowpass(xg, 4000, 44100, order=1, _type="stft_hard"), np.float32)
    out["bt_x"], out["bt_out"] = xl, xg
    out["bt_cutoff_index"] = np.array(bt._get_cutoff_index(xl))
    out["bt_post"] = np.asarray(bt.postprocessing(xl, xg.copy()), np.float32)

    # ---- A14: key naming / cutoff doubling (eval.py:121-126,401-421)
    h = object.__new__(SSR_Eval_Helper)
    user = {"cutoff_freq": [1000, 4000, 22050]}
    h.setting_fft = h._cutoff2sr(user)
    assert user["cutoff_freq"] == [2000, 8000, 44100]          # mutated in place (eval.py:125)
    ref_lowpass.f_helper = None
    d = h.lowpass_stft_hard("f.wav", x, 44100)
    manifest["fft_keys"] = list(d.keys())
    for k, v in d.items():
        out["key_" + k] = np.asarray(v, np.float32)
    h.setting_subsampling = h._cutoff2sr({"cutoff_freq": [4000]})
    d = h.lowpass_subsampling("f.wav", xs, 44100)
    manifest["subsampling_keys"] = list(d.keys())
    for k, v in d.items():
        out["key_" + k] = np.asarray(v, np.float32)
    out["helper_shift_p3"] = h.shift(np.arange(8.0), 3)
    out["helper_shift_m3"] = h.shift(np.arange(8.0), -3)
    manifest["cache_file_name"] = h.cache_file_name("proc_x", "/a/b/c.wav")

    # ---- A12: aggregation through the reference's evaluate() with a seeded evaluate_single
    rng = np.random.default_rng(99)
    counts = {"p360": 5, "p361": 3, "s5": 4}
    keys = ["proc_fft_24000_44100", "proc_fft_8000_44100"]
    per_file = {}
    with tempfile.TemporaryDirectory() as td:
        root = os.path.join(td, "vctk")
        for spk, c in counts.items():
            os.makedirs(os.path.join(root, spk))
            for i in range(c):
                fn = "%s_%03d_mic1.flac" % (spk, i)
                open(os.path.join(root, spk, fn), "w").close()
                per_file[os.path.join(root, spk, fn)] = {
                    k: {m: float(rng.random() * 5) for m in ("lsd", "log_sispec", "sispec", "ssim")} for k in keys}
        open(os.path.join(root, "p360", "p360_000_mic1_proc_x.flac"), "w").close()   # skipped: "proc"
        open(os.path.join(root, "README.txt"), "w").close()
        h2 = object.__new__(SSR_Eval_Helper)
        h2.test_data_root, h2.test_name = root, "golden"
        h2.evaluate_single = lambda f: per_file[f]
        cwd = os.getcwd()
        os.chdir(td)
        try:
            final = h2.evaluate()
        finally:
            os.chdir(cwd)
        agg = {"per_file": {os.path.relpath(k, root): v for k, v in per_file.items()},
               "each_speaker": final["each_speaker"], "averaged": final["averaged"],
               "speakers": [s for s in final if s not in ("each_speaker", "averaged")],
               "files": {s: list(final[s].keys()) for s in counts}}
    with open(os.path.join(HERE, "aggregate.json"), "w") as f:
        json.dump(agg, f, indent=1)

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    sz = os.path.getsize(os.path.join(HERE, "reference_vectors.npz"))
    print("wrote %d arrays, %.1f KB" % (len(out), sz / 1024))


if __name__ == "__main__":
    main()
