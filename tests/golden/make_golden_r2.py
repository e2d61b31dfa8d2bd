#!/usr/bin/env python
"""Round-2 additions to the golden vectors -> tests/golden/reference_vectors_r2.npz (build container only).

Same rules as make_golden.py (which this imports for its stubs + the reference import): /root/reference/ssr_eval is
imported as-is and executed; the file written holds data only (inputs + the reference's outputs).

  mc_*    multi-channel [B, C, T, F] spectrogram tensors through AudioMetrics.lsd / .sispec / .ssim and
          utils.energy_unify (metrics.py:109-132, utils.py:68-92: C > 1 mixes per-channel products with all-channel norms)
  c3_*    BASELINE cfg-3 in small: SSR_Eval_Helper.lowpass_stft_hard on a 48 kHz signal for the cutoff sweep
          {1, 2, 4, 6, 8, 12, 16} kHz (keys, cut bins, degraded signals) and AudioMetrics.evaluation of every
          (degraded, target) pair at (2048, 512) and at the API-true AudioMetrics(48000) sizes
  ssq_*   the subsampling degradation at its `low_rate == sr -> -1` quirk for a 16 kHz input (resample_poly 7349/7350)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the stubs, imports the reference)

AudioMetrics, ref_lowpass, ref_utils = G.AudioMetrics, G.ref_lowpass, G.ref_utils
SSR_Eval_Helper, BasicTestee = G.SSR_Eval_Helper, G.BasicTestee


def main():
    out = {}
    # ---- multi-channel tensors
    rng = np.random.default_rng(52)
    tsp = (np.abs(rng.standard_normal((2, 3, 19, 40))) * 2).astype(np.float32)
    esp = np.abs(tsp * (1 + 0.3 * rng.standard_normal(tsp.shape))).astype(np.float32)
    am = AudioMetrics(44100)
    E, T = torch.tensor(esp), torch.tensor(tsp)
    out["mc_est"], out["mc_tgt"] = esp, tsp
    out["mc_lsd"] = am.lsd(E.clone(), T.clone()).numpy()
    out["mc_sispec"] = np.array(float(am.sispec(E.clone(), T.clone())))
    out["mc_log_sispec"] = np.array(float(am.sispec(ref_utils.to_log(E.clone()), ref_utils.to_log(T.clone()))))
    out["mc_ssim"] = am.ssim(E.clone(), T.clone()).numpy()
    _, eu = ref_utils.energy_unify(E.clone(), T.clone())
    out["mc_energy_unify_tgt"] = eu.numpy()

    # ---- cfg-3 in small
    x = G.speechlike(31, 14400, 48000) + 0.01 * np.random.default_rng(31).standard_normal(14400).astype(np.float32)
    x = x.astype(np.float32)
    out["c3_x"] = x
    cutoffs = [1000, 2000, 4000, 6000, 8000, 12000, 16000]
    h = SSR_Eval_Helper(BasicTestee(), input_sr=48000, output_sr=48000, evaluation_sr=48000, test_data_root="/tmp/_r2_none",
                        setting_fft={"cutoff_freq": list(cutoffs)})
    ref_lowpass.f_helper = None
    ref_lowpass.lowpass(x[:1200], 100, 48000, _type="stft_hard")          # instantiate f_helper
    cuts = []
    orig = ref_lowpass.f_helper.spectrogram_phase_to_wav

    def spy(sps, coss, sins, length):
        cuts.append(int((sps[0, 0].abs().sum(dim=0) != 0).numpy().sum()))
        return orig(sps, coss, sins, length)
    ref_lowpass.f_helper.spectrogram_phase_to_wav = spy
    d = h.lowpass_stft_hard("", x, 48000)
    ref_lowpass.f_helper = None
    keys = list(d.keys())
    out["c3_keys"] = np.array(keys)
    out["c3_cut_bins"] = np.array(cuts)
    am48 = AudioMetrics(48000)
    amb = AudioMetrics(48000)
    amb.n_fft, amb.hop_length = 2048, 512
    res_api, res_bench = [], []
    for k in keys:
        y = np.asarray(d[k], np.float32)
        out["c3_y_" + k] = y
        for am_, res in ((am48, res_api), (amb, res_bench)):
            r = am_.evaluation(y, x, "")
            res.append([r["lsd"], r["log_sispec"], r["sispec"], r["ssim"]])
    out["c3_metrics_api2229"] = np.array(res_api, np.float64)
    out["c3_metrics_2048_512"] = np.array(res_bench, np.float64)

    # ---- subsampling quirk at 16 kHz: cutoff list [8000] is doubled to 16000 == sr -> 15999 -> highcut 7999
    x16 = G.speechlike(33, 4000, 16000)
    out["ssq_x"] = x16
    h2 = SSR_Eval_Helper(BasicTestee(), input_sr=16000, output_sr=16000, evaluation_sr=16000, test_data_root="/tmp/_r2_none",
                         setting_subsampling={"cutoff_freq": [8000]})
    d2 = h2.lowpass_subsampling("", x16, 16000)
    (k2, y2), = d2.items()
    out["ssq_key"] = np.array(k2)
    out["ssq_y"] = np.asarray(y2)
    np.savez_compressed(os.path.join(HERE, "reference_vectors_r2.npz"), **out)
    print("wrote", len(out), "arrays;", {k: (v.shape, str(v.dtype)) for k, v in out.items() if k.startswith(("c3_m", "c3_c", "ssq_k", "c3_k"))})
    print(out["c3_cut_bins"], out["c3_metrics_2048_512"][:, 0])


if __name__ == "__main__":
    main()
