#!/usr/bin/env python
"""Round-3 additions to the golden vectors -> tests/golden/reference_vectors_r3.npz (build container only).

Same rules as make_golden.py (which this imports for its stubs + the reference import): /root/reference/ssr_eval is imported
as-is and executed; the file written holds data only (inputs + the reference's outputs).

  ev3_*   AudioMetrics(rate).evaluation (ssr_eval/metrics.py:51-107) on pairs that reach the engines round 3 added or changed:
          speech32k      32 kHz (n_fft 1486 = 2 x 743, hop 320: two waves per frame pair), FFT-low-passed estimate
          speech48k_long 48 kHz (n_fft 2229 = 3 x 743, hop 480), 2.1 s: 211 frames - many rounds of the rotating four-wave kernel
          speech16k      16 kHz (n_fft 743, hop 160), band-limited estimate
  c5_*    BASELINE cfg-5 in small through the reference's own calls: an utterance at 16 kHz up-sampled with
          librosa.resample(..., res_type="polyphase") (the stub routes it to scipy.signal.resample_poly, which is what librosa
          runs for that res_type) 16 k -> 44.1 k -> 48 k as SSR_Eval_Helper.preprocess does (ssr_eval/eval.py:144-150), then
          AudioMetrics(48000, n_fft=2048, hop=512).evaluation against a 48 kHz target
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the stubs, imports the reference)

AudioMetrics, ref_lowpass = G.AudioMetrics, G.ref_lowpass


def main():
    out = {}
    cases = []
    sp = G.speechlike(41, 24000, 32000)
    ref_lowpass.f_helper = None
    lp = ref_lowpass.lowpass(sp, 4000, 32000, order=1, _type="stft_hard").astype(np.float32)
    cases.append(("speech32k", 32000, lp, sp))
    sp = G.speechlike(42, 100800, 48000)
    rng = np.random.default_rng(42)
    cases.append(("speech48k_long", 48000, (sp + 0.003 * rng.standard_normal(len(sp))).astype(np.float32), sp))
    sp = G.speechlike(43, 20000, 16000)
    ref_lowpass.f_helper = None
    lp = ref_lowpass.lowpass(sp, 2000, 16000, order=1, _type="stft_hard").astype(np.float32)
    cases.append(("speech16k", 16000, lp, sp))
    names = []
    for name, rate, e, t in cases:
        am = AudioMetrics(rate)
        res = am.evaluation(e, t, "")
        names.append(name)
        out["ev3_%s_est" % name], out["ev3_%s_tgt" % name] = e, t
        out["ev3_%s_rate" % name] = np.array(rate)
        out["ev3_%s_nfft_hop" % name] = np.array([am.n_fft, am.hop_length])
        out["ev3_%s_out" % name] = np.array([res["lsd"], res["log_sispec"], res["sispec"], res["ssim"]], np.float64)
    out["ev3_names"] = np.array(names)

    # ---- cfg-5 in small: the reference's resampling calls, then its metric at the bench's STFT sizes
    import librosa                                   # the stub make_golden installed (polyphase -> scipy.signal.resample_poly)
    x16 = G.speechlike(44, 12000, 16000)
    y44 = librosa.resample(x16, orig_sr=16000, target_sr=44100, res_type="polyphase")
    y48 = librosa.resample(y44, orig_sr=44100, target_sr=48000, res_type="polyphase")
    tgt = G.speechlike(45, len(y48), 48000)
    am = AudioMetrics(48000)
    am.n_fft, am.hop_length = 2048, 512
    res = am.evaluation(np.asarray(y48, np.float32), tgt, "")
    out["c5_x16"], out["c5_y44"], out["c5_y48"], out["c5_tgt"] = x16, np.asarray(y44, np.float32), np.asarray(y48, np.float32), tgt
    out["c5_out"] = np.array([res["lsd"], res["log_sispec"], res["sispec"], res["ssim"]], np.float64)
    np.savez_compressed(os.path.join(HERE, "reference_vectors_r3.npz"), **out)
    print("wrote reference_vectors_r3.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.endswith("_out")})
    for k in out:
        if k.endswith("_out"):
            print(k, out[k])


if __name__ == "__main__":
    main()
