"""Developer tool: randomized sweep of the degradation / helper entry points against SciPy and the oracle
(resampler and sosfiltfilt bit-exact, STFT-domain low-pass bit-exact against oracle/tl_chain.c, cross-correlation shift exact, float64-estimate metrics)."""
import os, sys
import numpy as np, torch
from scipy import signal
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
from ssr_eval_amd import backend as B
from ssr_eval_amd.lowpass import lowpass, lowpass_batch, cut_bin, align_length
from oracle import lowpass as olp, metrics as om, tl_chain

def main():
    rng = np.random.default_rng(int(os.environ.get("SEED", "2")))
    bad = 0
    for case in range(int(os.environ.get("CASES", "30"))):
        k = int(rng.integers(2, 6))
        sigs = [(0.2 * rng.standard_normal(int(rng.integers(300, 30000)))).astype(np.float32) for _ in range(k)]
        if rng.random() < 0.3:
            sigs[0][len(sigs[0]) // 4: len(sigs[0]) // 2] = 0
        # resampler
        up, down = [(441, 160), (160, 147), (160, 441), (80, 147), (147, 80), (3, 2), (2, 3), (7, 5), (1, 4), (320, 441)][int(rng.integers(0, 10))]
        f64 = rng.random() < 0.3
        xs = [s.astype(np.float64) * 1.0000001 for s in sigs] if f64 else sigs
        for x, y in zip(xs, B.resample_poly(xs, up, down)):
            if not np.array_equal(y.cpu().numpy(), signal.resample_poly(x, up, down)):
                bad += 1; print("MISS resample", up, down, len(x), f64)
        # IIR
        ft = ["butter", "cheby1", "ellip", "bessel"][int(rng.integers(0, 4))]
        order = int(rng.integers(2, 11)); hc = int(rng.integers(500, 20000))
        sos = olp.iir_sos(hc, 44100, order, ft)
        edge = 3 * (2 * sos.shape[0] + 1 - min(int((sos[:, 2] == 0).sum()), int((sos[:, 5] == 0).sum())))
        ok_sigs = [x for x in xs if len(x) > edge]
        for x, y in zip(ok_sigs, B.sosfiltfilt(sos, ok_sigs)):
            if not np.array_equal(y.cpu().numpy(), signal.sosfiltfilt(sos, x)):
                bad += 1; print("MISS sosfiltfilt", ft, order, hc, len(x), f64)
        # FFT low-pass
        long_sigs = [s for s in sigs if len(s) > 1100]
        hcs = int(rng.integers(500, 22000))
        # (the default engine since round 4 is torchlibrosa's own float32 arithmetic: its restatement is oracle/tl_chain.c, bit for bit;
        # the float64 FFT restatement olp.lowpass(..., "stft_hard") this line used to compare with is another arithmetic, 4e-7 away)
        cut = cut_bin(hcs / int(44100 / 2))
        for x, y in zip(long_sigs, lowpass_batch(long_sigs, hcs, 44100, order=1, _type="stft_hard")):
            ref = align_length(x, tl_chain.stft_hard_lowpass(x, cut, 2048, 441))
            if not np.array_equal(y, ref):
                bad += 1; print("MISS fft_lowpass", hcs, len(x), np.abs(y - ref).max())
        # cross-correlation shift
        a_list, b_list, want = [], [], []
        for s in sigs:
            d = int(rng.integers(-min(len(s) // 3, 2000), min(len(s) // 3, 2000)))
            dec = np.roll(s, d).copy()
            if d >= 0: dec[:d] = 0
            else: dec[d:] = 0
            dec = (dec + 0.01 * rng.standard_normal(len(s))).astype(np.float32)
            a_list.append(dec); b_list.append(s); want.append(int(np.argmax(signal.correlate(dec, s))))
        got = list(B.xcorr_argmax(a_list, b_list))
        if got != want:
            bad += 1; print("MISS xcorr", got, want)
        # float64 estimate metrics (IIR output) on the longer signals
        plan = B.get_plan(2229, 480, "f64")
        mets = [s for s in sigs if len(s) > 7 * 480 + 1200 and len(s) > edge]
        if mets:
            ests = [signal.sosfiltfilt(sos, s) for s in mets]
            for e, t, g in zip(ests, mets, B.pair_metrics(plan, ests, mets)):
                w = om.evaluation(e, t, n_fft=2229, hop=480)
                want_v = np.array([w["lsd"], w["log_sispec"], w["sispec"], w["ssim"]])
                if (np.abs(g - want_v) > 3e-5 * np.maximum(np.abs(want_v), 1.0)).any():   # dB values near 0: absolute 3e-5
                    bad += 1; print("MISS est64 metrics", len(t), g, want_v)
    print("misses", bad)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
