"""Developer tool: long randomized HIP-vs-oracle sweep (pair metrics, STFT magnitude / complex, silent segments, ragged
batches, every engine).  Prints the worst relative errors; exits non-zero on a miss.  Not part of the test suite."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
from ssr_eval_amd import backend as B
from oracle import metrics as om, stft as ostft

def main():
    n_cases = int(os.environ.get("CASES", "60"))
    rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
    sizes = [(2048, 512), (2229, 480), (2048, 441), (1024, 256), (743, 160), (1114, 240), (1486, 320), (512, 100), (4096, 1024), (256, 64), (771, 100)]
    # PREC=f32: the float32 transform (not the reference's arithmetic): same sweep, bars 100x wider - a crash / garbage check
    PREC = os.environ.get("PREC", "f64")
    bar_rel, bar_mag = (3e-5, 3e-7) if PREC == "f64" else (3e-3, 3e-5)
    worst = np.zeros(4); worst_mag = 0.0; bad = 0
    for case in range(n_cases):
        n_fft, hop = sizes[int(rng.integers(0, len(sizes)))]
        plan = B.get_plan(n_fft, hop, PREC)
        n_items = int(rng.integers(1, 7))
        ests, tgts = [], []
        for _ in range(n_items):
            n = int(rng.integers(7 * hop + n_fft // 2 + 1, 7 * hop + 5 * n_fft + 3000))
            t = (0.1 * rng.standard_normal(n)).astype(np.float32)
            kind = int(rng.integers(0, 5))
            if kind == 0: e = (t + 0.02 * rng.standard_normal(n)).astype(np.float32)
            elif kind == 1: e = (0.3 * t + 0.05 * rng.standard_normal(n)).astype(np.float32)
            elif kind == 2: e = np.convolve(t, np.ones(9, np.float32) / 9, mode="same").astype(np.float32)
            elif kind == 3:                       # silent stretch in the estimate
                e = (t + 0.02 * rng.standard_normal(n)).astype(np.float32); a = int(rng.integers(0, n // 2)); e[a:a + int(rng.integers(n_fft, 3 * n_fft))] = 0
            else:                                 # silent stretch in the target
                e = (t + 0.02 * rng.standard_normal(n)).astype(np.float32); t = t.copy(); a = int(rng.integers(0, n // 2)); t[a:a + int(rng.integers(n_fft, 3 * n_fft))] = 0
            ests.append(e); tgts.append(t)
        got = B.pair_metrics(plan, ests, tgts)
        mags = B.stft(plan, ests)
        re, im = B.stft(plan, tgts, kind="complex")
        for i, (e, t) in enumerate(zip(ests, tgts)):
            w, ex = om.evaluation_with_exact(e, t, n_fft=n_fft, hop=hop)
            want = np.array([w["lsd"], w["log_sispec"], w["sispec"], w["ssim"]])
            rel = np.abs(got[i] - want) / np.maximum(np.abs(want), np.array([1e-3, 1.0, 1.0, 1e-3]))   # dB values near 0: absolute
            # SISpec pair: where the reference's float32 sums are themselves off the float64 evaluation of its formula by more
            # than the bar, the kernel is held to that evaluation and to the band the reference's round-off spans
            # (tests/test_gpu_parity.py::assert_sispec_parity)
            for j, key in ((1, "log_sispec"), (2, "sispec")):
                scale = max(abs(ex[key]), 1.0)
                band = abs(want[j] - ex[key])
                if band > 3e-6 * scale and abs(got[i][j] - ex[key]) <= 1e-6 * scale + 1e-6 and abs(got[i][j] - want[j]) <= band + 2e-6 * scale:
                    rel[j] = abs(got[i][j] - ex[key]) / scale
            worst = np.maximum(worst, rel)
            ref = ostft.stft_mag_TF(e, n_fft, hop)
            dm = np.abs(mags[i].cpu().numpy() - ref).max() / ref.max()
            spec = ostft.librosa_stft(t, n_fft, hop).T
            dc = max(np.abs(re[i].cpu().numpy() - spec.real).max(), np.abs(im[i].cpu().numpy() - spec.imag).max()) / np.abs(spec).max()
            zero_ok = ((mags[i].cpu().numpy() == 0) == (ref == 0)).all()
            worst_mag = max(worst_mag, dm, dc)
            if (rel > bar_rel).any() or dm > bar_mag or dc > bar_mag or not zero_ok:
                bad += 1
                print("MISS case %d n_fft=%d hop=%d n=%d rel=%s dm=%.2e dc=%.2e zero_ok=%s got=%s want=%s" % (case, n_fft, hop, len(e), rel, dm, dc, zero_ok, got[i], want))
    print("cases", n_cases, "worst rel (lsd, log_sispec, sispec, ssim)", worst, "worst mag", worst_mag, "misses", bad)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
