"""Developer experiment: throughput of the degradation kernels (K6 FFT low-pass, K7 polyphase) on cfg-3 / cfg-5 shapes."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B, _lib

def t(fn, it=5):
    return bench.event_time_ms(fn, it)

def main():
    dev = torch.device("cuda", 0)
    res = {}
    # cfg-5: 16k -> 44.1k -> 48k, 64000-sample utterances
    n_utt = int(os.environ.get("UTT", "4096"))
    x = 0.1 * torch.randn((n_utt, 64000), device=dev)
    r = B.Ragged.from_uniform(x)
    y1 = B.resample_poly(r, 441, 160)
    r2 = B.Ragged.from_uniform(torch.stack(y1))

    def direct(rr, up, down):
        """C-ABI call with descriptors prepared once (no per-call Python / H2D overhead in the timed loop)."""
        rp = B.ResamplePlan.get(up, down, dev)
        out_len = np.array([rp.n_out(n) for n in rr.lens_host], dtype=np.int64)
        out_off = np.concatenate(([0], np.cumsum(out_len)[:-1])).astype(np.int64)
        out = torch.empty(int(out_len.sum()), dtype=torch.float32, device=dev)
        ood, old = torch.from_numpy(out_off).to(dev), torch.from_numpy(out_len.astype(np.int32)).to(dev)
        lib = _lib.load()
        def run():
            _lib.check(lib.ssr_resample_poly(B._vp(rr.data), B._vp(rr.off), B._vp(rr.len), B._vp(ood), B._vp(old), rr.n,
                                             int(out_len.max()), rp.up, rp.down, B._vp(rp.taps), int(rp.taps.numel()),
                                             rp.n_pre_remove, B._vp(out), B._stream()))
        return run
    f1, f2 = direct(r, 441, 160), direct(r2, 160, 147)
    ms1, ms2 = t(f1), t(f2)
    res["resample 16k->44.1k ms"] = ms1
    res["resample 44.1k->48k ms"] = ms2
    out_samples = n_utt * 192000
    res["cfg5 resampled Msamples/s (both stages)"] = out_samples / ((ms1 + ms2) * 1e-3) / 1e6
    res["cfg5 algorithmic GB/s"] = n_utt * 1024000 / ((ms1 + ms2) * 1e-3) / 1e9
    res["stage1 GB/s (r+w)"] = n_utt * (64000 + 176400) * 4 / (ms1 * 1e-3) / 1e9
    res["stage2 GB/s (r+w)"] = n_utt * (176400 + 192000) * 4 / (ms2 * 1e-3) / 1e9
    # cfg-3: FFT low-pass 2048/441 on 4 s @ 48k
    n_p = 1024
    z = 0.1 * torch.randn((n_p, 192000), device=dev)
    rz = B.Ragged.from_uniform(z)
    plan = B.get_plan(2048, 441, "f64", dev)
    cuts = [341] * n_p
    fl = lambda: B.fft_lowpass(plan, rz, cuts)
    ms = t(fl, 3)
    res["fft_lowpass 1024 x 4s ms"] = ms
    res["fft_lowpass utt/s"] = n_p / (ms * 1e-3)
    res["fft_lowpass algorithmic GB/s"] = n_p * 2 * 192000 * 4 / (ms * 1e-3) / 1e9
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main()


def iir():
    from scipy import signal
    dev = torch.device("cuda", 0)
    for n_utt in (256, 1024, 4096):
        z = 0.1 * torch.randn((n_utt, 176400), device=dev)
        rz = B.Ragged.from_uniform(z)
        for order in (4, 10):
            sos = signal.butter(order, 0.3, output="sos")
            f = lambda: B.sosfiltfilt(sos, rz)
            ms = bench.event_time_ms(f, 2)
            print("sosfiltfilt butter order %d, %d x 4 s @ 44.1k: %.2f ms -> %.0f utt/s" % (order, n_utt, ms, n_utt / (ms * 1e-3)))

if __name__ == "__main__" and os.environ.get("IIR"):
    iir()
