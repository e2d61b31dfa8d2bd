"""Developer tool: bench.py's `evaluate_end_to_end` side figure on its own (367 PCM .wav files, identity testee, 12 kHz FFT low-pass,
evaluation_sr 48000), PASSES timed passes after two warm-ups - the command `rocprofv3 --kernel-trace --stats` is pointed at to see which
kernels an evaluate() pass spends its GPU time in.  Prints the median files/s and the per-pass seconds."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
from ssr_eval_amd import SSR_Eval_Helper, BasicTestee  # noqa: E402
from ssr_eval_amd.io import write_wav  # noqa: E402


def host_profile(h, bf):
    """Host seconds per stage of one pass (no synchronisation added: what the host spends queueing, and where it waits)."""
    import collections
    import torch
    from ssr_eval_amd import io as IO, backend as Bk, eval as EV, metrics as MT
    acc, saved = collections.OrderedDict(), []

    def timed(mod, name, label, wrap_result=False):
        f = getattr(mod, name)

        def g(*a_, **k_):
            t = time.perf_counter(); r = f(*a_, **k_); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
            if wrap_result and callable(r):
                def r2(*a2, **k2):
                    t2 = time.perf_counter(); v = r(*a2, **k2); acc[label + " -> wait/collect"] = acc.get(label + " -> wait/collect", 0.0) + time.perf_counter() - t2
                    return v
                return r2
            return r
        raw = mod.__dict__.get(name, f)
        saved.append((mod, name, raw)); setattr(mod, name, staticmethod(g) if isinstance(raw, staticmethod) else g)
    timed(IO, "decode_packed_async", "decode_packed_async (start + wait)", True)
    timed(Bk, "upload_decoded", "upload_decoded")
    timed(IO, "to_rate_resident", "to_rate_resident")
    timed(EV.SSR_Eval_Helper, "preprocess_arrays", "preprocess_arrays")
    timed(EV.SSR_Eval_Helper, "_infer_and_collect", "_infer_and_collect")
    timed(Bk, "resample_poly", "resample_poly")
    timed(MT.AudioMetrics, "evaluation_batch", "evaluation_batch (queue)", True)
    timed(MT.AudioMetrics, "_prepare_pair", "  of which _prepare_pair")
    timed(Bk.Ragged, "from_list", "  Ragged.from_list (all stages)")
    timed(EV.SSR_Eval_Helper, "evaluate_files", "evaluate_files (queue)", True)
    timed(EV.SSR_Eval_Helper, "_assemble", "_assemble")
    try:
        for _ in range(2):
            acc.clear()
            torch.cuda.synchronize(); t0 = time.perf_counter(); h.evaluate(save_json=False, batch_files=bf); acc["whole pass"] = time.perf_counter() - t0
    finally:
        for mod, name, f in saved:
            setattr(mod, name, f)
    for k, v in acc.items():
        print("  host %-44s %7.2f ms" % (k, v * 1e3), flush=True)


def main():
    rng = np.random.default_rng(4)
    root = tempfile.mkdtemp(prefix="ssr_e2e_")
    try:
        n_files = 0
        counts = [424, 424, 123, 419, 301, 424, 424, 398] if os.environ.get("FULL_SET") else [53, 53, 15, 52, 38, 53, 53, 50]   # FULL_SET: VCTK test-set shape (2,937 files)
        for s, c in enumerate(counts):
            os.makedirs(os.path.join(root, "p%03d" % (360 + s)))
            for i in range(c):
                n = int(rng.integers(int(1.5 * 44100), 9 * 44100))
                write_wav(os.path.join(root, "p%03d" % (360 + s), "u%03d.wav" % i), 0.1 * rng.standard_normal(n), 44100)
                n_files += 1
        kw = {"setting_fft": {"cutoff_freq": [12000]}}
        if os.environ.get("IIR"):              # the zero-phase IIR degradations as well: IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8"
            f, c, o = os.environ["IIR"].split(":")
            kw["setting_lowpass_filtering"] = {"filter": f.split(","), "cutoff_freq": [int(v) for v in c.split(",")],
                                               "filter_order": [int(v) for v in o.split(",")]}
        eval_sr = int(os.environ.get("EVAL_SR", 48000))      # 48000: the reference's README / test.py; 44100: its signature default (2048 / 441)
        h = SSR_Eval_Helper(BasicTestee(), input_sr=44100, output_sr=44100, evaluation_sr=eval_sr, test_data_root=root, **kw)
        bf = int(os.environ["BATCH_FILES"]) if os.environ.get("BATCH_FILES") else None      # None: the helper's default_batch_files()
        h.evaluate(limit_test_nums=2, limit_test_speaker=1, save_json=False)
        h.evaluate(save_json=False, batch_files=bf)
        times = []
        for _ in range(int(os.environ.get("PASSES", 7))):
            t0 = time.perf_counter()
            res = h.evaluate(save_json=False, batch_files=bf)
            times.append(time.perf_counter() - t0)
        if os.environ.get("CPROFILE"):
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable(); h.evaluate(save_json=False, batch_files=bf); pr.disable()
            pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ["CPROFILE"]))
        if os.environ.get("HOSTPROF"):
            host_profile(h, bf)
        print("evaluate(): %d files, batch_files %s, median %.1f files/s, passes %s, averaged lsd %.6f" % (
            n_files, bf, n_files / float(np.median(times)), ["%.4f" % t for t in times], res["averaged"]["proc_fft_24000_44100"]["lsd"]), flush=True)
        if os.environ.get("IIR"):
            print("keys: %d; e.g. %s" % (len(res["averaged"]), {k: round(v["lsd"], 6) for k, v in list(res["averaged"].items())[:3]}), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
