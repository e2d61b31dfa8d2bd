"""Developer experiment: ssr_resample_sinc (N2 ingest resampler, kaiser_best) on batches shaped like evaluate()'s: FILES files of
1.5-9 s at each rate pair of PAIRS ("44100:48000,...").  Prints ms per launch and output samples per second."""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B


def main():
    n_files = int(os.environ.get("FILES", "32"))
    rng = np.random.default_rng(3)
    pairs = [tuple(int(v) for v in w.split(":")) for w in os.environ.get("PAIRS", "44100:48000,48000:44100,16000:44100,48000:16000,44100:16000").split(",")]
    for sr_orig, sr_new in pairs:
        xs = [(0.1 * rng.standard_normal(int(rng.uniform(1.5, 9.0) * sr_orig))).astype(np.float32) for _ in range(n_files)]
        r = B.Ragged.from_list(xs)
        ys = B.resample_sinc(r, sr_orig, sr_new)
        ms = bench.event_time_ms(lambda: B.resample_sinc(r, sr_orig, sr_new), 5)
        n_out = sum(int(y.shape[0]) for y in ys)
        print(json.dumps({"rates": [sr_orig, sr_new], "files": n_files, "ms_per_launch_incl_host": round(ms, 3),
                          "output_Msamples_per_s": round(n_out / ms / 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
