"""Developer experiment: SSR_Eval_Helper.evaluate() on bench.py's 367-file tree for several batch_files values (median of 5)."""
import os, sys, json, time, tempfile, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssr_eval_amd import SSR_Eval_Helper, BasicTestee
from ssr_eval_amd.io import write_wav

rng = np.random.default_rng(4)
root = tempfile.mkdtemp(prefix="ssr_e2e_")
try:
    for s, c in enumerate([53, 53, 15, 52, 38, 53, 53, 50]):
        os.makedirs(os.path.join(root, "p%03d" % (360 + s)))
        for i in range(c):
            n = int(rng.integers(int(1.5 * 44100), 9 * 44100))
            write_wav(os.path.join(root, "p%03d" % (360 + s), "u%03d.wav" % i), 0.1 * rng.standard_normal(n), 44100)
    h = SSR_Eval_Helper(BasicTestee(), input_sr=44100, output_sr=44100, evaluation_sr=48000, test_data_root=root,
                        setting_fft={"cutoff_freq": [12000]})
    h.evaluate(save_json=False)
    res = {}
    for bf in (32, 64, 128, 192, 256, 400):
        h.evaluate(save_json=False, batch_files=bf)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); r = h.evaluate(save_json=False, batch_files=bf); ts.append(time.perf_counter() - t0)
        res[bf] = round(float(np.median(ts)) * 1e3, 2)
    print(json.dumps({"ms_per_pass_by_batch_files": res, "averaged_lsd": r["averaged"][list(r["averaged"])[0]]["lsd"]}))
finally:
    shutil.rmtree(root, ignore_errors=True)
