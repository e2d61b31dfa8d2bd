"""Developer tool (CPU, build container: imports /root/reference): REAL members of the reference's SISpec / log-SISpec round-off class
(ssr_eval/metrics.py:114-121, ssr_eval/utils.py:68-92).

AudioMetrics.sispec runs three float32 reductions over the whole spectrogram (torch.sum of a product, torch.norm twice).  Their result
depends on how torch splits the reduction - on the thread count of the process and on the memory layout of the tensors (the reference
builds them from the TRANSPOSED view of librosa's [F, T] array, `.clone()` keeps those strides) - by more than the north_star bar of
1e-5 when the value is a difference of nearly equal energies (cfg-3's shallow cuts: SISpec of a low-passed noise against the noise).
This tool evaluates the REFERENCE'S OWN function (imported from /root/reference through tests/golden/make_golden.py's stubs) on cfg-3
shaped pairs - seeded targets, the published torchlibrosa low-pass in the multi-threaded conv1d order (oracle.lowpass: the member the
HIP conv engine reproduces bit for bit) - at 1 / 2 / 4 / 8 / 16 torch threads x {reference layout, contiguous} and writes

    tests/golden/sispec_members.json   per case: seed, cutoff, every member's value, min / max, the float64 evaluation ("exact")
    profiles/r06_sispec_members.json   the same + a summary table

The GPU test (tests/test_gpu_configs.py::test_cfg3_sispec_inside_the_references_member_spread) regenerates the same pairs, runs the HIP
low-pass + metrics and asserts that the HIP value lies INSIDE [min, max] of these real members (not merely inside |ref32 - exact|).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as G  # noqa: E402  (stubs for the four absent packages + the reference import)
from ssr_eval.utils import to_log  # noqa: E402  (the reference's)
from oracle import lowpass as olp, metrics as om  # noqa: E402

CUTOFFS = [1000, 2000, 4000, 6000, 8000, 12000, 16000]
N, SR, N_FFT, HOP = 192000, 48000, 2048, 512
THREADS = (1, 2, 4, 8, 16)


def target(i):
    return (0.1 * np.random.default_rng(20220328 + i).standard_normal(N)).astype(np.float32)


def members(es, ts, am):
    """{label: (sispec, log_sispec)} through the reference's AudioMetrics.sispec."""
    out = {}
    old = torch.get_num_threads()
    for th in THREADS:
        torch.set_num_threads(th)
        for layout in ("reference", "contiguous"):
            e, t = (es, ts) if layout == "reference" else (es.contiguous(), ts.contiguous())
            si = float(am.sispec(e.clone(), t.clone()))
            lsi = float(am.sispec(to_log(e.clone()), to_log(t.clone())))
            out["%s_t%d" % (layout, th)] = (si, lsi)
    torch.set_num_threads(old)
    return out


def main():
    n_t = int(os.environ.get("TARGETS", "4"))
    am = G.AudioMetrics(48000)
    torch.set_num_threads(8)
    cases = []
    for i in range(n_t):
        tgt = target(i)
        ts = om.wav_to_spectrogram(tgt, N_FFT, HOP)
        for c, hc in enumerate(CUTOFFS):
            torch.set_num_threads(8)
            est = olp.lowpass(tgt, hc, SR, 1, "stft_hard")               # the multi-threaded conv1d member (= the HIP engine, bit for bit)
            es = om.wav_to_spectrogram(est, N_FFT, HOP)
            m = members(es, ts, am)
            ex = (float(om.sispec_exact(es, ts)), float(om.sispec_exact(om.to_log(es.clone()), om.to_log(ts.clone()))))
            case = {"target_seed": 20220328 + i, "target_index": i, "cutoff_hz": hc, "est_crc": int(np.abs(est).sum(dtype=np.float64) * 1e6) % (1 << 31),
                    "members": {k: list(v) for k, v in m.items()}, "exact": list(ex)}
            for j, name in enumerate(("sispec", "log_sispec")):
                vals = [v[j] for v in m.values()]
                case[name] = {"min": min(vals), "max": max(vals), "reference_t8": m["reference_t8"][j], "exact": ex[j],
                              "spread_db": max(vals) - min(vals), "spread_rel": (max(vals) - min(vals)) / max(abs(ex[j]), 1e-30),
                              "exact_inside": bool(min(vals) <= ex[j] <= max(vals))}
            cases.append(case)
            print("target %d cut %5d Hz | sispec %+.6f dB members [%+.6f, %+.6f] (spread %.2e dB = %.1e rel) exact %+.6f %s | log_sispec %+.6f [%+.6f, %+.6f] (%.2e dB) exact %+.6f %s"
                  % (i, hc, case["sispec"]["reference_t8"], case["sispec"]["min"], case["sispec"]["max"], case["sispec"]["spread_db"],
                     case["sispec"]["spread_rel"], ex[0], "in" if case["sispec"]["exact_inside"] else "OUT",
                     case["log_sispec"]["reference_t8"], case["log_sispec"]["min"], case["log_sispec"]["max"], case["log_sispec"]["spread_db"],
                     ex[1], "in" if case["log_sispec"]["exact_inside"] else "OUT"), flush=True)
    doc = {"source": "tools/exp_sispec_members.py: ssr_eval.metrics.AudioMetrics.sispec imported from /root/reference, torch %s, %d targets x %d cutoffs of "
                     "4 s @ 48 kHz (0.1 N(0,1), numpy default_rng(20220328 + i)), estimate = the published torchlibrosa low-pass (conv1d, 8 threads), "
                     "spectrograms 2048 / 512; members: torch threads %s x layouts (reference = the transposed view the reference builds, contiguous)"
                     % (torch.__version__, n_t, len(CUTOFFS), list(THREADS)),
           "cases": cases}
    gold = {"source": doc["source"], "cases": [{k: c[k] for k in ("target_seed", "target_index", "cutoff_hz", "est_crc", "sispec", "log_sispec")} for c in cases]}
    json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "sispec_members.json"), "w"), indent=1)
    json.dump(doc, open(os.path.join(ROOT, "profiles", "r06_sispec_members.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
