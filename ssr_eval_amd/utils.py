"""Host bookkeeping helpers with the semantics of ssr_eval/utils.py (I/O half; the numeric half of that
file - to_log, pow_p_norm, pow_norm, energy_unify - lives inside the HIP kernels, see csrc/ssr_stft.h)."""
import json
import wave

import numpy as np

EPS = 1e-12


def dict_mean(dict_list):
    """Per-key float64 mean over a list of dicts (ssr_eval/utils.py:24-28)."""
    first = dict_list[0]
    return {k: np.mean([d[k] for d in dict_list]) for k in first.keys()}


def write_json(obj, fname):
    """indent=4 JSON (ssr_eval/utils.py:18-21)."""
    with open(fname, "w") as f:
        f.write(json.dumps(obj, indent=4))


def load_json(fname):
    with open(fname, "r") as f:
        return json.load(f)


def write_list(items, fname):
    with open(fname, "w") as f:
        f.write("".join(str(w) + "\n" for w in items))


def read_list(fname):
    with open(fname, "r") as f:
        return [line.rstrip("\n") for line in f]


def get_sample_rate(fname):
    with wave.open(fname) as f:
        return f.getframerate()


def get_framesLength(fname):
    with wave.open(fname) as f:
        return f.getnframes()
