"""Drop-in for ssr_eval/utils.py.  The I/O helpers are host bookkeeping; the tensor helpers (to_log, from_log,
pow_p_norm, pow_norm, energy_unify - SURVEY 8(a) A6) run in libssrhip.so.  Inside AudioMetrics.sispec they are
fused into the metric kernels (csrc/ssr_stft.h); the stand-alone versions below exist for callers that import
them directly.  Tensors may live on any device: they are moved to the GPU, the result returns to the input's."""
import json
import wave

import numpy as np
import torch

EPS = 1e-12


def to_log(input):
    """log10(input + 1e-12) (utils.py:43-44)."""
    from . import backend as B
    return B.elementwise("to_log", input).to(input.device)


def from_log(input):
    """10 ** clip(input, max=5) (utils.py:47-49)."""
    from . import backend as B
    return B.elementwise("from_log", input).to(input.device)


def pow_p_norm(signal):
    """Squared L2 norm over every dimension but the first, keepdim (utils.py:68-76): float32 norm, then squared."""
    from . import backend as B
    s = B.energy_sums(signal, signal, signal.shape[0])[:, 0]
    nrm = torch.sqrt(s).to(torch.float32)
    return (nrm * nrm).reshape([signal.shape[0]] + [1] * (signal.dim() - 1)).to(signal.device)


def pow_norm(s1, s2):
    """sum(s1 * s2) over dimensions 2.., keepdim (utils.py:85-92)."""
    from . import backend as B
    if s1.dim() < 2:
        raise ValueError("pow_norm needs [B, C, ...] tensors")
    s = B.energy_sums(s1, s2, s1.shape[0] * s1.shape[1])[:, 2].to(torch.float32)
    return s.reshape(list(s1.shape[:2]) + [1] * (s1.dim() - 2)).to(s1.device)


def energy_unify(estimated, original):
    """(estimated, original * <estimated, original> / (|original|^2 + EPS)) (utils.py:79-82).  With C > 1 the inner
    product is per (batch, channel) while the norm runs over every dimension but the batch, exactly as the reference
    broadcasts them."""
    from . import backend as B
    if original.dim() < 2:
        raise ValueError("energy_unify needs [B, C, ...] tensors")
    C = original.shape[1]
    mul = pow_norm(estimated, original).reshape(-1)                                  # [B * C]
    div = pow_p_norm(original).reshape(-1) + torch.tensor(EPS, dtype=torch.float32)  # [B]
    return estimated, B.scale_items(original, mul, div.repeat_interleave(C)).to(original.device)


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
