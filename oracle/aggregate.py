"""Oracle (test infrastructure): mean-of-speaker-means aggregation.

ssr_eval/eval.py:200-216 with dict_mean (ssr_eval/utils.py:24-28): per speaker, np.mean over files
of every metric of every degradation key; then np.mean over speakers.
"""
import numpy as np


def dict_mean(dict_list):
    return {k: np.mean([d[k] for d in dict_list]) for k in dict_list[0].keys()}


def aggregate(final_result):
    """final_result: {speaker: {file: {key: {metric: float}}}} -> (each_speaker, averaged)."""
    each = {}
    keys = None
    for spk, files in final_result.items():
        keys = list(next(iter(files.values())).keys())
        each[spk] = {k: dict_mean([v[k] for v in files.values()]) for k in keys}
    averaged = {k: dict_mean([each[s][k] for s in final_result]) for k in keys}
    return each, averaged
