#!/usr/bin/env python
"""bench.py - throughput of the ssr_eval metric hot path on MI355X.

A "step" is one pass of the hot path (ssr_pair_metrics: STFT n_fft=2048 / hop=512 of both signals,
fused LSD epilogue, SSIM) over one batch of synthetic (estimate, target) pairs that is ALREADY RESIDENT
IN HBM when the timed region starts.  Workload = BASELINE.json configs[1]: 1024 pairs of 4 s @ 48 kHz
float32 per GPU (weak scaling: every rank owns its own 1024 pairs; the only collective is one float64
all-reduce of the per-rank metric sums per step - the final mean-LSD/SSIM - over RCCL).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the task contract): metric/value/unit..., plus
  "roofline":     dominant-kernel algorithmic HBM GB/s (2*n*4+32 bytes per pair) vs 8 TB/s,
                  the kernel's duration measured here with HIP events on the launch stream;
  "cpu_baseline": the CPU oracle (NumPy pocketfft f64 STFT + torch-CPU LSD + scipy.ndimage SSIM, i.e. the
                  reference's arithmetic restated) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR, SECONDS, N_FFT, HOP = 48000, 4, 2048, 512
N_SAMPLES = SR * SECONDS                     # 192,000
BYTES_PER_PAIR = 2 * N_SAMPLES * 4 + 32      # read est + target once, write 4 doubles
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6


def make_inputs(n_pairs, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    tgt = 0.1 * torch.randn((n_pairs, N_SAMPLES), generator=g, device=device, dtype=torch.float32)
    est = tgt + 0.01 * torch.randn((n_pairs, N_SAMPLES), generator=g, device=device, dtype=torch.float32)
    return est.contiguous(), tgt.contiguous()


def pmc_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_traffic.json, produced by tools/collect_profiles.sh + tools/pmc_to_json.py: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return d["traffic_bytes_per_launch"][kernel_key]["total_bytes"], os.path.basename(files[-1])
    except Exception:
        return None, None


def event_time_ms(fn, iters):
    """Average duration of fn() in ms, HIP events on the current stream (the one the kernels launch on)."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters


def _cpu_pair(args):
    """LSD + SSIM of one pair through the oracle (the reference's arithmetic restated on the CPU)."""
    est, tgt = args
    torch.set_num_threads(1)
    from oracle import metrics as om
    es, ts = om.wav_to_spectrogram(est, N_FFT, HOP), om.wav_to_spectrogram(tgt, N_FFT, HOP)
    return float(om.lsd(es, ts)), float(om.ssim(es, ts))


def cpu_baseline(est, tgt, budget_s=20.0):
    """pairs/s of the oracle on the host: (i) 1 process / 1 thread, (ii) one process per core."""
    import multiprocessing as mp
    n_avail = est.shape[0]
    pairs = [(est[i].cpu().numpy(), tgt[i].cpu().numpy()) for i in range(min(n_avail, 256))]
    torch.set_num_threads(1)
    _cpu_pair(pairs[0])                                   # warm-up
    t0 = time.perf_counter()
    vals = [_cpu_pair(pairs[i]) for i in range(4)]
    one = (time.perf_counter() - t0) / 4
    cores = max(1, min(os.cpu_count() or 1, 64))
    n_pool = int(max(cores, min(len(pairs), cores * max(1, int(budget_s * 0.6 / max(one, 1e-3))))))
    n_pool = min(n_pool, len(pairs))
    pool_rate = None
    try:
        ctx = mp.get_context("fork")
        with ctx.Pool(cores) as pool:
            pool.map(_cpu_pair, pairs[:cores])            # warm the workers
            t0 = time.perf_counter()
            pool.map(_cpu_pair, pairs[:n_pool], chunksize=max(1, n_pool // (cores * 4)))
            pool_rate = n_pool / (time.perf_counter() - t0)
    except Exception as e:                                # a locked-down box: report the 1-thread number only
        sys.stderr.write("cpu_baseline pool failed: %r\n" % (e,))
    return vals, 1.0 / one, pool_rate, cores, n_pool


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=1024, help="pairs per GPU per step (BASELINE config: 1024)")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    from ssr_eval_amd import backend as B
    from ssr_eval_amd import dist as D
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    if world > 1:
        D.init_from_env("nccl")
    assert world == a.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    dev = torch.device("cuda", local_rank)

    est, tgt = make_inputs(a.pairs, dev, 20220328 + rank)
    plan = B.get_plan(N_FFT, HOP, a.precision, dev)
    batch = B.PairBatch(plan, B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
    mask = B.M_LSD | B.M_SSIM
    cnt = torch.full((1,), float(a.pairs), dtype=torch.float64, device=dev)
    agg = torch.zeros(3, dtype=torch.float64, device=dev)

    def step():
        out = batch.run(mask)
        # per-rank sums (LSD, SSIM, count) -> the job-wide mean needs exactly one tiny all-reduce
        torch.cat([out[:, 0].sum(0, keepdim=True), out[:, 3].sum(0, keepdim=True), cnt], out=agg)
        if world > 1:
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        return out

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    total_pairs = a.pairs * world * a.steps
    value = total_pairs / elapsed
    mean_lsd, mean_ssim = float(agg[0] / agg[2]), float(agg[1] / agg[2])

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- dominant-kernel timing (HIP events on the launch stream), outside the timed region ----------
    it = max(3, min(10, a.steps))
    ms_stft = event_time_ms(lambda: batch.run(mask, stages=1), it)
    ms_ssim = event_time_ms(lambda: batch.run(mask, stages=2), it)
    ms_fin = event_time_ms(lambda: batch.run(mask, stages=4), it)
    ms_all4 = event_time_ms(lambda: batch.run(B.M_ALL), it)
    dom_name, dom_ms = ("ssr_stft_pair(k_stft)", ms_stft) if ms_stft >= ms_ssim else ("ssr_ssim(k_ssim)", ms_ssim)
    alg_bytes = BYTES_PER_PAIR * a.pairs
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic("k_stft<double, 11" if ms_stft >= ms_ssim else "k_ssim")
    if a.pairs != 1024 or a.precision != "f64":
        traffic, traffic_src = None, None          # the PMC file was collected at the default workload only
    fft_flops = 2 * 376 * 2.5 * 2048 * 11 * a.pairs          # 42.4 MFLOP of real-FFT work per pair
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(dom_ms, 4),
                "note": "fused path is compute-side (f64 FFT + f64 SSIM moments); secondary: STFT kernel runs at "
                        "%.2f TFLOP/s of real-FFT work = %.3f of the f64 vector peak"
                        % (fft_flops / (ms_stft * 1e-3) / 1e12, fft_flops / (ms_stft * 1e-3) / 1e12 / FP64_PEAK_TFLOPS)}

    # ---- API-true variant AudioMetrics(48000): n_fft 2229 (Bluestein, M = 8192) / hop 480, small batch ---
    extra = {"stage_ms": {"stft+lsd": round(ms_stft, 4), "ssim": round(ms_ssim, 4), "finalize": round(ms_fin, 4)},
             "full_metric_set_pairs_per_s_per_gpu": round(a.pairs / (ms_all4 * 1e-3), 1),
             "mean_lsd": mean_lsd, "mean_ssim": mean_ssim}
    try:
        if world > 1:
            raise RuntimeError("skipped at N > 1 (other ranks are waiting at the barrier)")
        nb = min(a.pairs, 128)
        plan2 = B.get_plan(2229, 480, a.precision, dev)
        b2 = B.PairBatch(plan2, B.Ragged.from_uniform(est[:nb].contiguous()), B.Ragged.from_uniform(tgt[:nb].contiguous()))
        ms2 = event_time_ms(lambda: b2.run(mask), 3)
        extra["api_true_2229_480_pairs_per_s_per_gpu"] = round(nb / (ms2 * 1e-3), 1)
    except Exception as e:  # pragma: no cover
        extra["api_true_error"] = repr(e)

    cpu = None
    if not a.no_cpu_baseline and world == 1:      # contract: the CPU baseline is timed on rank 0 at N = 1 only
        vals, rate1, rate_pool, cores, n_pool = cpu_baseline(est, tgt)
        got = out[:len(vals)].cpu().numpy()
        rel = max(max(abs(got[i, 0] - v[0]) / abs(v[0]), abs(got[i, 3] - v[1]) / abs(v[1])) for i, v in enumerate(vals))
        extra["parity_vs_oracle_max_rel_err"] = float(rel)
        cpu = {"value": round(rate_pool if rate_pool else rate1, 3), "unit": "pairs/s", "cores": cores if rate_pool else 1,
               "kind": "port",
               "sample": "%d pairs of the same workload (4 s @ 48 kHz, STFT 2048/512 + LSD + SSIM) through the NumPy/SciPy "
                         "oracle, one process per core (1 thread each)" % (n_pool if rate_pool else 4),
               "value_1thread": round(rate1, 3)}

    line = {"metric": "utterance-pairs/sec (LSD+SSIM, 48kHz, n_fft=2048)", "value": round(value, 2), "unit": "pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "cfg-2: %d synthetic 48 kHz 4 s float32 (est, target) pairs per GPU resident in HBM, "
                                   "STFT n_fft=2048 hop=512 (T=376, F=1025), LSD + SSIM, transform precision %s"
                                   % (a.pairs, a.precision),
                       "pairs_per_gpu": a.pairs, "samples_per_utterance": N_SAMPLES, "n_fft": N_FFT, "hop": HOP,
                       "parallelism": "utterance-sharded x%d, one float64 all-reduce per step" % world},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
