#!/usr/bin/env python
"""bench.py - throughput of the ssr_eval DSP / metric hot path on MI355X.

    python bench.py [--config cfg2|cfg3|cfg4|cfg5] [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no torchrun environment spawns the N ranks itself (one process per GPU, RCCL) and fails if
fewer than N devices exist.  Rank 0 prints ONE JSON line (the task contract) with `roofline` and `cpu_baseline`.

A "step" is one pass of the hot path over one batch of synthetic input that is ALREADY RESIDENT IN HBM when the timed
region starts.  Workloads (BASELINE.json `configs`; weak scaling: every rank owns its own batch, the only collective
is one float64 all-reduce of the per-rank metric sums per step - the final mean - over RCCL):

  cfg2 (default, the headline metric): 1024 (est, target) pairs of 4 s @ 48 kHz per GPU -> ssr_pair_metrics with
        STFT 2048/512, LSD + SSIM.  Unit: pairs/s.
  cfg3: 1024 targets per GPU x the cutoff sweep {2,4,8,12,16,24,32 kHz} (cut bins 42..683 at fs 48 kHz):
        est = ssr_fft_lowpass(target, cut) (FDomainHelper 2048/441), then the full metric set.  Unit: pairs/s,
        a pair = (utterance, cutoff).
  cfg4: the full VCTK-shaped test set - 2,937 ragged utterances (1.5-9 s @ 48 kHz) of 8 speakers with the reference's file
        counts - STRONG scaling: the fixed set is sharded round-robin over the ranks, every step ends with the path's two
        real exchanges (ssr_eval/eval.py:200-216): one float64 SUM all-reduce of the [speakers, 4 metrics + count] buffer and
        one padded all-gather of the per-utterance rows.  Unit: pairs/s over the whole job.
  cfg5: 12,500 utterances of 64,000 samples @ 16 kHz per GPU (100 k on 8 GPUs): ssr_resample_poly 441/160 then
        160/147, then LSD (2048/512) against a 48 kHz target.  Unit: resampled samples/s (192,000 per utterance).

With the default config at N = 1 the cfg3 / cfg5 / API-true / end-to-end figures are measured too (short runs, outside
the timed region) and reported under "extra".
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
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6
CUTOFFS_HZ = [1000, 2000, 4000, 6000, 8000, 12000, 16000]    # sweep labels 2..32 kHz = 2 x cutoff
CUT_BINS = [int(1025 * (c / int(SR / 2))) for c in CUTOFFS_HZ]  # 42 85 170 256 341 512 683


def make_inputs(n_pairs, device, seed):
    """The cfg-2 synthetic pairs: target 0.1 N(0,1), estimate = target + 0.01 N(0,1); [n_pairs, 192000] float32."""
    g = torch.Generator(device=device).manual_seed(seed)
    tgt = (0.1 * torch.randn((n_pairs, N_SAMPLES), generator=g, device=device, dtype=torch.float32)).contiguous()
    est = (tgt + 0.01 * torch.randn((n_pairs, N_SAMPLES), generator=g, device=device, dtype=torch.float32)).contiguous()
    return est, tgt


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


def pmc_traffic(kernel_key):
    """HBM bytes per launch of a kernel (or of several: keys joined by "+", summed) from the committed rocprofv3 PMC passes
    of this same command (profiles/rNN_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, corrected per
    MI355X_MICROARCH.md)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))["traffic_bytes_per_launch"]
            total = 0.0
            for key in kernel_key.split("+"):
                hit = [v["total_bytes"] for k, v in d.items() if key in k]
                if not hit:
                    raise KeyError(key)
                total += hit[0]
            return total, os.path.basename(f)
        except Exception:
            continue
    return None, None


_HBM_MEASURED = {}


def measured_hbm_peak():
    """What THIS box sustains on a plain streaming read and on a copy (tools/ubench/hbm_probe.hip: 16-byte accesses, four
    independent loads in flight per lane, 2 GiB buffers - past the 256 MB Infinity Cache), timed with HIP events in this process:
    the measured ceiling SURVEY 8(d) asks to quote next to the 8 TB/s data-sheet peak (MI355X_MICROARCH.md: 6.29 TB/s on a
    float4 copy).  {"read_GBs", "copy_GBs"} (copy counts bytes read + bytes written) or {} when the probe library is missing."""
    if _HBM_MEASURED or not torch.cuda.is_available():
        return _HBM_MEASURED
    path = os.path.join(ROOT, "tools", "_build", "libhbmprobe.so")
    try:
        import ctypes as C
        lib = C.CDLL(path)
        lib.hbm_probe_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        nbytes = 2 << 30
        a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
        b = torch.empty_like(a)
        stream = torch.cuda.current_stream().cuda_stream
        best = {}
        for mode, name, moved in ((0, "read_GBs", nbytes), (1, "copy_GBs", 2 * nbytes)):
            for blocks in (2048, 4096, 8192):
                ms = event_time_ms(lambda: lib.hbm_probe_launch(a.data_ptr(), b.data_ptr(), nbytes, mode, blocks, stream), 5)
                best[name] = max(best.get(name, 0.0), round(moved / (ms * 1e-3) / 1e9, 1))
        del a, b
        torch.cuda.empty_cache()
        _HBM_MEASURED.update(best)
    except Exception as e:                               # the probe is a convenience: the line survives without it
        _HBM_MEASURED["error"] = repr(e)
    return _HBM_MEASURED


def fp64_mix_ceiling():
    """tools/_build/fp64_mix (tools/ubench/fp64_mix.hip, built by __graft_entry__.build()): units/s of a kernel with k_stft_wave's FP64
    instruction mix and LDS exchange pattern and no global memory, and the FP64 matrix-instruction rate - measured on this box."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "_build", "fp64_mix")
    if not os.path.exists(exe):
        return {}
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=90).stdout
        r = {}
        m = re.search(r"LDS exchanges/unit, 2 wave\(s\)/SIMD: ([0-9.]+) M units/s.*?clock ([0-9]+) MHz", out)
        if m:
            r["units_M_per_s_2_waves"], r["clock_MHz"] = float(m.group(1)), int(m.group(2))
        m = re.search(r"FP64 mix alone\s*, 2 wave\(s\)/SIMD: ([0-9.]+) M units/s", out)
        if m:
            r["units_M_per_s_2_waves_without_lds"] = float(m.group(1))
        m = re.search(r"DS variant 2 .*?: ([0-9.]+) M units/s", out)
        if m:                                           # the same unit with its exchange stores as single ds_write_b64 (profiles/r05_notes.md section 8)
            r["units_M_per_s_2_waves_split_writes"] = float(m.group(1))
        m = re.search(r"v_mfma_f64_16x16x4_f64, 2 wave\(s\)/SIMD: ([0-9.]+) shader cycles", out)
        if m:
            r["mfma_f64_16x16x4_cycles_per_instruction_and_simd"] = float(m.group(1))
        return r
    except Exception as e:
        return {"error": repr(e)}


def hbm_roofline(kernel, alg_bytes, ms, traffic_key=None, note=None):
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    traffic, src = pmc_traffic(traffic_key) if traffic_key else (None, None)
    meas = measured_hbm_peak()
    r = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 5),
         "peak_measured": meas.get("read_GBs"), "peak_measured_copy": meas.get("copy_GBs"),
         "frac_of_measured": round(achieved / meas["read_GBs"], 5) if meas.get("read_GBs") else None,
         "peak_measured_source": "streaming read / copy of 2 GiB with 16-byte accesses in this process (tools/ubench/hbm_probe.hip)",
         "traffic": traffic, "traffic_source": src,
         "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_ms": round(ms, 4),
         "kernel_ms_source": "hip_events on the launch stream, 3-10 iterations on this box (the rocprofv3 average of the same command: profiles/)"}
    if note:
        r["note"] = note
    return r


# ----------------------------------------------------------------------------------------------------------
# workloads
def _prec_name(a):
    return {"f64": "double", "f32": "float"}[a.precision]


class Cfg2:
    name = "cfg2"
    metric = "utterance-pairs/sec (LSD+SSIM, 48kHz, n_fft=2048)"
    unit = "pairs/s"
    n_fft, hop = N_FFT, HOP
    label = "cfg-2"

    @classmethod
    def frames(cls):
        return 1 + N_SAMPLES // cls.hop

    @classmethod
    def stft_kernel(cls, a):
        """The EXACT instantiated name of the transform kernel of this workload's step, as rocprofv3 prints it (LSD + magnitudes for
        SSIM, no running SISpec sums): <T, SUMS = false, SPLIT = true, MAG = true>."""
        return "k_stft_wave<%s, false, true, true>" % _prec_name(a)

    def __init__(self, a, dev, rank):
        from ssr_eval_amd import backend as B
        self.B, self.a, self.dev = B, a, dev
        n = a.pairs
        self.est, self.tgt = make_inputs(n, dev, 20220328 + rank)
        self.plan = B.get_plan(self.n_fft, self.hop, a.precision, dev)
        self.batch = B.PairBatch(self.plan, B.Ragged.from_uniform(self.est), B.Ragged.from_uniform(self.tgt))
        self.mask = B.M_LSD | B.M_SSIM
        self.units_per_step = n
        self.cnt = torch.full((1,), float(n), dtype=torch.float64, device=dev)
        self.agg = torch.zeros(3, dtype=torch.float64, device=dev)
        self.agg[2] = float(n)

    def step(self):
        out = self.batch.run(self.mask)
        # per-rank sums (LSD, SSIM, count) -> the job-wide mean needs exactly one tiny all-reduce.  One reduction kernel over the
        # strided view of columns 0 (LSD) and 3 (SSIM) straight into the buffer; the count sits in its last slot since __init__.
        torch.sum(out[:, 0::3], dim=0, out=self.agg[:2])
        return self.agg

    def config(self, world):
        a = self.a
        return {"workload": "%s: %d synthetic 48 kHz 4 s float32 (est, target) pairs per GPU resident in HBM, STFT n_fft=%d "
                            "hop=%d (T=%d, F=%d), LSD + SSIM, transform precision %s"
                            % (self.label, a.pairs, self.n_fft, self.hop, self.frames(), self.n_fft // 2 + 1, a.precision),
                "pairs_per_gpu": a.pairs, "samples_per_utterance": N_SAMPLES, "n_fft": self.n_fft, "hop": self.hop,
                "parallelism": "utterance-sharded x%d, one float64 all-reduce (24 B) per step%s"
                               % (world, ", issued asynchronously: it overlaps the next step's kernels" if world > 1 else "")}

    def report(self, a):
        B, batch, mask = self.B, self.batch, self.mask
        it = max(3, min(10, a.steps))
        ms_stft = event_time_ms(lambda: batch.run(mask, stages=1), it)
        ms_ssim = event_time_ms(lambda: batch.run(mask, stages=2), it)
        ms_fin = event_time_ms(lambda: batch.run(mask, stages=4), it)
        ms_all4 = event_time_ms(lambda: batch.run(B.M_ALL), it)
        alg = (2 * N_SAMPLES * 4 + 32) * a.pairs          # read est + target once, write 4 doubles
        T = self.frames()
        fft_tflops = 2 * T * 2.5 * self.n_fft * np.log2(self.n_fft) * a.pairs / (ms_stft * 1e-3) / 1e12   # 42.4 MFLOP of real-FFT work per pair at 2048/512
        kname = self.stft_kernel(a)
        dom = (kname, ms_stft, kname[:kname.index(",", kname.index(",") + 1)]) if ms_stft >= ms_ssim else ("k_ssim<4, true>", ms_ssim, "k_ssim")
        default_wl = a.pairs == 1024 and a.precision == "f64"
        ceil = fp64_mix_ceiling() if default_wl and self.n_fft == 2048 else {}
        units_s = T * a.pairs / (ms_stft * 1e-3)                   # frame pairs per second of the transform kernel
        roof = hbm_roofline(dom[0], alg, dom[1], dom[2] if default_wl else None,
                            "fused path is compute-side (f64 FFT + f64 SSIM moments); secondary: STFT kernel runs at %.2f TFLOP/s "
                            "of real-FFT work = %.3f of the f64 vector peak at 2.4 GHz; %.1f M frame pairs/s%s"
                            % (fft_tflops, fft_tflops / FP64_PEAK_TFLOPS, units_s / 1e6,
                               " = %.2f of the %.1f M/s a kernel with its hot loop's FP64 instruction mix and LDS exchanges, ideal ILP and no "
                               "global memory sustains on this box at two waves per SIMD (tools/ubench/fp64_mix, run in this process's shadow: "
                               "the ceiling of the mix, not of the kernel's whole instruction stream - profiles/r05_notes.md section 3)"
                               % (units_s / 1e6 / ceil["units_M_per_s_2_waves"], ceil["units_M_per_s_2_waves"]) if ceil.get("units_M_per_s_2_waves") else ""))
        if ceil:
            roof["fp64_mix_ceiling"] = ceil
        roof["entry_point"] = "ssr_pair_metrics (C ABI) -> ssr_stft_pair"
        # every kernel of the timed step under its rocprofv3 name (hip-event averages on the launch stream; the stats CSV of
        # `bench.py --config %s --no-side` under profiles/ holds the same three rows)
        extra = {"kernels_ms": {kname: round(ms_stft, 4), "k_ssim<4, true>": round(ms_ssim, 4), "k_finalize": round(ms_fin, 4)},
                 "stage_ms": {"stft+lsd": round(ms_stft, 4), "ssim": round(ms_ssim, 4), "finalize": round(ms_fin, 4)},
                 "full_metric_set_pairs_per_s_per_gpu": round(a.pairs / (ms_all4 * 1e-3), 1)}
        return roof, extra

    # CPU baseline: LSD + SSIM of one pair through the oracle
    _cpu_nfft, _cpu_hop = N_FFT, HOP            # (read by the static worker function: set per workload in cpu_inputs)

    def cpu_inputs(self, n):
        Cfg2._cpu_nfft, Cfg2._cpu_hop = self.n_fft, self.hop
        return [(self.est[i].cpu().numpy(), self.tgt[i].cpu().numpy()) for i in range(min(n, self.a.pairs))]

    @staticmethod
    def cpu_unit(item):
        from oracle import metrics as om
        est, tgt = item
        es, ts = om.wav_to_spectrogram(est, Cfg2._cpu_nfft, Cfg2._cpu_hop), om.wav_to_spectrogram(tgt, Cfg2._cpu_nfft, Cfg2._cpu_hop)
        return float(om.lsd(es, ts)), float(om.ssim(es, ts))

    cpu_desc = "pairs of the same workload (4 s @ 48 kHz, this workload's STFT size + LSD + SSIM) through the NumPy/SciPy/torch-CPU oracle"

    def parity(self, out_vals, n, first):
        """max relative error of (LSD, SSIM) of pairs first .. first + n - 1 against the oracle values the CPU baseline
        produced for exactly those pairs."""
        got = self.batch.run(self.mask)[first:first + n].cpu().numpy()
        return max(max(abs(got[i, 0] - v[0]) / abs(v[0]), abs(got[i, 3] - v[1]) / abs(v[1])) for i, v in enumerate(out_vals[:n]))


class ApiTrue(Cfg2):
    """The reference's OWN configuration of the headline metric: AudioMetrics(48000) = n_fft int(2048 / (44100 / 48000)) = 2229, hop 480
    (ssr_eval/metrics.py:16-19) - n_fft = 3 x 743: radix-3 over three Bluestein-743 sub-sequences on 1536-point transforms, four
    autonomous waves rotating through the sub-sequence transforms of consecutive frames (k_stft_r3_rot)."""
    name = "apitrue"
    metric = "utterance-pairs/sec (LSD+SSIM, 48kHz, AudioMetrics(48000): n_fft=2229 hop=480)"
    n_fft, hop = 2229, 480
    label = "API-true AudioMetrics(48000)"

    @classmethod
    def stft_kernel(cls, a):
        return "k_stft_r3_rot<%s, false, 3, 24, 0>" % _prec_name(a)


class Cfg3:
    name = "cfg3"
    metric = "utterance-pairs/sec (full metric set + FFT low-pass degradation, cutoff sweep {2k..32k}, 48kHz, n_fft=2048)"
    unit = "pairs/s"

    @staticmethod
    def engine_of(a):
        """--lowpass-engine, by default the PRODUCT's default (ssr_eval_amd.lowpass.DEFAULT_ENGINE = "conv": the reference's arithmetic)."""
        import importlib
        L = importlib.import_module("ssr_eval_amd.lowpass")      # (the package attribute `lowpass` is the function)
        return getattr(a, "lowpass_engine", None) or L.DEFAULT_ENGINE

    def __init__(self, a, dev, rank):
        from ssr_eval_amd import backend as B
        self.B, self.a, self.dev = B, a, dev
        self.engine = self.engine_of(a)
        g = torch.Generator(device=dev).manual_seed(20220328 + rank)
        n = a.pairs
        self.tgt = (0.1 * torch.randn((n, N_SAMPLES), generator=g, device=dev, dtype=torch.float32)).contiguous()
        tr = B.Ragged.from_uniform(self.tgt)
        self.lp_plan = B.get_plan(2048, 441, a.precision, dev, lowpass_engine=self.engine)   # FDomainHelper() of ssr_eval/lowpass.py:167
        self.plan = B.get_plan(N_FFT, HOP, a.precision, dev)
        # ONE estimate buffer, key-major [7][n][samples]: ONE ssr_fft_lowpass_multi call fills it (eval.py:401-410 loops the cutoffs
        # over the same waveform: on the conv engine the padded copy and the forward dense-DFT product are shared by the 7 keys),
        # then ONE ssr_pair_metrics_multi launch sequence scores the 7 keys (each target transformed and stored once -
        # ssr_eval/eval.py:136-154 scores every key against the same target)
        K = len(CUT_BINS)
        self.est = torch.empty((K, n, N_SAMPLES), dtype=torch.float32, device=dev)
        self.mlp = B.MultiLowpassBatch(self.lp_plan, tr, CUT_BINS, out=self.est.view(-1))
        self.batch = B.MultiPairBatch(self.plan, B.Ragged.from_uniform(self.est.view(K * n, N_SAMPLES)), tr, K)
        self.units_per_step = n * K
        self.cnt = torch.full((1,), float(self.units_per_step), dtype=torch.float64, device=dev)
        self.agg = torch.zeros(5, dtype=torch.float64, device=dev)

    def step(self):
        self.mlp.run()                                   # est[k] <- fft_lowpass(target, cut k), k = 0 .. 6
        acc = self.batch.run(self.B.M_ALL).sum((0, 1))
        torch.cat([acc, self.cnt], out=self.agg)
        return self.agg

    def config(self, world):
        a = self.a
        self.engine = self.engine_of(a)                  # (config() is also called on a bare shell of the class)
        return {"workload": "cfg-3: %d synthetic 48 kHz 4 s float32 targets per GPU resident in HBM x 7 cutoffs "
                            "(cut bins %s of FDomainHelper 2048/441 at fs 48 kHz): est[k] = stft_hard low-pass(target, cut k) through ONE "
                            "ssr_fft_lowpass_multi call (engine: %s%s), then LSD + log-SISpec + SISpec + SSIM of the 7 keys at STFT 2048/512 "
                            "through ONE ssr_pair_metrics_multi call, metric transform precision %s" % (
                                a.pairs, CUT_BINS, self.engine, " = the product default, torchlibrosa's float32 dense-DFT arithmetic on the "
                                "fp32 matrix cores" if self.engine == "conv" else "", a.precision),
                "lowpass_engine": self.engine,
                "targets_per_gpu": a.pairs, "cutoffs_hz": CUTOFFS_HZ, "cut_bins": CUT_BINS, "samples_per_utterance": N_SAMPLES,
                "parallelism": "utterance-sharded x%d, one float64 all-reduce (40 B) per step" % world}

    @staticmethod
    def conv_flops(n, cuts, shared_forward=True):
        """Useful flops of the dense-DFT low-pass of n utterances: per frame forward 2 n_fft (2 cut) + inverse 2 n_fft (2 K),
        K = cut + min(cut - 1, 1023) channels of the mirrored spectrum; the forward product once at the largest cut when shared."""
        rows = n * (1 + N_SAMPLES // 441)
        fwd = max(cuts) if shared_forward else sum(cuts)
        return rows * 2.0 * 2048 * 2 * (fwd + sum(c + min(c - 1, 1023) for c in cuts))

    def report(self, a):
        B, batch = self.B, self.batch
        it, K, n = 3, len(CUT_BINS), a.pairs
        engine = self.engine
        tr = B.Ragged.from_uniform(self.tgt)
        ms_lp_all = event_time_ms(lambda: self.mlp.run(), it)
        one = B.LowpassBatch(self.lp_plan, tr, [CUT_BINS[3]] * n, out=self.est[3].reshape(-1))
        one.ws = self.mlp.ws
        ms_lp = event_time_ms(lambda: one.run(), it)
        ms_multi = event_time_ms(lambda: batch.run(B.M_ALL), it)
        per_key = B.PairBatch(self.plan, B.Ragged.from_uniform(self.est[3]), tr)
        ms_pair = event_time_ms(lambda: per_key.run(B.M_ALL), it)       # what one key cost before ssr_pair_metrics_multi
        self.mlp.run()                                                  # (restore plane 3)
        if engine == "conv":
            # the dense-DFT engine is bound by the fp32 matrix cores (MFMA where the DFT is deliberately cast as a GEMM)
            flops = self.conv_flops(n, CUT_BINS)
            ach = flops / (ms_lp_all * 1e-3) / 1e12
            counts = {"k_tl_pad": 1, "k_tl_fwd": 1, "k_tl_inv": K, "k_tl_fold": K}
            traffic, src = None, None
            if a.pairs == 1024:
                parts = {k: pmc_traffic(k) for k in counts}
                if all(v[0] is not None for v in parts.values()):
                    traffic, src = sum(counts[k] * parts[k][0] for k in counts), parts["k_tl_inv"][1]
            roof = {"bound": "mfma_f32", "kernel": "ssr_fft_lowpass_multi launch sequence: k_tl_pad + k_tl_fwd<0> (once, cut 683) + 7 x (k_tl_inv + k_tl_fold)",
                    "achieved": round(ach, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach / 157.3, 5),
                    "traffic": traffic, "traffic_source": src, "useful_flops_per_launch_sequence": flops,
                    "algorithmic_bytes_per_launch_sequence": int(2 * N_SAMPLES * 4 * n * K),
                    "kernel_ms": round(ms_lp_all, 4),
                    "kernel_ms_source": "hip_events on the launch stream around the ONE C-ABI call (its kernels one by one: the rocprofv3 "
                                        "summary of the same command under profiles/)",
                    "note": "v_mfma_f32_32x32x2_f32 (exact float32 products in torch-CPU's conv1d accumulation order: the reference's "
                            "arithmetic, bit for bit): dense peak 157.3 TFLOP/s.  Useful flops per frame: forward 2 n_fft (2 x 683) ONCE for the 7 "
                            "keys + inverse 2 n_fft (2 K_k) per key, K = 2 cut - 1 channels of the mirrored spectrum; the sequence's time "
                            "includes the overlap-add (k_tl_fold) and padding kernels.  HBM-wise the same sequence moves 2 n 4 bytes per "
                            "(utterance, cutoff) algorithmically (SURVEY 8(d)): %.4f of 8 TB/s - not the bound." % (
                                2 * N_SAMPLES * 4 * n * K / (ms_lp_all * 1e-3) / 1e9 / HBM_PEAK_GBS)}
        else:
            stages = {"fft_lowpass": (ms_lp, 2 * N_SAMPLES * 4 * n), "pair_metrics_multi/7": (ms_multi / K, (2 * N_SAMPLES * 4 + 32) * n)}
            dom = max(stages, key=lambda k: stages[k][0])
            tkey = {"segments": "k_lowpass_wave+k_ola_paired", "fused": "k_lowpass_group"}[engine]
            lp_name = {"segments": "ssr_fft_lowpass(k_lowpass_wave+k_ola_paired)",
                       "fused": "ssr_fft_lowpass(k_lowpass_group: transforms + overlap-add in one kernel)"}[engine]
            roof = hbm_roofline(lp_name if dom == "fft_lowpass" else "ssr_pair_metrics_multi (7 keys, per key)", stages[dom][1], stages[dom][0],
                                tkey if dom == "fft_lowpass" and a.pairs == 1024 and a.precision == "f64" else None,
                                "per cutoff and 1024 utterances; algorithmic bytes: low-pass 2*n*4 per (utterance, cutoff), pair metrics "
                                "2*n*4+32 per pair (SURVEY 8(d))")
        extra = {"lowpass_engine": engine,
                 "stage_ms": {"fft_lowpass_one_cutoff(bin 256, own forward product)": round(ms_lp, 4), "fft_lowpass_multi_7_cutoffs": round(ms_lp_all, 4),
                              "pair_metrics_multi_7_keys": round(ms_multi, 4), "pair_metrics_one_key(round 3 path)": round(ms_pair, 4)},
                 "metric_stage_speedup_vs_7_pair_calls": round(K * ms_pair / ms_multi, 3),
                 "fft_lowpass_utterance_cutoffs_per_s": round(n * K / (ms_lp_all * 1e-3), 1)}
        if engine == "conv":
            extra["conv_engine_one_cutoff_useful_TFLOPs(bin 256)"] = round(self.conv_flops(n, [CUT_BINS[3]]) / (ms_lp * 1e-3) / 1e12, 1)
        if not getattr(a, "no_conv_side", False):
            # SIDE figures: the float64 / float32 FFT engines on the same batch - the exact low-pass, 2-7 % off the reference's LSD of
            # the degraded input (profiles/r05_cfg3_engines.json), for callers that want speed rather than the reference's numbers
            try:
                for label, prec in (("float64_fft_engine(segments)", a.precision), ("float32_fft_engine(SSR_F32 plan)", "f32")):
                    if engine != "conv" and prec == a.precision:
                        continue
                    fplan = B.get_plan(2048, 441, prec, self.dev, lowpass_engine="segments")
                    m2 = B.MultiLowpassBatch(fplan, tr, CUT_BINS, out=self.est.view(-1))
                    ms2 = event_time_ms(lambda: m2.run(), 2)
                    extra[label] = {"ms_7_cutoffs": round(ms2, 3), "cfg3_pairs_per_s_with_it": round(n * K / ((ms2 + ms_multi) * 1e-3), 1),
                                    "note": "NOT the reference's arithmetic: LSD of the degraded input 2-7 % high"}
                    del m2
                self.mlp.run()                           # (restore the timed engine's estimates)
            except Exception as e:                       # a side figure must not take the line down
                extra["side_engine_error"] = repr(e)
        return roof, extra

    def cpu_inputs(self, n):
        return [(self.tgt[i % self.a.pairs].cpu().numpy(), CUTOFFS_HZ[i % 7]) for i in range(n)]

    @staticmethod
    def cpu_unit(item):
        from oracle import lowpass as olp, metrics as om
        tgt, cutoff = item
        est = olp.lowpass(tgt, cutoff, SR, 1, "stft_hard")
        r = om.evaluation(est, tgt, n_fft=N_FFT, hop=HOP)
        return r["lsd"], r["ssim"]

    cpu_desc = "(utterance, cutoff) pairs of the same workload (oracle stft_hard low-pass 2048/441 = the published torchlibrosa conv1d arithmetic + 4 metrics at 2048/512)"

    def parity(self, out_vals, n, first):
        """CPU-baseline items first .. first + n - 1 (item i = target i mod pairs at cutoff i mod 7): max relative error of
        (LSD, SSIM) of the pair (HIP-degraded, target) against the oracle's metrics of THAT pair - the stop band of a
        low-passed signal is round-off, so the two sides must look at the same degraded signal - and, recorded on the
        side, the degraded signal itself against the oracle's torchlibrosa restatement (torch-CPU conv1d, >= 2 threads: samples
        that differ, max abs difference) and the PIPELINE (low-pass -> metrics) of the timed engine and of the float64 FFT engine
        against the reference's arithmetic."""
        from oracle import lowpass as olp, metrics as om
        B, worst, self.parity_lowpass_max_abs, self.parity_lowpass_samples_differing = self.B, 0.0, 0.0, 0
        dev_timed, dev_f64 = [0.0, 0.0], [0.0, 0.0]
        old_threads = torch.get_num_threads()
        torch.set_num_threads(max(2, old_threads))
        fplan = B.get_plan(2048, 441, self.a.precision, self.dev, lowpass_engine="segments")
        try:
            for i in range(first, first + n):
                j, c = i % self.a.pairs, i % 7
                t = B.Ragged.from_uniform(self.tgt[j:j + 1])
                lp = B.LowpassBatch(self.lp_plan, t, [CUT_BINS[c]])
                est = lp.run().cpu().numpy().copy()
                # (the timed step's own output plane holds the same samples)
                assert torch.equal(self.est[c, j].cpu(), torch.from_numpy(est)), "ssr_fft_lowpass_multi differs from ssr_fft_lowpass"
                tgt = self.tgt[j].cpu().numpy()
                got = B.PairBatch(self.plan, lp.out_ragged(), t).run(B.M_ALL)[0].cpu().numpy()
                want = om.evaluation(est, tgt, n_fft=N_FFT, hop=HOP)
                worst = max(worst, abs(got[0] - want["lsd"]) / abs(want["lsd"]), abs(got[3] - want["ssim"]) / abs(want["ssim"]))
                ref_est = olp.lowpass(tgt, CUTOFFS_HZ[c], SR, 1, "stft_hard")        # the published torchlibrosa arithmetic on torch-CPU
                self.parity_lowpass_max_abs = max(self.parity_lowpass_max_abs, float(np.abs(est - ref_est).max()))
                self.parity_lowpass_samples_differing += int((est != ref_est).sum())
                ref = om.evaluation(ref_est, tgt, n_fft=N_FFT, hop=HOP)
                lpf = B.LowpassBatch(fplan, t, [CUT_BINS[c]])
                lpf.run()
                gf = B.PairBatch(self.plan, lpf.out_ragged(), t).run(B.M_ALL)[0].cpu().numpy()
                for dst, g in ((dev_timed, got), (dev_f64, gf)):
                    dst[0] = max(dst[0], abs(g[0] / ref["lsd"] - 1))
                    dst[1] = max(dst[1], abs(g[1] - ref["log_sispec"]))
        finally:
            torch.set_num_threads(old_threads)
        self.pipeline_dev = {"timed_engine(%s)" % self.engine: {"lsd_rel_max": dev_timed[0], "log_sispec_abs_max_db": dev_timed[1]},
                             "float64_fft_engine(side figure)": {"lsd_rel_max": dev_f64[0], "log_sispec_abs_max_db": dev_f64[1]},
                             "lowpass_samples_differing_from_torch_conv1d": self.parity_lowpass_samples_differing,
                             "against": "published torchlibrosa low-pass (float32 conv1d, torch-CPU, >= 2 threads) -> oracle metrics, same targets"}
        return worst


class Cfg5:
    name = "cfg5"
    metric = "resampled-samples/sec (polyphase 16000->44100->48000 + LSD, 64000-sample utterances)"
    unit = "samples/s"
    N_IN = 64000

    def __init__(self, a, dev, rank):
        from ssr_eval_amd import backend as B
        self.B, self.a, self.dev = B, a, dev
        n = a.utterances
        g = torch.Generator(device=dev).manual_seed(20220328 + rank)
        self.x = (0.1 * torch.randn((n, self.N_IN), generator=g, device=dev, dtype=torch.float32)).contiguous()
        self.tgt = (0.1 * torch.randn((n, N_SAMPLES), generator=g, device=dev, dtype=torch.float32)).contiguous()
        # the two resample_poly stages in ONE kernel, the 44.1 kHz signal in LDS only (ssr_resample_poly_chain); --resample-chain
        # two-calls runs them as two ssr_resample_poly launches through an 8.8 GB intermediate
        self.chain = B.ResampleChainBatch(B.Ragged.from_uniform(self.x), 16000, 44100, 48000,
                                          fused=None if getattr(a, "resample_chain", "fused") == "fused" else False)
        self.s1, self.s2 = self.chain.s1, self.chain.s2
        assert int(self.s1.out_len[0]) == 176400
        assert int(self.s2.out_len[0]) == N_SAMPLES
        self.plan = B.get_plan(N_FFT, HOP, a.precision, dev)
        self.batch = B.PairBatch(self.plan, self.chain.out_ragged(), B.Ragged.from_uniform(self.tgt))
        self.units_per_step = n * N_SAMPLES
        self.cnt = torch.full((1,), float(n), dtype=torch.float64, device=dev)
        self.agg = torch.zeros(2, dtype=torch.float64, device=dev)

    def step(self):
        self.chain.run()
        out = self.batch.run(self.B.M_LSD)
        torch.cat([out[:, 0].sum(0, keepdim=True), self.cnt], out=self.agg)
        return self.agg

    def config(self, world):
        a = self.a
        return {"workload": "cfg-5: %d synthetic utterances of 64,000 float32 samples @ 16 kHz per GPU resident in HBM: polyphase "
                            "resample 441/160 (8821 taps) then 160/147 (3201 taps) -> 192,000 samples, LSD at STFT 2048/512 against a "
                            "48 kHz target; value counts the 192,000 output samples per utterance" % a.utterances,
                "utterances_per_gpu": a.utterances, "samples_in": self.N_IN, "samples_out": N_SAMPLES,
                "parallelism": "utterance-sharded x%d (100k utterances = 8 x 12,500), one float64 all-reduce (16 B) per step" % world}

    def report(self, a):
        it = 3
        n = a.utterances
        # the resampling chain's algorithmic bytes are 4*(n_in + n_out_final) = 1,024,000 B / utterance
        chain_alg = 4 * (64000 + 192000) * n
        msc = event_time_ms(lambda: self.chain.run(), it)
        fused = bool(self.chain.ran_fused)
        ms1 = event_time_ms(lambda: self.chain.run_stage1(), it)          # the two stages as two launches through HBM
        ms2 = event_time_ms(lambda: self.chain.run_stage2(), it)
        self.chain.run()
        ms3 = event_time_ms(lambda: self.batch.run(self.B.M_LSD), it)
        stages = {"resample_441_160": (ms1, 4 * (64000 + 176400) * n), "resample_160_147": (ms2, 4 * (176400 + 192000) * n),
                  "stft+lsd": (ms3, (2 * N_SAMPLES * 4 + 32) * n)}
        if fused:
            stages["resample_chain_fused"] = (msc, chain_alg)
            roof = hbm_roofline("ssr_resample_poly_chain (k_resample_chain: both stages, the 44.1 kHz signal in LDS)", chain_alg, msc,
                                "k_resample_chain" if a.utterances == 12500 else None,
                                "the resampling chain is the HBM-side kernel of this config; the two stages as separate launches and "
                                "the LSD stage are under extra.stage_ms / extra.stage_GBs")
        else:
            roof = hbm_roofline("ssr_resample_poly x2 (k_resample_rc, both stages)", chain_alg, ms1 + ms2,
                                "k_resample_rc stage 1+k_resample_rc stage 2" if a.utterances == 12500 else None,
                                "the resampling chain is the HBM-side kernel of this config; per-stage read+write rates and the LSD "
                                "stage are under extra.stage_ms / extra.stage_GBs")
        # the kernel most of the step's TIME goes to is the LSD stage's transform, not the chain: its roofline object rides along
        roof["time_dominant_kernel"] = hbm_roofline(
            "ssr_stft_pair(k_stft_wave<double, false, true, false>: resampled signal and target -> LSD)", (2 * N_SAMPLES * 4 + 32) * n, ms3,
            "k_stft_wave<double, false, true, false>" if a.utterances == 12500 else None,
            "%.0f %% of the step's kernel time; algorithmic bytes 2*n*4+32 per pair" % (100 * ms3 / ((msc if fused else ms1 + ms2) + ms3)))
        extra = {"stage_ms": {k: round(v[0], 4) for k, v in stages.items()},
                 "stage_GBs_read_plus_write": {k: round(v[1] / (v[0] * 1e-3) / 1e9, 1) for k, v in stages.items()},
                 "resample_chain": "fused (ssr_resample_poly_chain)" if fused else "two ssr_resample_poly calls",
                 "resample_only_output_samples_per_s": round(n * N_SAMPLES / ((msc if fused else ms1 + ms2) * 1e-3), 1)}
        # side figure: the matrix-core mode of the resampler (ssr_resample_poly_mfma: fused multiply-adds, not SciPy's bits; the
        # timed region above runs the bit-exact kernel).  Its outputs replace the exact ones for this measurement only.
        try:
            B = self.B
            lsd_exact = self.batch.run(B.M_LSD)[:, 0].clone()
            y_exact = self.s2.out.clone()
            self.s1.exact = self.s2.exact = False
            mm1 = event_time_ms(lambda: self.chain.run_stage1(), it)
            mm2 = event_time_ms(lambda: self.chain.run_stage2(), it)
            lsd_mfma = self.batch.run(B.M_LSD)[:, 0]
            extra["matrix_core_mode"] = {
                "stage_ms": {"resample_441_160": round(mm1, 4), "resample_160_147": round(mm2, 4)},
                "samples_per_s_with_lsd_stage": round(n * N_SAMPLES / ((mm1 + mm2 + ms3) * 1e-3), 1),
                "max_abs_sample_diff_vs_exact": float((self.s2.out - y_exact).abs().max()),
                "max_rel_lsd_diff_vs_exact": float(((lsd_mfma - lsd_exact).abs() / lsd_exact.abs()).max()),
                "note": "side figure: v_mfma_f32_32x32x2_f32 formulation (float32 fused multiply-adds in SciPy's order); not the "
                        "mode `value` is measured in"}
        finally:
            self.s1.exact = self.s2.exact = True
            self.chain.run()
        return roof, extra

    def cpu_inputs(self, n):
        return [(self.x[i % self.a.utterances].cpu().numpy(), self.tgt[i % self.a.utterances].cpu().numpy()) for i in range(n)]

    @staticmethod
    def cpu_unit(item):
        from scipy import signal
        from oracle import metrics as om
        x, tgt = item
        y = signal.resample_poly(signal.resample_poly(x, 441, 160), 160, 147)
        return float(om.lsd(om.wav_to_spectrogram(y, N_FFT, HOP), om.wav_to_spectrogram(tgt, N_FFT, HOP))), 0.0

    cpu_desc = "utterances of the same workload (scipy.signal.resample_poly 441/160 + 160/147, oracle STFT 2048/512 + LSD)"
    cpu_scale = N_SAMPLES          # value unit = output samples

    def parity(self, out_vals, n, first):
        """max relative LSD error of utterances first .. first + n - 1 of the full 12,500-utterance launch against the
        SciPy + oracle values of the CPU baseline for exactly those utterances."""
        self.step()
        got = self.batch.out[first:first + n, 0].cpu().numpy()
        return max(abs(got[i] - v[0]) / abs(v[0]) for i, v in enumerate(out_vals[:n]))


class Cfg4:
    name = "cfg4"
    metric = "utterance-pairs/sec (utterance-sharded VCTK-shaped test set: 2,937 ragged utterances, 8 speakers, full metric set, 48kHz, n_fft=2048)"
    unit = "pairs/s"
    scaling = "strong"
    own_collectives = True
    SPEAKER_COUNTS = [424, 424, 123, 419, 301, 424, 424, 398]          # files per VCTK test speaker
    SEED = 20220328

    @classmethod
    def layout(cls):
        """Lengths (samples @ 48 kHz, U(1.5 s, 9 s)) and speaker ids of the 2,937 utterances - the same on every rank."""
        n = sum(cls.SPEAKER_COUNTS)
        lens = (np.random.default_rng(cls.SEED).uniform(1.5, 9.0, n) * SR).astype(np.int64)
        spk = np.repeat(np.arange(len(cls.SPEAKER_COUNTS)), cls.SPEAKER_COUNTS)
        return lens, spk

    def __init__(self, a, dev, rank):
        from ssr_eval_amd import backend as B
        self.B, self.a, self.dev, self.rank = B, a, dev, rank
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        lens, spk = self.layout()
        self.n_total = len(lens)
        from ssr_eval_amd import dist as D
        self.shard = getattr(a, "shard", "balanced")
        # length-balanced dealing (longest first to the least-loaded rank) or static round-robin
        self.mine = D.shard_indices_balanced(lens, rank, self.world) if self.shard == "balanced" else np.arange(rank, self.n_total, self.world)
        loads = [int(lens[D.shard_indices_balanced(lens, r, self.world) if self.shard == "balanced" else np.arange(r, self.n_total, self.world)].sum())
                 for r in range(self.world)]
        self.shard_balance = max(loads) / (sum(loads) / len(loads))
        ml = lens[self.mine]
        off = np.concatenate(([0], np.cumsum(ml)[:-1])) if len(ml) else np.zeros(0, np.int64)
        tot = int(ml.sum())
        self.fake = bool(getattr(a, "cpu_skeleton", False))            # launcher / collective self-test on CPU (gloo): no kernels
        if self.fake:
            tot = 0
        self.est = torch.empty(tot, dtype=torch.float32, device=dev)
        self.tgt = torch.empty(tot, dtype=torch.float32, device=dev)
        g = torch.Generator(device=dev)
        for i, o, n in zip(self.mine if not self.fake else [], off, ml):   # one seed per utterance: the set does not depend on N
            g.manual_seed(self.SEED * 7 + int(i))
            t = self.tgt[o:o + n]
            torch.randn(int(n), generator=g, device=dev, out=t)
            t.mul_(0.1)
            e = self.est[o:o + n]
            torch.randn(int(n), generator=g, device=dev, out=e)
            e.mul_(0.01).add_(t)
        mk = lambda buf: B.Ragged(buf, torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ml.astype(np.int32)).to(dev), ml)
        if not self.fake:
            self.plan = B.get_plan(N_FFT, HOP, a.precision, dev)
            self.batch = B.PairBatch(self.plan, mk(self.est), mk(self.tgt))
        else:   # four "metrics" that are pure functions of the GLOBAL utterance index: any sharding must reproduce one aggregate
            gi = torch.from_numpy(self.mine.astype(np.float64)).to(dev)
            self.fake_out = torch.stack([gi, 0.5 * gi, gi * gi * 1e-3, torch.cos(gi)], dim=1)
        self.lens_local = ml
        self.collective = getattr(a, "collective", "allgather")
        self.n_spk = len(self.SPEAKER_COUNTS)
        self.spk_local = torch.from_numpy(spk[self.mine]).to(dev)
        self.cnt_local = torch.bincount(self.spk_local, minlength=self.n_spk).to(torch.float64)
        self.buf = torch.zeros((self.n_spk, 5), dtype=torch.float64, device=dev)      # [speakers, 4 metrics + count] (eval.py:200-216)
        sizes = [len(D.shard_indices_balanced(lens, r, self.world)) if self.shard == "balanced" else len(range(r, self.n_total, self.world))
                 for r in range(self.world)]
        self.cap = max(sizes)
        # ONE collective per step: the rank's padded (global index, 4 metrics) rows and, behind them, its
        # [speakers, 4 sums + count] block; every rank adds the gathered speaker blocks in rank order (bit-identical aggregate)
        self.pack = torch.full((self.cap + self.n_spk, 5), float("nan"), dtype=torch.float64, device=dev)
        self.pack[:self.cap, 0] = -1.0
        self.pack[:len(self.mine), 0] = torch.from_numpy(self.mine.astype(np.float64)).to(dev)
        self.gathered = torch.empty((self.world, self.cap + self.n_spk, 5), dtype=torch.float64, device=dev)
        self.units_per_step = self.n_total                              # strong scaling: the whole set per step, whatever N

    def step(self):
        import torch.distributed as dist
        out = self.fake_out if self.fake else self.batch.run(self.B.M_ALL)     # [n_local, 4]
        self.buf.zero_()
        self.buf[:, :4].index_add_(0, self.spk_local, out)
        self.buf[:, 4] = self.cnt_local
        self.pack[:len(self.mine), 1:] = out
        self.pack[self.cap:] = self.buf
        if self.world > 1 and self.collective == "allreduce":
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)                       # BASELINE cfg-4's collective: 320 B of float64 sums
        elif self.world > 1:
            dist.all_gather_into_tensor(self.gathered.view(-1, 5), self.pack)      # RCCL over xGMI: the step's only collective
            torch.sum(self.gathered[:, self.cap:], dim=0, out=self.buf)           # ranks added in rank order, on every rank alike
        return self.buf

    def job_means(self, agg):
        """mean over speakers of the per-speaker means (ssr_eval/eval.py:200-216), from the all-reduced buffer."""
        a = np.asarray(agg).reshape(self.n_spk, 5)
        ok = a[:, 4] > 0
        return (a[ok, :4] / a[ok, 4:5]).mean(axis=0).tolist()

    def config(self, world):
        a = self.a
        return {"workload": "cfg-4: the VCTK-shaped test set - %d ragged synthetic (est, target) pairs, 1.5-9 s @ 48 kHz, speakers with %s files - "
                            "sharded over the ranks (length-balanced dealing), resident in HBM; LSD + log-SISpec + SISpec + SSIM at STFT 2048/512, transform "
                            "precision %s; every step ends with ONE all-gather: the per-utterance rows and, behind them, the rank's "
                            "[8 speakers, 4 sums + count] block, added in rank order on every rank" % (sum(self.SPEAKER_COUNTS), self.SPEAKER_COUNTS, a.precision),
                "utterances_total": sum(self.SPEAKER_COUNTS), "n_fft": N_FFT, "hop": HOP,
                "parallelism": "strong scaling: %d utterances sharded x%d (%s dealing), ONE all-gather of %d B per rank and step"
                               % (sum(self.SPEAKER_COUNTS), world, getattr(a, "shard", "balanced"),
                                  (-(-sum(self.SPEAKER_COUNTS) // world) + 8 + len(self.SPEAKER_COUNTS)) * 40)}

    def report(self, a):
        import torch.distributed as dist
        B = self.B
        if self.fake:
            gathered = self.world > 1 and self.collective == "allgather"
            g = self.gathered[:, :self.cap].reshape(-1, 5).cpu().numpy() if gathered else self.pack[:self.cap].cpu().numpy()
            return None, {"allgather_rows_received": int((g[:, 0] >= 0).sum()), "shard_utterances": int(len(self.mine)),
                          "shard_balance_max_over_mean": round(self.shard_balance, 5), "collective": self.collective if self.world > 1 else None}
        ms_all = event_time_ms(lambda: self.batch.run(B.M_ALL), 3)
        ms_stft = event_time_ms(lambda: self.batch.run(B.M_ALL, stages=1), 3)
        alg = int((2 * self.lens_local * 4 + 32).sum())                  # each pair's own n
        roof = hbm_roofline("ssr_stft_pair(k_stft_wave, four metrics, ragged)", alg, ms_stft, None,
                            "rank 0's shard (%d of %d utterances); algorithmic bytes 2*n*4+32 per pair with each pair's own n"
                            % (len(self.mine), self.n_total))
        extra = {"shard_utterances": int(len(self.mine)), "shard_samples": int(self.lens_local.sum()),
                 "stage_ms_rank0": {"pair_metrics": round(ms_all, 4), "stft+lsd+sispec": round(ms_stft, 4)},
                 "shard": self.shard, "shard_balance_max_over_mean": round(self.shard_balance, 5),
                 "collective": self.collective if self.world > 1 else None,
                 "collective_payload_bytes_per_rank": int(self.pack.numel() * 8) if self.collective == "allgather" else int(self.buf.numel() * 8),
                 "collectives_per_step": 1 if self.world > 1 else 0}
        if self.world > 1:                                               # every rank calls report() for this workload
            scratch = self.buf.clone()
            extra["allgather_latency_us"] = round(1e3 * event_time_ms(lambda: dist.all_gather_into_tensor(self.gathered.view(-1, 5), self.pack), 20), 2)
            extra["allreduce_latency_us"] = round(1e3 * event_time_ms(lambda: dist.all_reduce(scratch, op=dist.ReduceOp.SUM), 20), 2)
            g = self.gathered[:, :self.cap].reshape(-1, 5).cpu().numpy()
            extra["allgather_rows_received"] = int((g[:, 0] >= 0).sum())
        return roof, extra

    def cpu_inputs(self, n):
        o = np.concatenate(([0], np.cumsum(self.lens_local)))
        return [(self.est[o[i]:o[i + 1]].cpu().numpy(), self.tgt[o[i]:o[i + 1]].cpu().numpy()) for i in range(min(n, len(self.mine)))]

    @staticmethod
    def cpu_unit(item):
        from oracle import metrics as om
        r = om.evaluation(item[0], item[1], n_fft=N_FFT, hop=HOP)
        return r["lsd"], r["ssim"], r["log_sispec"], r["sispec"]

    cpu_desc = "ragged pairs of the same workload (rank 0's first utterances; STFT 2048/512 + 4 metrics through the oracle)"

    def parity(self, out_vals, n, first):
        got = self.batch.run(self.B.M_ALL)[first:first + n].cpu().numpy()
        return max(max(abs(got[i, 0] - v[0]) / abs(v[0]), abs(got[i, 3] - v[1]) / abs(v[1])) for i, v in enumerate(out_vals[:n]))


class Skeleton:
    """No kernels: exercises the launcher, the barrier / MAX-over-ranks timing and the all-reduce on CPU (gloo) in
    tests/test_bench_skeleton.py.  Selected by --_cpu-skeleton only; never a measurement."""
    name = "skeleton"
    metric = "skeleton-units/sec (no kernels; launcher self-test)"
    unit = "units/s"
    parity = None

    def __init__(self, a, dev, rank):
        # units per rank and step as the selected config counts them (cfg5: 12,500 utterances x 192,000 output samples per GPU =
        # 100 k utterances over 8 GPUs; cfg2 / cfg3: the pairs), 10 for the plain launcher test
        self.a = a
        self.units_per_step = {"cfg5": a.utterances * N_SAMPLES, "cfg3": a.pairs * len(CUT_BINS)}.get(a.config, 10 if a.pairs == 1024 else a.pairs)
        self.agg = torch.tensor([float(rank + 1), 1.0], dtype=torch.float64, device=dev)
        self.rank = rank

    def step(self):
        self.agg[0], self.agg[1] = float(self.rank + 1), 1.0
        return self.agg

    def config(self, world):
        return {"workload": "skeleton", "parallelism": "x%d" % world}

    def report(self, a):
        return None, {}


WORKLOADS = {"cfg2": Cfg2, "cfg3": Cfg3, "cfg4": Cfg4, "cfg5": Cfg5, "apitrue": ApiTrue}

# ----------------------------------------------------------------------------------------------------------
# CPU baseline (SURVEY 8(d) protocol): >= 64 units where a unit is short, first 4 discarded as warm-up, median of 3
# repeats; (i) 1 process / 1 thread, (ii) one process per host core.  Workers inherit the inputs by fork (copy-on-write)
# and receive only indices: shipping 1.5 MB waveforms through the pool's pipe was what capped round 1's 64-process
# figure at 3.7x one thread.
_CPU_ITEMS, _CPU_FN = None, None


def _cpu_run(idx):
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    vals = [_CPU_FN(_CPU_ITEMS[i]) for i in idx]
    return time.perf_counter() - t0, vals


def _usable_cores():
    """Host cores this process may actually use: os.cpu_count() capped by the scheduler affinity mask and by the cgroup
    CPU quota (a container that SEES 256 logical CPUs but is throttled to a few cores' worth of time runs a 256-process
    pool slower than a small one)."""
    n = os.cpu_count() or 1
    info = {"os_cpu_count": n}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
        n = min(n, info["affinity"])
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                info["cgroup_quota_cores"] = round(float(quota) / period, 2)
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except Exception:
            continue
    return n, info


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(wl, budget_s=24.0):
    import multiprocessing as mp
    global _CPU_ITEMS, _CPU_FN
    _CPU_FN = wl.cpu_unit
    probe = wl.cpu_inputs(2)
    _CPU_ITEMS = probe
    torch.set_num_threads(1)
    _cpu_run([0])
    t_unit = _cpu_run([1])[0]
    scale = getattr(wl, "cpu_scale", 1)
    # (i) one thread
    n1 = 64 if t_unit * 68 * 3 <= budget_s * 0.4 else max(4, int(budget_s * 0.4 / 3 / t_unit) - 4)
    cores, core_info = _usable_cores()
    per_worker = 2 if t_unit * 2 * 3 * 2 <= budget_s * 0.5 else 1
    n_pool = max(64, cores * per_worker)
    _CPU_ITEMS = wl.cpu_inputs(max(n1 + 4, min(n_pool, 512)))
    n_items = len(_CPU_ITEMS)
    reps1, vals = [], None
    for _ in range(3):
        _cpu_run(list(range(4)))                                        # discarded warm-up
        dt, v = _cpu_run([4 + i for i in range(n1)])
        vals = vals or v
        reps1.append(n1 / dt)
    rate1 = float(np.median(reps1)) * scale
    # (ii) one process per core, fork: inputs are inherited
    rate_pool, err = None, None
    try:
        ctx = mp.get_context("fork")
        chunks = [[(w * per_worker + j) % n_items for j in range(per_worker)] for w in range(max(cores, n_pool // per_worker))]
        with ctx.Pool(cores) as pool:
            pool.map(_cpu_run, [[i % n_items] for i in range(cores)])   # warm the workers (discarded)
            reps = []
            for _ in range(3):
                t0 = time.perf_counter()
                pool.map(_cpu_run, chunks, chunksize=1)
                reps.append(sum(len(c) for c in chunks) / (time.perf_counter() - t0))
            rate_pool = float(np.median(reps)) * scale
    except Exception as e:                                              # a locked-down box: report the 1-thread number only
        err = repr(e)
    out = {"value": round(rate_pool if rate_pool else rate1, 3), "unit": wl.unit, "cores": cores if rate_pool else 1, "kind": "port",
           "cpu_model": _cpu_model(), "host_cores": core_info, "value_1thread": round(rate1, 3),
           "sample": "%s: 1 thread = median of 3 repeats of %d units after 4 discarded; %d processes (usable host cores, 1 thread "
                     "each, inputs inherited by fork) = median of 3 repeats of %d units after one discarded unit per worker"
                     % (wl.cpu_desc, n1, cores, sum(len(c) for c in chunks) if rate_pool else 0)}
    if err:
        out["pool_error"] = err
    return out, vals


# ----------------------------------------------------------------------------------------------------------
def side_figures(a, dev):
    """Short measurements of the other BASELINE configs, the API-true STFT size and the end-to-end helper (N = 1 only,
    outside the timed region) -> dict for `extra`."""
    from ssr_eval_amd import backend as B
    ex = {}

    def guarded(name, fn):
        try:
            ex[name] = fn()
        except Exception as e:  # pragma: no cover
            ex[name] = {"error": repr(e)}
        torch.cuda.empty_cache()

    def api_true():
        wl = ApiTrue(a, dev, 0)
        wl.step()
        torch.cuda.synchronize()
        ms = event_time_ms(lambda: wl.step(), 3)
        roof, extra = wl.report(a)
        return {"workload": wl.config(1)["workload"], "pairs_per_s": round(a.pairs / (ms * 1e-3), 1), "ms_per_step": round(ms, 4),
                "roofline": roof, "kernels_ms": extra["kernels_ms"],
                "note": "side figure: average of 3 steps in one short run; `python bench.py --config apitrue` is the measurement "
                        "(its rocprofv3 summary: profiles/r06_apitrue_kernel_stats.csv)"}

    def rates():
        """Every AudioMetrics(rate) size of the reference (ssr_eval/metrics.py:16-19), 4 s signals at that rate, four metrics."""
        out = {}
        nb = min(a.pairs, 512)
        for rate in (16000, 24000, 32000, 44100, 48000):
            hop, n_fft = int(rate / 100), int(2048 / (44100 / rate))
            g = torch.Generator(device=dev).manual_seed(rate)
            tgt = (0.1 * torch.randn((nb, 4 * rate), generator=g, device=dev)).contiguous()
            est = (tgt + 0.01 * torch.randn((nb, 4 * rate), generator=g, device=dev)).contiguous()
            b2 = B.PairBatch(B.get_plan(n_fft, hop, a.precision, dev), B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
            b2.run(B.M_ALL)
            ms = event_time_ms(lambda: b2.run(B.M_ALL), 3)
            out[str(rate)] = {"n_fft": n_fft, "hop": hop, "pairs_per_s": round(nb / (ms * 1e-3), 1)}
            del b2, est, tgt
            torch.cuda.empty_cache()
        return {"workload": "%d pairs of 4 s per rate, LSD + log-SISpec + SISpec + SSIM" % nb, "rates": out,
                "note": "side figures: average of 3 launches per rate in one short run"}

    def other(cfg, steps):
        def run():
            wl = WORKLOADS[cfg](a, dev, 0)
            for _ in range(1):
                wl.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                wl.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            roof, extra = wl.report(a)
            parity = None
            if getattr(wl, "parity", None) is not None and not a.no_cpu_baseline:
                # a handful of this config's own units through the oracle (no CPU rate is reported for a side figure)
                global _CPU_ITEMS, _CPU_FN
                _CPU_FN, _CPU_ITEMS = wl.cpu_unit, wl.cpu_inputs(3)
                parity = float(wl.parity(_cpu_run(list(range(len(_CPU_ITEMS))))[1], len(_CPU_ITEMS), 0))
            return {"metric": wl.metric, "value": round(wl.units_per_step * steps / dt, 1), "unit": wl.unit, "steps": steps,
                    "ms_per_step": round(dt / steps * 1e3, 3), "config": wl.config(1), "roofline": roof, "extra": extra,
                    "parity_vs_oracle_max_rel_err": parity,
                    "note": "side figure: ONE short run (%d steps after 1 warm-up) outside the headline's timed region; "
                            "`python bench.py --config %s` is the measurement" % (steps, cfg)}
        return run

    def e2e():
        import tempfile
        import shutil
        from ssr_eval_amd import SSR_Eval_Helper, BasicTestee
        from ssr_eval_amd.io import write_wav
        rng = np.random.default_rng(4)
        root = tempfile.mkdtemp(prefix="ssr_e2e_")
        try:
            n_files = 0
            counts = [53, 53, 15, 52, 38, 53, 53, 50]           # cfg-4's per-speaker file counts (424 ... 398), one eighth
            for s, c in enumerate(counts):
                os.makedirs(os.path.join(root, "p%03d" % (360 + s)))
                for i in range(c):
                    n = int(rng.integers(int(1.5 * 44100), 9 * 44100))
                    write_wav(os.path.join(root, "p%03d" % (360 + s), "u%03d.wav" % i), 0.1 * rng.standard_normal(n), 44100)
                    n_files += 1
            h = SSR_Eval_Helper(BasicTestee(), input_sr=44100, output_sr=44100, evaluation_sr=48000, test_data_root=root,
                                setting_fft={"cutoff_freq": [12000]})
            h.evaluate(limit_test_nums=2, limit_test_speaker=1, save_json=False)        # warm-up (plans, taps)
            h.evaluate(save_json=False)                                                 # warm-up (allocator, page cache)
            times = []
            for _ in range(7):
                t0 = time.perf_counter()
                res = h.evaluate(save_json=False)
                times.append(time.perf_counter() - t0)
            dt = float(np.median(times))
            # stage breakdown: one more pass with synchronising wrappers around the stage functions (wall clock per stage,
            # GPU work included; the wrappers serialise what the timed passes overlap, so the stages sum to more than a pass)
            import collections
            from ssr_eval_amd import io as IO, backend as Bk, eval as EV, metrics as MT
            acc, saved = collections.OrderedDict(), []

            def timed(mod, name, label):
                f = getattr(mod, name)

                def g(*a_, **k_):
                    torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a_, **k_); torch.cuda.synchronize()
                    acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
                    return r
                saved.append((mod, name, f))
                setattr(mod, name, g)
            timed(Bk, "upload_decoded", "upload (int16 PCM, pinned, + GPU conversion)")
            timed(IO, "to_rate_resident", "rate changes (kaiser_best, targets + inputs)")
            timed(EV.SSR_Eval_Helper, "preprocess_arrays", "degradation (FFT low-pass)")
            timed(Bk, "resample_poly", "polyphase resample to the evaluation rate")
            timed(MT.AudioMetrics, "evaluation_batch", "four metrics")
            timed(EV.SSR_Eval_Helper, "_assemble", "aggregation")
            try:
                t0 = time.perf_counter()
                h.evaluate(save_json=False)
                acc["whole pass with the wrappers"] = time.perf_counter() - t0
            finally:
                for mod, name, f in saved:
                    setattr(mod, name, f)
            return {"workload": "SSR_Eval_Helper.evaluate() on %d PCM .wav files (8 speakers with cfg-4's file-count proportions, "
                                "1.5-9 s @ 44.1 kHz), identity testee, setting_fft cutoff 12 kHz, evaluation_sr 48000: host decode + H2D + "
                                "resample + low-pass + 4 metrics + aggregation" % n_files,
                    "files_per_s": round(n_files / dt, 1), "seconds": round(dt, 4),
                    "seconds_all_passes": [round(t, 4) for t in times],
                    "stage_seconds": {k: round(v, 4) for k, v in acc.items()},
                    "note": "median of 7 passes after two warm-up passes; host-side timings vary by +-40 % between boxes",
                    "averaged_lsd": float(res["averaged"]["proc_fft_24000_44100"]["lsd"])}
        finally:
            shutil.rmtree(root, ignore_errors=True)

    guarded("api_true_2229_480", api_true)
    guarded("audio_metrics_rates", rates)
    guarded("cfg3", other("cfg3", 2))
    guarded("cfg4", other("cfg4", 2))
    guarded("cfg5", other("cfg5", 2))
    guarded("evaluate_end_to_end", e2e)
    return ex


def run(a):
    from ssr_eval_amd import dist as D
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch with --nproc-per-node == --gpus)" % (a.gpus, world))
    if a.cpu_skeleton:
        dev, sync, backend = torch.device("cpu"), (lambda: None), "gloo"
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        if torch.cuda.device_count() <= local_rank and not a.share_gpu:
            raise SystemExit("bench.py: rank %d has no device (%d visible)" % (local_rank, torch.cuda.device_count()))
        if a.share_gpu:
            # TEST HOOK (tests/test_gpu_configs.py): every rank on cuda:0 with gloo for the collectives - the one-GPU box of the test suite
            # runs the real workloads, the cfg-4 side figure and the rank-0 report under world_size > 1; never a measurement
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev, sync, backend = torch.device("cuda", local_rank), torch.cuda.synchronize, ("gloo" if a.share_gpu else "nccl")
    if world > 1:
        D.init_from_env(backend)
        joined = dist.get_world_size()                # the ranks that actually joined the group
        if joined != a.gpus:
            raise SystemExit("bench.py: %d ranks joined the %s group, expected %d" % (joined, backend, a.gpus))
    else:
        joined = 1

    wl = (Skeleton if (a.cpu_skeleton and a.config != "cfg4") else WORKLOADS[a.config])(a, dev, rank)
    wl_cls = type(wl)
    strong = getattr(wl_cls, "scaling", "weak") == "strong"

    def timed(w, steps, warmup):
        """`warmup` untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; MAX over ranks."""
        # The step's one collective (a few dozen bytes of float64 sums) is issued asynchronously on RCCL's own stream against a
        # private copy of the sums, so the kernels of step k + 1 run under the all-reduce of step k; a reduction is waited for two
        # steps later, and every one of them before the closing synchronize - the timed region contains all K collectives.
        overlap = world > 1 and not getattr(w, "own_collectives", False)
        pending = []

        def step():
            agg = w.step()
            if overlap:
                red = agg.clone()
                pending.append((dist.all_reduce(red, op=dist.ReduceOp.SUM, async_op=True), red))
                if len(pending) > 2:
                    pending.pop(0)[0].wait()
                return red
            return agg

        def drain():
            while pending:
                pending.pop(0)[0].wait()
        for _ in range(warmup):
            step()
        drain()
        sync()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            agg = step()
        drain()
        sync()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            every = [torch.zeros_like(tmax) for _ in range(world)]
            dist.all_gather(every, tmax)                 # (outside the timed region: the per-rank clocks for `extra`)
            per_rank[:] = [float(t.item()) for t in every]
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, agg.cpu().numpy()

    per_rank = []
    elapsed, agg = timed(wl, a.steps, a.warmup)
    per_rank_ms = [round(t / a.steps * 1e3, 4) for t in per_rank]
    # weak scaling: every rank owns units_per_step units; strong scaling: units_per_step is the whole job's
    value = wl.units_per_step * (1 if strong else joined) * a.steps / elapsed
    means = wl.job_means(agg) if hasattr(wl, "job_means") else (agg[:-1] / agg[-1]).tolist()
    payload = int(agg.nbytes)

    # the fixed-size sharded test set next to a weak-scaling headline at N > 1 (every rank takes part; rank 0 reports)
    cfg4_side = None
    if world > 1 and a.config == "cfg2" and not a.no_side and not a.cpu_skeleton:
        w4 = Cfg4(a, dev, rank)
        dt4, agg4 = timed(w4, 3, 1)
        roof4, extra4 = w4.report(a)
        cfg4_side = {"metric": Cfg4.metric, "value": round(w4.units_per_step * 3 / dt4, 1), "unit": Cfg4.unit, "scaling": "strong",
                     "n_gpus": joined, "steps": 3, "ms_per_step": round(dt4 / 3 * 1e3, 3), "config": w4.config(joined),
                     "roofline": roof4, "extra": dict(extra4, job_means=w4.job_means(agg4)),
                     "note": "side figure: one short run (3 steps after 1 warm-up), outside the timed region of the headline"}
        del w4
        torch.cuda.empty_cache()

    if rank != 0:
        if getattr(wl_cls, "own_collectives", False) and world > 1:
            wl.report(a)                                                # its report times the collectives: every rank takes part
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    roofline, extra = wl.report(a)
    extra["job_means"] = means
    extra["allreduce_payload_bytes_per_step"] = payload if world > 1 else 0
    if world > 1:
        # what a reader of the scaling curve needs to trust it: the ranks that joined, each rank's own clock, the library underneath
        rccl = None
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception as e:
            rccl = repr(e)
        extra["ranks"] = {"joined": joined, "backend": backend, "rccl_version": rccl, "ms_per_step_per_rank": per_rank_ms,
                          "max_over_min": round(max(per_rank_ms) / max(min(per_rank_ms), 1e-9), 4) if per_rank_ms else None,
                          "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", world))}
    if cfg4_side is not None:
        extra["cfg4_strong_scaling"] = cfg4_side
    cpu = None
    if not a.no_cpu_baseline and world == 1 and not a.cpu_skeleton:      # contract: the CPU baseline is timed on rank 0 at N = 1 only
        cpu, vals = cpu_baseline(wl)
        if getattr(wl, "parity", None) is not None:
            extra["parity_vs_oracle_max_rel_err"] = float(wl.parity(vals, min(4, len(vals)), 4))   # vals[i] is item 4 + i
            if hasattr(wl, "parity_lowpass_max_abs"):
                extra["parity_lowpass_max_abs_err_vs_oracle"] = wl.parity_lowpass_max_abs
            if hasattr(wl, "pipeline_dev"):
                extra["pipeline_deviation_from_reference_arithmetic"] = wl.pipeline_dev
    if world == 1 and a.config == "cfg2" and not a.no_side and not a.cpu_skeleton:
        del wl
        torch.cuda.empty_cache()
        extra.update(side_figures(a, dev))

    line = {"metric": wl_cls.metric, "value": round(value, 2), "unit": wl_cls.unit, "n_gpus": joined,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": _config_of(wl_cls, a, joined),
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _config_of(cls, a, world):
    """The config description without building the workload again."""
    shell = cls.__new__(cls)
    shell.a = a
    return cls.config(shell, world)


def _spawned(local_rank, a, port):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    run(a)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=1024, help="cfg2/cfg3: pairs (targets) per GPU per step (BASELINE: 1024)")
    ap.add_argument("--utterances", type=int, default=12500, help="cfg5: utterances per GPU per step (100k over 8 GPUs)")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--lowpass-engine", dest="lowpass_engine", default=None, choices=["segments", "fused", "conv"],
                    help="cfg3: engine of the STFT-domain low-pass; default = the product's (ssr_eval_amd.lowpass.DEFAULT_ENGINE = conv: the "
                         "reference's float32 dense-DFT arithmetic on the matrix cores); segments / fused = float64 FFT (exact, not the reference's)")
    ap.add_argument("--resample-chain", dest="resample_chain", default="fused", choices=["fused", "two-calls"],
                    help="cfg5: both resample_poly stages in one kernel (ssr_resample_poly_chain) or two ssr_resample_poly launches")
    ap.add_argument("--shard", default="balanced", choices=["balanced", "round-robin"], help="cfg4: how the fixed set is dealt to the ranks")
    ap.add_argument("--collective", default="allgather", choices=["allgather", "allreduce"],
                    help="cfg4: the step's one collective - allgather (default): per-utterance rows + per-speaker sums, added in rank order "
                         "on every rank (what evaluate() needs for its JSON); allreduce: the per-speaker [sums, count] block only, one RCCL "
                         "all-reduce (the collective BASELINE cfg-4 names)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the cfg3 / cfg5 / API-true / end-to-end side figures")
    ap.add_argument("--_cpu-skeleton", dest="cpu_skeleton", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--_share-gpu", dest="share_gpu", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one process per GPU.  Never degrade to fewer ranks than asked for.
        n_dev = torch.cuda.device_count()
        if n_dev < a.gpus and not a.cpu_skeleton and not a.share_gpu:
            raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) are visible" % (a.gpus, n_dev))
        import torch.multiprocessing as mp
        port = 29400 + os.getpid() % 2000
        mp.spawn(_spawned, args=(a, port), nprocs=a.gpus, join=True)
        return
    run(a)


if __name__ == "__main__":
    main()
