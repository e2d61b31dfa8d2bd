"""Developer tool (CPU): REAL members of the reference's low-pass arithmetic class.

The published torchlibrosa ISTFT is a float32 GEMM whose accumulation order the BLAS kernel decides.  Every float32 GEMM this
image has is run on the same low-pass and the metrics of the degraded signal are tabulated:
  conv_t{1,2,4,8,16}   torch F.conv1d (oneDNN) at that many threads            - the code path the reference itself takes
  mm_t{1,8}            torch.mm on the same matrices (MKL sgemm)
  np_openblas          numpy float32 @ (OpenBLAS sgemm)
  fold_mm_t{1,8}, fold_np   the same libraries on the Hermitian-FOLDED inverse (bins 0..c-1 only, weights of bins >= 1 doubled:
                       half the flops)
  blocks{64,...,2048}  oracle/tl_chain.c: fused-multiply-add chains over blocks of that many channels of the FULL spectrum
                       (256 = what the HIP conv engine runs; it is conv_t{2..16} bit for bit)
Prints per cut: LSD relative to conv at 8 threads (per cent), and for the restated members the fraction of samples that differ from
conv_t8; writes the table (with log-SISpec differences in dB) to $OUT (default profiles/r05_lowpass_class_members.json).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stft as ostft, metrics as om, tl_chain  # noqa: E402

CUTS = [42, 85, 170, 256, 341, 512, 683]
N_FFT, HOP = 2048, 441


def forward(x):
    re, im = ostft.tl_stft_conv(x[None])
    mag = np.clip(re ** 2 + im ** 2, np.float32(1e-8), np.inf) ** np.float32(0.5)
    c, s_ = re / mag, im / mag
    return mag * c, mag * s_            # [1,1,T,F]


def finish(s, T, length):
    """s [n_fft, T] torch float32 -> waveform (fold, window-sum division, trim: torchlibrosa's own ops)."""
    L = (T - 1) * HOP + N_FFT
    y = Fn.fold(s[None], output_size=(1, L), kernel_size=(1, N_FFT), stride=(1, HOP))[0, 0, 0, :]
    y = y / ostft.tl_window_sum_f32(T, N_FFT, HOP)
    return y[N_FFT // 2:N_FFT // 2 + length].numpy()


def inverse_members(R, I, cut, length):
    """R, I: [T, F] float32 (not yet cut)."""
    T = R.shape[0]
    R = R.copy(); I = I.copy()
    R[:, cut:] = 0; I[:, cut:] = 0
    _, _, ir, ii = ostft.tl_weights(N_FFT)              # [n_fft(sample), n_fft(channel)]
    re = torch.from_numpy(R.T.copy())                   # [F, T]
    im = torch.from_numpy(I.T.copy())
    full_re = torch.cat((re, torch.flip(re[1:-1], dims=[0])), dim=0)      # [n_fft, T]
    full_im = torch.cat((im, -torch.flip(im[1:-1], dims=[0])), dim=0)
    ir_t, ii_t = torch.from_numpy(ir), torch.from_numpy(ii)
    out = {}
    old = torch.get_num_threads()
    for th in (1, 2, 4, 8, 16):
        torch.set_num_threads(th)
        s = Fn.conv1d(full_re[None], ir_t[:, :, None]) - Fn.conv1d(full_im[None], ii_t[:, :, None])
        out["conv_t%d" % th] = finish(s[0], T, length)
    for th in (1, 8):
        torch.set_num_threads(th)
        s = torch.mm(ir_t, full_re) - torch.mm(ii_t, full_im)
        out["mm_t%d" % th] = finish(s, T, length)
        # compact: only the non-zero channels (what a sparse-aware caller would hand the same sgemm)
        nz = np.r_[0:cut, N_FFT - (cut - 1):N_FFT]
        s = torch.mm(ir_t[:, nz].contiguous(), full_re[nz].contiguous()) - torch.mm(ii_t[:, nz].contiguous(), full_im[nz].contiguous())
        out["mmcompact_t%d" % th] = finish(s, T, length)
        # folded: bins 0..cut-1, doubled weights for bins >= 1
        g = np.full(cut, 2.0, np.float32); g[0] = 1.0
        wr = torch.from_numpy(ir[:, :cut] * g[None, :]); wi = torch.from_numpy(ii[:, :cut] * g[None, :])
        s = torch.mm(wr, re[:cut].contiguous()) - torch.mm(wi, im[:cut].contiguous())
        out["fold_mm_t%d" % th] = finish(s, T, length)
    torch.set_num_threads(old)
    s = ir @ full_re.numpy() - ii @ full_im.numpy()
    out["np_openblas"] = finish(torch.from_numpy(s), T, length)
    s = (ir[:, :cut] * g[None, :]) @ R.T[:cut] - (ii[:, :cut] * g[None, :]) @ I.T[:cut]
    out["fold_np"] = finish(torch.from_numpy(np.ascontiguousarray(s)), T, length)
    for kb in (64, 128, 256, 384, 512, 2048):
        out["blocks%d" % kb] = tl_chain.istft(R, I, length, N_FFT, HOP, kbf=kb, nbz=cut)
    return out


def main():
    n_sig = int(os.environ.get("NSIG", 4))
    table = []
    for si in range(n_sig):
        rng = np.random.default_rng(20220328 + si)
        x = (0.1 * rng.standard_normal(48000)).astype(np.float32)
        Rf, If = forward(x)
        for cut in CUTS:
            mem = inverse_members(Rf[0, 0], If[0, 0], cut, len(x))
            ms = {}
            for k, y in mem.items():
                m = om.evaluation(y.astype(np.float32), x, n_fft=2048, hop=512)
                ms[k] = (m["lsd"], m["log_sispec"])
            ref = ms["conv_t8"]
            row = {"sig": si, "cut": cut, "ref_lsd": ref[0], "members[lsd_rel, log_sispec_db, samples_differing_from_conv_t8]": {}}
            for k, v in ms.items():
                row[k] = (v[0] / ref[0] - 1, v[1] - ref[1])
                row["members[lsd_rel, log_sispec_db, samples_differing_from_conv_t8]"][k] = [
                    v[0] / ref[0] - 1, v[1] - ref[1], float((mem[k] != mem["conv_t8"]).mean())]
            table.append(row)
            real = [k for k in ms if k.startswith(("conv", "mm_", "mmcompact", "np_"))]
            lsds = [ms[k][0] for k in real]
            print("sig %d cut %3d | real spread %.3f %% | " % (si, cut, 100 * (max(lsds) - min(lsds)) / ref[0]) +
                  " ".join("%s %+.2f" % (k, 100 * row[k][0]) for k in ms if k != "conv_t8"), flush=True)
    real = lambda k: k.startswith(("conv", "mm_", "np_"))          # noqa: E731  (full-matrix products of real libraries)
    spread = max(max(r[k][0] for k in r if isinstance(r[k], tuple) and real(k)) - min(r[k][0] for k in r if isinstance(r[k], tuple) and real(k))
                 for r in table)
    hip = max(r["members[lsd_rel, log_sispec_db, samples_differing_from_conv_t8]"]["blocks256"][2] for r in table)
    slim = [{k: v for k, v in r.items() if not isinstance(v, tuple)} for r in table]
    with open(os.environ.get("OUT", os.path.join(ROOT, "profiles", "r05_lowpass_class_members.json")), "w") as f:
        json.dump({"note": "LSD (relative to torch F.conv1d at 8 threads) / log-SISpec (dB difference) of a hard-low-passed 1 s noise target, "
                           "48 kHz, metrics at 2048/512, per float32 GEMM implementation of torchlibrosa's ISTFT; torch %s, numpy %s, %d cores" % (
                               torch.__version__, np.__version__, os.cpu_count()),
                   "largest_lsd_spread_of_real_full_matrix_members": spread,
                   "largest_fraction_of_samples_where_blocks256_differs_from_conv_t8": hip, "rows": slim}, f, indent=1)
    print("largest LSD spread of the real full-matrix members: %.3f %%; blocks256 differs from conv_t8 in %.4f of the samples" % (100 * spread, hip))


if __name__ == "__main__":
    main()
