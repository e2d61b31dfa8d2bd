// Kernel body K6 on the wave-autonomous engine (ssr_stft_wave.h): STFT-domain hard low-pass / inverse STFT with ONE WAVE
// PER PAIR OF FRAMES.  Semantics are those of ssr_lowpass.h (ssr_eval/lowpass.py:17-28, ssr_eval/dsp.py:76-119).
//
//   frames 2g / 2g+1 as re / im of one complex sequence -> forward FFT-2048 -> zero cut <= k <= N - cut
//   -> inverse FFT-2048 -> synthesis window / N -> frames[t][m]        (k_ola then overlap-adds, ssr_ola_sample)
//
// The forward transform leaves lane l with Z[k], k = l + 64 (b + 4 q) - every k congruent to l mod 64, which is exactly
// the set the in-register first pass of the NEXT transform wants: the inverse transform starts from the registers the
// forward one ended in (a compile-time renaming), with no exchange in between.  The inverse is the forward engine on
// exchanged (im, re) parts.  ISTFT mode builds the Hermitian-extended packed spectrum straight into those registers.
//
// PAIRED = true (hop <= n_fft / 2) adds the unit's two frames up before they leave the wave: frame 2g is parked in the
// exchange array and read back hop samples further on, so the unit writes ONE segment of n_fft + hop samples
//     seg[j] = frame_2g[j] (j < n_fft)  +  frame_2g+1[j - hop] (j >= hop)
// instead of two frames of n_fft (61 % of the bytes at hop 441, for k_ola_paired to read back: <= 3 terms per sample
// instead of <= 5).  The sum is formed in T and rounded to float32 once.  Row layout: ssr_seg_stride (ssr_lowpass.h).
#pragma once
#include "ssr_lowpass.h"
#include "ssr_stft_wave.h"

template <typename T> struct SsrLowpassWaveRegs {
  cx<T> v[SSR_W_P];
  T tx[SSR_W_P];
  float pa[SSR_W_P], pb[SSR_W_P];     // the unit's samples (analysis mode)
  T wl[SSR_W_P / 2];                  // hann[tid + 64 r], r < 16  (hann[m + N/2] = 1 - hann[m]); requested twice per unit -
                                      // ahead of the analysis and ahead of the synthesis - instead of living through it
  cx<T> tw1[7];
  cx<T> tw2[12];
};

SSR_DEV int ssr_neg_mask(int x) { return x >> 31; }               // all ones iff x < 0
SSR_DEV double ssr_and_not(double v, int mask) {                  // v with its bits cleared where mask is all ones
  unsigned long long u;
  memcpy(&u, &v, 8);
  u &= ~(((unsigned long long)(unsigned)mask << 32) | (unsigned)mask);
  memcpy(&v, &u, 8);
  return v;
}
SSR_DEV float ssr_and_not(float v, int mask) {
  unsigned u;
  memcpy(&u, &v, 4);
  u &= ~(unsigned)mask;
  memcpy(&v, &u, 4);
  return v;
}

// samples of frames 2g / 2g+1 (reflect-padded the torch way: callers guarantee len > N/2) into the prefetch registers
template <typename T, typename REGS>
SSR_DEV void ssr_lowpass_wave_prefetch(REGS& R, int tid, const SsrView<float>& vs, int g, int hop, int n, int n_frames) {
  const int ta = (2 * g < n_frames) ? 2 * g : n_frames - 1;
  const int tb = (2 * g + 1 < n_frames) ? 2 * g + 1 : n_frames - 1;      // a missing frame re-reads the last one
  const int base_a = ta * hop - SSR_W_N / 2, base_b = tb * hop - SSR_W_N / 2;
  if (base_a >= 0 && base_b + SSR_W_N <= n) {                             // wave-uniform: both frames inside the signal
    SSR_UNROLL for (int r = 0; r < SSR_W_P; ++r) {
      R.pa[r] = vs.at(SSR_UIDX(tid + 64 * r), base_a);
      R.pb[r] = vs.at(SSR_UIDX(tid + 64 * r), base_b);
    }
  } else {
    SSR_UNROLL for (int r = 0; r < SSR_W_P; ++r) {
      R.pa[r] = vs.at(SSR_UIDX(ssr_reflect(base_a + tid + 64 * r, n)));
      R.pb[r] = vs.at(SSR_UIDX(ssr_reflect(base_b + tid + 64 * r, n)));
    }
  }
}

// grid = n_items * n_chunks workgroups of one wave; 2048-point plans only.
template <typename T, bool SPLIT, bool ANALYSIS, bool PAIRED, typename BLK>
SSR_BODY void ssr_lowpass_wave_body(const SsrLowpassParams<T>& p, BLK& blk, int chunk, int item, char* lds_base) {
  constexpr int N = SSR_W_N, F = N / 2 + 1;
  using Regs = SsrLowpassWaveRegs<T>;
  SsrWaveLds<T, SPLIT> L(lds_base);
  const int n = p.len[item], hop = p.hop;
  if (n <= N / 2) return;     // precondition (include/ssr_hip.h): len > n_fft / 2; such an item is skipped (k_ola* zero its output)
  const int n_frames = ssr_num_frames_dev(n, N, hop);
  const int n_pairs = (n_frames + 1) / 2;
  // this chunk's frame pairs: g0, g0 + S, ... (< g1); S > 1: the S chunks of a group interleave over the group's span and run
  // at the same time on one XCD, so the 78 % overlap of neighbouring frames comes from its L2 (ssr_stft_wave.h)
  const int S = p.interleave > 1 ? p.interleave : 1;
  const int span0 = (chunk / S) * S * p.pairs_per_chunk;
  const int g0 = span0 + chunk % S;
  const int g1 = (span0 + S * p.pairs_per_chunk < n_pairs) ? span0 + S * p.pairs_per_chunk : n_pairs;
  const int64_t row0 = p.frame_off[item];
  constexpr bool analysis = ANALYSIS;            // false: ISTFT mode (p.spec_re / p.spec_im given)
  const int cut = analysis ? p.cut[item] : F;
  const T inv_n = (T)1 / (T)N;
  const SsrView<float> vs(analysis ? p.in + p.in_off[item] : p.spec_re, analysis ? n : 0);
  // ISTFT mode: the item's two half spectra as [T, F] float32 views (rows ta / tb are wave-uniform offsets)
  const SsrView<float> vre(analysis ? nullptr : p.spec_re + row0 * F, analysis ? 0 : (int64_t)n_frames * F);
  const SsrView<float> vim(analysis ? nullptr : p.spec_im + row0 * F, analysis ? 0 : (int64_t)n_frames * F);
  const SsrView<T> vw(p.window, N);
  const SsrView<cx<T>> vt(p.tw, N + SSR_W_TWP);

  SSR_REGS(Regs, regs, blk);
  SSR_WPHASE(blk, regs, {
    if (g0 < g1) {
      SSR_UNROLL for (int r = 0; r < SSR_W_P / 2; ++r) R.wl[r] = vw.at(SSR_UIDX(tid + 64 * r));
    }
  });
  BLK blk0 = blk;
  for (int g = g0; g < g1; g += S) {
    const int ta = 2 * g, tb = 2 * g + 1;
    const bool b_valid = tb < n_frames;
    blk = blk0; ssr_launder(blk);
    if constexpr (analysis) {
      // The unit's samples are requested here, not a unit ahead: 64 more live registers across the store phase do not
      // fit next to the 128 of the transform (measured: 70 spilled registers), and the second wave of the SIMD covers the
      // latency.
      SSR_WPHASE(blk, regs, {
        ssr_lowpass_wave_prefetch<T>(R, tid, vs, g, hop, n, n_frames);
        SSR_UNROLL for (int r = 0; r < SSR_W_P; ++r) {
          const T w = (r < SSR_W_P / 2) ? R.wl[r] : (T)1 - R.wl[r - SSR_W_P / 2];
          R.v[r] = {(T)R.pa[r] * w, b_valid ? (T)R.pb[r] * w : (T)0};
        }
        ssr_dft32(R.v);
      });
#define VT vt
      SSR_W_FFT_TAIL(blk, blk0, regs, L, );
#undef VT
      // zero the stop band; hand the spectrum to the inverse transform: its input register r takes swap(Z[tid + 64 r]),
      // r = b + 4 q, which is forward output register 8 b + q
      SSR_WPHASE(blk, regs, {
        cx<T> z[SSR_W_P];
        SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q = 0; q < 8; ++q) {
          const int k = tid + 64 * b + 256 * q;
          const bool zero = (k >= cut) && (k <= N - cut);
          z[b + 4 * q] = {zero ? (T)0 : R.v[8 * b + q].y, zero ? (T)0 : R.v[8 * b + q].x};
        }
        SSR_UNROLL for (int r = 0; r < SSR_W_P; ++r) R.v[r] = z[r];
        ssr_dft32(R.v);
      });
    } else {
      // ISTFT mode: Z = Xa + i Xb (Hermitian-extended), inverse-engine input = swap(Z)
      SSR_WPHASE(blk, regs, {
        SSR_UNROLL for (int r = 0; r < SSR_W_P; ++r) {
          const int k = tid + 64 * r;
          const int kk = (k <= N / 2) ? k : N - k;
          const T sgn = (k <= N / 2) ? (T)1 : (T)-1;
          const bool edge = (kk == 0) || (kk == N / 2);
          const int64_t oa = (int64_t)ta * F, ob = (int64_t)(b_valid ? tb : ta) * F;
          const T ar = (T)vre.at(SSR_UIDX(kk), oa), ai = edge ? (T)0 : sgn * (T)vim.at(SSR_UIDX(kk), oa);
          const T br = b_valid ? (T)vre.at(SSR_UIDX(kk), ob) : (T)0;
          const T bi = (b_valid && !edge) ? sgn * (T)vim.at(SSR_UIDX(kk), ob) : (T)0;
          R.v[r] = {ai + br, ar - bi};
        }
        ssr_dft32(R.v);
      });
    }
#define SSR_W_LOAD_WIN ; SSR_UNROLL for (int r = 0; r < SSR_W_P / 2; ++r) R.wl[r] = vw.at(SSR_UIDX(tid + 64 * r));
#define VT vt
    // (paired: the outputs of the last pass all stay live until frame ta is parked, so the window is requested after it)
    if constexpr (PAIRED) { SSR_W_FFT_TAIL(blk, blk0, regs, L, ); } else { SSR_W_FFT_TAIL(blk, blk0, regs, L, SSR_W_LOAD_WIN); }
#undef VT
    // registers: swap(N * IFFT): frame ta = .y, frame tb = .x at sample m = tid + 64 (b + 4 q).  Window, scale, write.
    blk = blk0; ssr_launder(blk);
    if constexpr (PAIRED) {
      const int stride = ssr_seg_stride(N, hop), shift = (2 * g * hop) & 3;           // row layout: ssr_lowpass.h
      const SsrRwView<float> vseg(p.frames + row0 * N + (int64_t)g * stride + shift, N + hop);
      // window; frame ta to the exchange array (its first hop samples stay in .y), frame tb stays in .x
      SSR_WPHASE(blk, regs, {
        SSR_SCHED_BARRIER() SSR_W_LOAD_WIN;
        SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q = 0; q < 8; ++q) {
          const int r = b + 4 * q, i = 8 * b + q;
          const T w = ((r < SSR_W_P / 2) ? R.wl[r] : (T)1 - R.wl[r - SSR_W_P / 2]) * inv_n;
          const T av = R.v[i].y * w;
          R.v[i] = {b_valid ? R.v[i].x * w : (T)0, av};
          L.re[tid + 64 * r] = av;
        }
      });
      SSR_WPHASE(blk, regs, {
        // per-lane conditions as all-ones masks (no compare: 64 lane masks in scalar registers do not fit): a lane beyond
        // the frame contributes +0.0 (its bits ANDed away), a lane beyond the head stores out of range
        SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q = 0; q < 8; ++q) {
          const int r = b + 4 * q, i = 8 * b + q, m = tid + 64 * r;
          const T tail = ssr_and_not(L.re[(m + hop) & (N - 1)], ssr_neg_mask(N - 1 - (m + hop)));
          vseg.st_raw((m + hop) * 4, (float)(R.v[i].x + tail));
          if (r < SSR_W_P / 2) vseg.st_raw((m * 4) | ssr_neg_mask(hop - 1 - m), (float)R.v[i].y);   // hop <= n_fft / 2
        }
      });
    } else {
    float* fa = p.frames + (row0 + ta) * (int64_t)N;
    float* fb = p.frames + (row0 + tb) * (int64_t)N;
    SSR_WPHASE(blk, regs, {
      SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q = 0; q < 8; ++q) {
        const int r = b + 4 * q;                                   // m = tid + 64 r
        const T w = ((r < SSR_W_P / 2) ? R.wl[r] : (T)1 - R.wl[r - SSR_W_P / 2]) * inv_n;
        fa[SSR_UIDX(tid + 64 * r)] = (float)(R.v[8 * b + q].y * w);
        if (b_valid) fb[SSR_UIDX(tid + 64 * r)] = (float)(R.v[8 * b + q].x * w);
      }
    });
    }
    // the analysis window of the next unit (the same values; requested again rather than kept)
    if constexpr (analysis) {
      SSR_WPHASE(blk, regs, SSR_UNROLL for (int r = 0; r < SSR_W_P / 2; ++r) R.wl[r] = vw.at(SSR_UIDX(tid + 64 * r)));
    }
  }
}
