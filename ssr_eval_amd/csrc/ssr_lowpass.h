// Kernel bodies K6: STFT-domain hard low-pass and inverse STFT.
//
// Reference semantics: ssr_eval/lowpass.py:17-28 (stft_hard_lowpass_v0) on FDomainHelper(2048, 441)
// (ssr_eval/dsp.py:76-119 -> torchlibrosa STFT / ISTFT): centred reflect-padded frames, periodic Hann
// analysis window, bins >= cut set to zero, inverse DFT fused with the Hann synthesis window,
// overlap-add at stride hop, division by the overlap-added squared window clamped to 1e-11, samples
// [n_fft/2, n_fft/2 + length).  mag*cos / mag*sin of the reference reassemble (re, im) up to float32
// rounding, so the zeroing is applied to the complex spectrum directly.
//
// Stage 1 (this body): per pair of frames (2g, 2g+1) packed as re/im of ONE complex transform:
//     forward FFT -> zero cut <= k <= n_fft - cut -> inverse FFT -> * window / n_fft -> frames[t][m]
//   or, in ISTFT mode, Hermitian-extend two given half spectra, pack, inverse FFT -> frames.
// Stage 2 (ssr_ola_sample): deterministic gather overlap-add + normalisation, one thread per sample.
#pragma once
#include "ssr_stft.h"

template <typename T> struct SsrLowpassParams {
  const float* in;           // signals (analysis mode)
  const int64_t* in_off;     // [n_items]
  const int32_t* len;        // [n_items] samples (defines the number of frames)
  const int32_t* cut;        // [n_items] first zeroed bin (analysis mode)
  const int64_t* frame_off;  // [n_items] first row of item i in `frames` / spec_re / spec_im
  int n_fft, hop, pairs_per_chunk, n_chunks;
  int interleave;            // wave engine: S chunks of a group take every S-th frame pair of the group's span (0 / 1: consecutive
                             // pairs per chunk); n_chunks is then a multiple of S - see ssr_stft_wave.h
  const T* window;
  const cx<T>* tw;
  const float* spec_re;      // ISTFT mode: [rows, F] real parts (null in analysis mode)
  const float* spec_im;
  float* frames;             // [rows, n_fft] windowed inverse frames; PAIRED (ssr_lowpass_wave.h): per item ceil(T / 2)
                             // segments of n_fft + hop samples in the same region (frames 2g and 2g+1 already added up)
};

// grid = (n_chunks, n_items), block = n_fft / 8
template <typename T, int LOGN, typename BLK>
SSR_BODY void ssr_lowpass_frames_body(const SsrLowpassParams<T>& p, BLK& blk, int chunk, int item, char* lds_base) {
  using P = SsrFftPlan<LOGN>;
  constexpr int N = P::N, LAST = P::NPASS - 1, F = N / 2 + 1;
  using Regs = SsrStftRegs<T, false>;
  SsrStftLds<T, LOGN> L(lds_base);
  const int n = p.len[item], hop = p.hop;
  if (n <= N / 2) return;     // precondition (include/ssr_hip.h): len > n_fft / 2; such an item is skipped (k_ola* zero its output)
  const int n_frames = ssr_num_frames_dev(n, N, hop);
  const int n_pairs = (n_frames + 1) / 2;
  const int g0 = chunk * p.pairs_per_chunk;
  const int g1 = (g0 + p.pairs_per_chunk < n_pairs) ? g0 + p.pairs_per_chunk : n_pairs;
  const int64_t row0 = p.frame_off[item];
  const bool analysis = (p.spec_re == nullptr);
  const float* sig = analysis ? p.in + p.in_off[item] : nullptr;
  const int cut = analysis ? p.cut[item] : F;
  const T inv_n = (T)1 / (T)N;
  // signal and plan tables as the loads see them (raw buffer resources on the device)
  const SsrView<float> vs(analysis ? sig : p.spec_re, analysis ? n : 0);
  const SsrView<T> vw(p.window, N);
  const SsrView<cx<T>> vt(p.tw, N);

  SSR_REGS(Regs, regs, blk);
  for (int g = g0; g < g1; ++g) {
    const int ta = 2 * g, tb = 2 * g + 1;
    const bool b_valid = tb < n_frames;
    if (analysis) {
      SSR_PHASE(blk, regs, {
        float fa[8], fb[8];
        T w[8];
        SSR_UNROLL for (int r = 0; r < 8; ++r) {       // all loads first (branch-free addresses), then the arithmetic
          const int m = ssr_fft_first_index<LOGN>(tid, r);
          const int tb_c = (tb < n_frames) ? tb : n_frames - 1;          // a missing frame re-reads the last one
          fa[r] = vs.at(SSR_UIDX(ssr_reflect(ta * hop + m - N / 2, n)));
          fb[r] = vs.at(SSR_UIDX(ssr_reflect(tb_c * hop + m - N / 2, n)));
          w[r] = vw.at(SSR_UIDX(m));
        }
        SSR_UNROLL for (int r = 0; r < 8; ++r) R.v[r] = {(T)fa[r] * w[r], b_valid ? (T)fb[r] * w[r] : (T)0};
        ssr_fft_compute<T, LOGN, 0>(tid, R.v, vt);
        ssr_fft_store<T, LOGN, 0>(tid, L.re, L.im, R.v);
        ssr_fft_load_tw<T, LOGN, 1, 8>(tid, vt, R.twp);                 // pass 1's twiddles, in flight across the barrier
      });
      ssr_fft_mid_passes<T, LOGN, 1, 8, true>(blk, regs, L.re, L.im, vt);
      SSR_PHASE(blk, regs, {
        SSR_UNROLL for (int r = 0; r < 8; ++r) {
          const int k = ssr_fft_out_index<LOGN, LAST>(tid, r);
          const bool zero = (k >= cut) && (k <= N - cut);
          L.re[ssr_pad(k)] = zero ? (T)0 : R.v[r].x;
          L.im[ssr_pad(k)] = zero ? (T)0 : R.v[r].y;
        }
      });
      // inverse = forward engine on exchanged arrays
      SSR_PHASE(blk, regs, ssr_fft_load<T, LOGN, 0>(tid, L.im, L.re, R.v);
                ssr_fft_compute<T, LOGN, 0>(tid, R.v, vt));
    } else {
      // ISTFT mode: pack Z = Xa + i*Xb (Hermitian-extended) straight into inverse pass-0 registers
      SSR_PHASE(blk, regs, {
        SSR_UNROLL for (int r = 0; r < 8; ++r) {
          const int k = ssr_fft_first_index<LOGN>(tid, r);
          const int kk = (k <= N / 2) ? k : N - k;
          const T sgn = (k <= N / 2) ? (T)1 : (T)-1;
          const bool edge = (kk == 0) || (kk == N / 2);
          const int64_t ia = (row0 + ta) * F + kk, ib = (row0 + tb) * F + kk;
          const T ar = (T)p.spec_re[ia], ai = edge ? (T)0 : sgn * (T)p.spec_im[ia];
          const T br = b_valid ? (T)p.spec_re[ib] : (T)0;
          const T bi = (b_valid && !edge) ? sgn * (T)p.spec_im[ib] : (T)0;
          // Z = (ar - bi) + i (ai + br); inverse engine input is swap(Z)
          R.v[r] = {ai + br, ar - bi};
        }
        ssr_fft_compute<T, LOGN, 0>(tid, R.v, vt);
      });
    }
    SSR_PHASE(blk, regs, ssr_fft_store<T, LOGN, 0>(tid, L.im, L.re, R.v);
              ssr_fft_load_tw<T, LOGN, 1, 8>(tid, vt, R.twp));
    ssr_fft_mid_passes<T, LOGN, 1, 8, true>(blk, regs, L.im, L.re, vt);
    // registers: swap(N * IFFT): frame ta = .y, frame tb = .x.  Window, scale, write (coalesced).
    SSR_PHASE(blk, regs, {
      SSR_UNROLL for (int r = 0; r < 8; ++r) {
        const int m = ssr_fft_out_index<LOGN, LAST>(tid, r);
        const T w = vw.at(SSR_UIDX(m)) * inv_n;
        p.frames[(row0 + ta) * N + m] = (float)(R.v[r].y * w);
        if (b_valid) p.frames[(row0 + tb) * N + m] = (float)(R.v[r].x * w);
      }
    });
  }
}

struct SsrOlaParams {
  const float* frames;       // [rows, n_fft], or paired segments (see SsrLowpassParams)
  const int64_t* frame_off;  // [n_items]
  const int32_t* len;        // [n_items] output samples
  const int64_t* out_off;    // [n_items]
  int n_fft, hop;
  const double* window;      // [n_fft]
  float* out;
  const double* wss_tab;     // [hop] overlap-added squared window where every overlapping frame exists (paired kernel)
  float inv_hop, inv_2hop;   // 1 / hop, 1 / (2 hop) rounded to float32 (quotient estimates, corrected exactly)
};

// Overlap-added squared window at padded position `pos` (= sample + n_fft/2): the frames t with 0 <= pos - t hop < n_fft
// that exist, added in ascending t; clamped like torchlibrosa's ISTFT (1e-11).
SSR_DEV double ssr_ola_wss(const double* window, int N, int hop, int n_frames, int pos) {
  int t_hi = pos / hop;
  if (t_hi > n_frames - 1) t_hi = n_frames - 1;
  const int t_lo = (pos < N) ? 0 : (pos - N) / hop + 1;
  double wss = 0.0;
  for (int t = t_lo; t <= t_hi; ++t) {
    const int m = pos - t * hop;
    wss += window[m] * window[m];
  }
  return wss < 1e-11 ? 1e-11 : wss;
}

SSR_DEV void ssr_ola_sample(const SsrOlaParams& p, int item, int s) {
  const int n = p.len[item];
  if (s >= n) return;
  const int N = p.n_fft, hop = p.hop;
  if (n <= N / 2) { p.out[p.out_off[item] + s] = 0.0f; return; }   // item skipped by the frame kernel (reflect padding undefined)
  const int n_frames = ssr_num_frames_dev(n, N, hop);
  const int pos = s + N / 2;
  int t_hi = pos / hop;
  if (t_hi > n_frames - 1) t_hi = n_frames - 1;
  const int t_lo = (pos < N) ? 0 : (pos - N) / hop + 1;
  const float* fr = p.frames + p.frame_off[item] * (int64_t)N;
  double acc = 0.0;
  for (int t = t_lo; t <= t_hi; ++t) acc += (double)fr[(int64_t)t * N + (pos - t * hop)];
  p.out[p.out_off[item] + s] = (float)(acc / ssr_ola_wss(p.window, N, hop, n_frames, pos));
}

// floor(a / d) for 0 <= a < 2^24 from the float32 reciprocal of d (one multiply + an exact correction instead of the
// ~40-instruction integer division)
SSR_DEV int ssr_fast_div(int a, int d, float inv_d) {
  int q = (int)((float)a * inv_d);
  if ((q + 1) * d <= a) ++q;
  if (q * d > a) --q;
  return q;
}

// PAIRED segments: unit u = frames 2u, 2u+1 starts at padded position 2 u hop and is n_fft + hop long.  Segment rows are
// ssr_seg_stride floats apart and row u is shifted right by (2 u hop) mod 4 elements, so that the element of padded
// position pos sits at u stride + pos - ((2 u hop) & ~3): 16-byte aligned whenever pos is a multiple of 4 (the quad loads of
// ssr_ola_paired_quad; a dwordx4 access that is only 4-byte aligned costs several times an aligned one).
SSR_HD int ssr_seg_stride(int n_fft, int hop) { return (n_fft + hop + ((hop & 1) ? 2 : 0) + 3) & ~3; }

SSR_DEV void ssr_ola_paired_sample(const SsrOlaParams& p, int item, int s) {
  const int n = p.len[item];
  if (s >= n) return;
  const int N = p.n_fft, hop = p.hop, SEG = N + hop, two = 2 * hop, stride = ssr_seg_stride(N, hop);
  if (n <= N / 2) { p.out[p.out_off[item] + s] = 0.0f; return; }   // item skipped by the frame kernel: its rows hold no segment
  const int n_frames = ssr_num_frames_dev(n, N, hop);
  const int n_units = (n_frames + 1) / 2;
  const int pos = s + N / 2;
  int u_hi = pos / two;
  if (u_hi > n_units - 1) u_hi = n_units - 1;
  const int u_lo = (pos < SEG) ? 0 : (pos - SEG) / two + 1;
  const float* sg = p.frames + p.frame_off[item] * (int64_t)N;
  double acc = 0.0;
  for (int u = u_lo; u <= u_hi; ++u) acc += (double)sg[(int64_t)u * stride + (pos - ((u * two) & ~3))];
  p.out[p.out_off[item] + s] = (float)(acc / ssr_ola_wss(p.window, N, hop, n_frames, pos));
}

// Four consecutive samples s0 .. s0+3 (s0 a multiple of 4) per thread.  Where every frame that overlaps them exists (all but
// the first and last few frames of a signal) the quotients are formed once (float32 reciprocal estimate, corrected
// exactly), every unit that reaches ANY of the four contributes one aligned 16-byte load - elements outside the unit are
// replaced by 0.0, which leaves the sample's sum bit-identical - and the window sums come from the table; otherwise sample
// by sample as above.  No lane of a wave leaves the fast path because a unit or a frame starts inside its quad (with hop
// 441 that would be most waves).
SSR_DEV void ssr_ola_paired_quad(const SsrOlaParams& p, int item, int s0) {
  const int n = p.len[item];
  if (s0 >= n) return;
  const int N = p.n_fft, hop = p.hop, SEG = N + hop, two = 2 * hop, stride = ssr_seg_stride(N, hop);
  const int pos0 = s0 + N / 2, pos3 = pos0 + 3;
  if (s0 + 3 < n && pos3 < (1 << 24) && hop >= 4) {
    const int n_frames = ssr_num_frames_dev(n, N, hop);
    const int n_units = (n_frames + 1) / 2;
    const int t0 = ssr_fast_div(pos0, hop, p.inv_hop);
    if (t0 >= (N - 1) / hop && t0 + 1 <= n_frames - 1) {
      int u_hi = ssr_fast_div(pos3, two, p.inv_2hop);
      if (u_hi > n_units - 1) u_hi = n_units - 1;
      const int u_lo = (pos0 < SEG) ? 0 : ssr_fast_div(pos0 - SEG, two, p.inv_2hop) + 1;
      const float* sg = p.frames + p.frame_off[item] * (int64_t)N;
      double a[4] = {0.0, 0.0, 0.0, 0.0};
      for (int u = u_lo; u <= u_hi; ++u) {
        float v[4];
        memcpy(v, sg + (int64_t)u * stride + (pos0 - ((u * two) & ~3)), 16);
        const int j0 = pos0 - u * two;                             // index of the quad's first sample in unit u
        for (int k = 0; k < 4; ++k) a[k] += ((unsigned)(j0 + k) < (unsigned)SEG) ? (double)v[k] : 0.0;
      }
      const int m0 = pos0 - t0 * hop;
      float o[4];
      for (int k = 0; k < 4; ++k) o[k] = (float)(a[k] / p.wss_tab[(m0 + k < hop) ? m0 + k : m0 + k - hop]);
      memcpy(p.out + p.out_off[item] + s0, o, 16);
      return;
    }
  }
  for (int k = 0; k < 4; ++k) ssr_ola_paired_sample(p, item, s0 + k);
}
