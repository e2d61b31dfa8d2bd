// Kernel body K1 (variant): STFT for n_fft = 3 q through one radix-3 decimation-in-time step over three
// length-q Bluestein transforms (q = 743, M = 2048 for n_fft = 2229, the AudioMetrics(48000) size).
//
//   X[k + q j] = sum_{r<3}  W3^{r j} * ( W_n^{r k} * DFT_q{ x[3 m + r] }[k] ),   k < q, j < 3,  n = 3 q.
//
// Per unit (two real frames packed as one complex sequence, as in ssr_stft.h) the workgroup runs, for
// r = 0, 1, 2:  decimated frame * (window * chirp)  ->  FFT_M  ->  * filter  ->  IFFT_M  ->  * (chirp * W_n^{rk} / 2)
// and parks the q results in LDS; the epilogue forms X[K] and X[n - K] with the 3-point butterfly, separates the
// two spectra and feeds the same emit / accumulate code as the other engines.
// LDS: the 2^LOGN FFT arrays of ssr_stft.h + 3 q complex values (71 KB for q = 743, float64) -> 2 workgroups / CU.
#pragma once
#include "ssr_stft.h"

template <typename T, int LOGN> struct SsrStftR3Lds {
  using Base = SsrStftLds<T, LOGN, 8>;
  static size_t bytes(int q) { return Base::bytes() + sizeof(T) * 2 * 3 * (size_t)q; }
};

// X[K] (already carrying the factor 1/2) from the three parked sub-spectra
template <typename T> SSR_DEV cx<T> ssr_r3_combine(const T* yre, const T* yim, int q, int K) {
  const T c = (T)-0.5, s = (T)0.86602540378443864676;   // W3 = exp(-2 pi i / 3) = c - i s
  const int j = K / q, k = K - j * q;
  const cx<T> y0 = {yre[k], yim[k]}, y1 = {yre[q + k], yim[q + k]}, y2 = {yre[2 * q + k], yim[2 * q + k]};
  if (j == 0) return {y0.x + y1.x + y2.x, y0.y + y1.y + y2.y};
  // j = 1: y0 + W3 y1 + W3^2 y2 ;  j = 2: y0 + W3^2 y1 + W3 y2   (W3^2 = conj(W3))
  const cx<T> a = (j == 1) ? y1 : y2, b = (j == 1) ? y2 : y1;     // a * W3 + b * conj(W3)
  const T sr = a.x + b.x, si = a.y + b.y;                          // real-coefficient part (c)
  const T dr = a.x - b.x, di = a.y - b.y;                          // (-i s) * (a - b)
  return {y0.x + c * sr + s * di, y0.y + c * si - s * dr};
}

// grid = (n_chunks, n_items); block = 2^LOGN / 8 threads.
template <typename T, int LOGN, int MODE, bool SUMS, int IN64, typename BLK>
SSR_BODY void ssr_stft_r3_body(const SsrStftParams<T>& p, BLK& blk, int chunk, int item, char* lds_base) {
  constexpr int PPT = 8;
  using P = SsrFftPlan<LOGN, PPT>;
  constexpr int NT = P::NT, LAST = P::NPASS - 1, NW = (NT + 63) / 64;
  using Regs = SsrStftRegs<T, SUMS, PPT>;
  SsrStftLds<T, LOGN, PPT> L(lds_base);
  const int n_fft = p.n_fft, hop = p.hop, F = n_fft / 2 + 1, q = n_fft / 3;
  T* yre = reinterpret_cast<T*>(lds_base + SsrStftLds<T, LOGN, PPT>::bytes());
  T* yim = yre + 3 * q;

  const int n = p.len[item];
  const int n_frames = ssr_num_frames_dev(n, n_fft, hop);
  const int n_units = (MODE == SSR_MODE_PAIR) ? n_frames : (n_frames + 1) / 2;
  const int u0 = chunk * p.units_per_chunk;
  const int u1 = (u0 + p.units_per_chunk < n_units) ? u0 + p.units_per_chunk : n_units;
  static_assert(IN64 == 0 || MODE == SSR_MODE_PAIR, "float64 signals exist on the pair path only");
  using SA = typename SsrSample<(IN64 & 1) != 0>::type;
  using SB = typename SsrSample<(IN64 & 2) != 0>::type;
  const SA* sa;
  if constexpr (IN64 & 1) sa = p.a64 + p.a_off[item];
  else sa = p.a + p.a_off[item];
  const SB* sb;
  if constexpr (IN64 & 2) sb = p.b64 + p.b_off[item];
  else sb = (MODE == SSR_MODE_PAIR) ? p.b + p.b_off[item] : p.a + p.a_off[item];
  const int64_t row0 = p.frame_off[item];
  double* part = p.part ? p.part + ((int64_t)item * p.n_chunks + chunk) * SSR_NPART : nullptr;
  const bool want_lsd = (MODE == SSR_MODE_PAIR) && (p.metric_mask & SSR_M_LSD);
  const int pad = n_fft / 2;
  // the item's signals and the plan tables as the loads see them (raw buffer resources on the device)
  const SsrView<SA> va(sa, n);
  const SsrView<SB> vb(sb, n);
  const SsrView<cx<T>> vwc(p.wchirp, n_fft), vbf(p.bfilt, (int64_t)1 << LOGN), vch(p.chirp, n_fft), vt(p.tw, (int64_t)1 << LOGN);

  SSR_REGS(Regs, regs, blk);
  SSR_PHASE(blk, regs, {
    for (int i = tid; i < 6 * 16; i += NT) L.wacc[i] = 0.0;   // NT may be as small as 32
    if (tid == 0) L.res[0] = 0.0;
    for (int i = 0; i < (SUMS ? 6 : 1); ++i) R.sums[i] = 0.0;
  });

  for (int u = u0; u < u1; ++u) {
    const int ta = (MODE == SSR_MODE_PAIR) ? u : 2 * u;
    const int tb = (MODE == SSR_MODE_PAIR) ? u : 2 * u + 1;
    const bool a_ok = ta < n_frames, b_ok = tb < n_frames;
    const int ta_c = a_ok ? ta : n_frames - 1, tb_c = b_ok ? tb : n_frames - 1;
    const int base_a = ta_c * hop - pad, base_b = tb_c * hop - pad;
    const bool interior = base_a >= 0 && base_b >= 0 && base_a + n_fft <= n && base_b + n_fft <= n;

    for (int r = 0; r < 3; ++r) {
      const int64_t wch_off = (int64_t)r * q, post_off = (int64_t)r * q;      // round r's slice of the two chirp tables
      // ---- decimated frame (samples 3m + r) -> registers, pre-multiply, pass 0, store.
      // Only indices m < q are non-zero; q <= M/2, so at least the upper half of the registers is skipped.
      SSR_PHASE(blk, regs, {
        bool nza = false, nzb = false;
        SSR_UNROLL for (int g = 0; g < PPT; ++g) {
          const int m_min = ssr_fft_first_index<LOGN, PPT>(0, g);          // block-uniform: smallest index of this register
          if (m_min < q) {
            const int m = ssr_fft_first_index<LOGN, PPT>(tid, g);
            const int mc = (m < q) ? m : q - 1;
            const int s3 = 3 * mc + r;
            const int ia = interior ? base_a + s3 : ssr_reflect(base_a + s3, n);
            const int ib = interior ? base_b + s3 : ssr_reflect(base_b + s3, n);
            const SA fa = va.at(SSR_UIDX(ia));
            const SB fb = vb.at(SSR_UIDX(ib));
            const cx<T> z = cmul(cx<T>{a_ok ? (T)fa : (T)0, b_ok ? (T)fb : (T)0}, vwc.at(SSR_UIDX(mc), wch_off));
            R.v[g] = (m < q) ? z : cx<T>{(T)0, (T)0};
            nza = nza || (m < q && s3 != 0 && fa != 0);             // frame sample 0: window weight exactly 0 (periodic Hann)
            nzb = nzb || (m < q && s3 != 0 && fb != 0);
          } else {
            R.v[g] = cx<T>{(T)0, (T)0};
          }
        }
        // non-zero flags of the frame: one slot per decimation round (three rounds make a frame)
        SSR_WAVE_ANY_STORE(tid, nza, L.nz + r * 4);
        SSR_WAVE_ANY_STORE(tid, nzb, L.nz + 16 + r * 4);
        ssr_fft_compute<T, LOGN, 0, PPT>(tid, R.v, vt);
        ssr_fft_store<T, LOGN, 0, PPT>(tid, L.re, L.im, R.v);
        ssr_fft_load_tw<T, LOGN, 1, PPT>(tid, vt, R.twp);          // pass 1's twiddles, in flight across the barrier
        if (r == 0 && want_lsd && u > u0 && tid == 0) {               // close the previous unit's LSD
          double s = 0.0;
          for (int w = 0; w < NW; ++w) s += L.sc1[w];
          L.res[0] += sqrt(s / (double)F);
        }
      });
      ssr_fft_mid_passes<T, LOGN, 1, PPT, true>(blk, regs, L.re, L.im, vt);
      SSR_PHASE(blk, regs, {
        SSR_UNROLL for (int g = 0; g < PPT; ++g) {
          const int k = ssr_fft_out_index<LOGN, LAST, PPT>(tid, g);
          const cx<T> y = cmul(R.v[g], vbf.at(SSR_UIDX(k)));
          L.re[ssr_pad(k)] = y.x;
          L.im[ssr_pad(k)] = y.y;
        }
      });
      SSR_PHASE(blk, regs, ssr_fft_load<T, LOGN, 0, PPT>(tid, L.im, L.re, R.v);
                ssr_fft_compute<T, LOGN, 0, PPT>(tid, R.v, vt));
      SSR_PHASE(blk, regs, ssr_fft_store<T, LOGN, 0, PPT>(tid, L.im, L.re, R.v);
                ssr_fft_load_tw<T, LOGN, 1, PPT>(tid, vt, R.twp));
      ssr_fft_mid_passes<T, LOGN, 1, PPT, true>(blk, regs, L.im, L.re, vt);
      // registers hold swap(IFFT * M): true real part = .y, true imaginary part = .x.  Park Y_r.
      // (yre / yim are only read by the epilogue, after the barrier that ends the r = 2 round.)
      SSR_PHASE(blk, regs, {
        SSR_UNROLL for (int g = 0; g < PPT; ++g) {
          const int k = ssr_fft_out_index<LOGN, LAST, PPT>(tid, g);
          if (k < q) {
            const cx<T> y = cmul(cx<T>{R.v[g].y, R.v[g].x}, vch.at(SSR_UIDX(k), post_off));
            yre[r * q + k] = y.x;
            yim[r * q + k] = y.y;
          }
        }
      });
    }

    const int64_t OP = p.out_pitch ? p.out_pitch : F;      // floats between output rows
    float* ra0 = p.out_a ? p.out_a + (row0 + ta) * OP : nullptr;
    float* ra1 = p.out_a ? p.out_a + (row0 + tb) * OP : nullptr;
    float* rb0 = p.out_b ? p.out_b + (row0 + ta) * OP : nullptr;
    float* rb1 = p.out_b ? p.out_b + (row0 + tb) * OP : nullptr;
    if (MODE == SSR_MODE_PAIR) { ra1 = ra0; rb1 = rb0; }
    SSR_PHASE(blk, regs, {
      double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      bool a_nz = false, b_nz = false;
      for (int i = 0; i < 3 * 4; ++i) {
        if ((i & 3) < NW) { a_nz = a_nz || L.nz[i] != 0; b_nz = b_nz || L.nz[16 + i] != 0; }
      }
      for (int K = tid; K < F; K += NT) {
        const int Kn = (K == 0) ? 0 : n_fft - K;
        const cx<T> zk = ssr_r3_combine<T>(yre, yim, q, K);
        const cx<T> zn = ssr_r3_combine<T>(yre, yim, q, Kn);
        ssr_emit_bin<T, MODE, IN64>(p, acc, (unsigned)K, zk, zn, ra0, ra1, rb0, rb1, b_ok, a_nz, b_nz);
      }
      if (want_lsd) SSR_WAVE_SUM_STORE(tid, NT, acc[0], L.sc1);
      if constexpr (SUMS)
        for (int i = 0; i < 6; ++i) R.sums[i] += acc[1 + i];
    });
  }

  if (part == nullptr) return;
  if constexpr (SUMS) {
    SSR_PHASE(blk, regs, for (int i = 0; i < 6; ++i) SSR_WAVE_SUM_ADD(tid, NT, R.sums[i], L.wacc + i * 16));
  }
  SSR_PHASE(blk, regs, if (tid == 0) {
    double lsd = L.res[0];
    if (want_lsd && u1 > u0) {
      double s = 0.0;
      for (int w = 0; w < NW; ++w) s += L.sc1[w];
      lsd += sqrt(s / (double)F);
    }
    part[0] = lsd;
    for (int i = 0; i < 6; ++i) {
      double s = 0.0;
      for (int w = 0; w < NW; ++w) s += L.wacc[i * 16 + w];
      part[1 + i] = s;
    }
    part[7] = 0.0;
  });
}
