// In-LDS power-of-two complex FFT for one workgroup (Stockham autosort, radix 8 with one leading
// radix-2/4 pass), forward transform exp(-2*pi*i*j*k/N).
//
// Geometry: N = 2^LOGN points, PPT (8 or 16) points per thread in registers, NT = N/PPT threads; a thread
// runs PPT/R independent radix-R butterflies per pass (PPT = 16: two radix-8 butterflies -> twice the
// instruction-level parallelism per wave and half the waves per transform; used where 8 points per thread
// would need more threads than the register file can feed, i.e. the 8192-point Bluestein transforms).
// Data lives in two LDS arrays (re[], im[], structure-of-arrays, index-padded by ssr_pad); the
// transform is in place: a pass is  load -> twiddle -> butterfly  |barrier|  store  |barrier|.
// The first pass takes its 8 points straight from registers (the caller loaded them from HBM in
// first-pass order), so a frame costs no LDS staging before the transform.
// An inverse transform is the same code with the re/im array pointers exchanged
// (IFFT(z) = swap(FFT(swap(z))) / N).
//
// Twiddles: table tw[i] = exp(-2*pi*i * i / N), i < N, computed on the host in long double and kept in
// HBM/L2 (16 B loads for double).  A radix-8 butterfly loads w^1, w^2, w^4 and forms the other four
// powers with one complex multiply each (error <= 2 ulp of the working type).
#pragma once
#include "ssr_block.h"

template <typename T> SSR_DEV void ssr_bfly2(cx<T>* v) {
  cx<T> a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}

template <typename T> SSR_DEV void ssr_bfly4(cx<T>* v) {
  cx<T> b0 = cadd(v[0], v[2]), b1 = csub(v[0], v[2]);
  cx<T> b2 = cadd(v[1], v[3]), b3 = cmul_negi(csub(v[1], v[3]));
  v[0] = cadd(b0, b2);
  v[1] = cadd(b1, b3);
  v[2] = csub(b0, b2);
  v[3] = csub(b1, b3);
}

template <typename T> SSR_DEV void ssr_bfly8(cx<T>* v) {
  const T h = (T)0.70710678118654752440;
  cx<T> u[4], d[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    u[n] = cadd(v[n], v[n + 4]);
    d[n] = csub(v[n], v[n + 4]);
  }
  // d[n] *= exp(-2*pi*i*n/8)
  d[1] = {h * (d[1].x + d[1].y), h * (d[1].y - d[1].x)};
  d[2] = cmul_negi(d[2]);
  d[3] = {h * (d[3].y - d[3].x), -h * (d[3].x + d[3].y)};
  ssr_bfly4(u);
  ssr_bfly4(d);
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    v[2 * n] = u[n];
    v[2 * n + 1] = d[n];
  }
}

template <int R, typename T> SSR_DEV void ssr_bfly(cx<T>* v) {
  if constexpr (R == 8) ssr_bfly8(v);
  else if constexpr (R == 4) ssr_bfly4(v);
  else ssr_bfly2(v);
}

// pass schedule ----------------------------------------------------------------------------------
template <int LOGN, int PPT = 8> struct SsrFftPlan {
  static constexpr int N = 1 << LOGN;
  static constexpr int NT = N / PPT;
  static constexpr int LR = (PPT >= 8) ? 3 : 2;          // log2 of the working radix: 8, or 4 at 4 points per thread
  static constexpr int RW = 1 << LR;
  static constexpr int R0 = (LOGN % LR == 0) ? RW : (1 << (LOGN % LR));
  static constexpr int NPASS = LOGN / LR + ((LOGN % LR) ? 1 : 0);
  static constexpr int radix(int p) { return p == 0 ? R0 : RW; }
  static constexpr int ns(int p) {  // product of the radices of the passes before p
    int s = 1;
    for (int i = 0; i < p; ++i) s *= radix(i);
    return s;
  }
};

// Order in which a thread's PPT registers map onto natural input indices for pass 0:
// register b*R0+q  <->  index  tid + b*NT + q*(N/R0).
template <int LOGN, int PPT = 8> SSR_DEV int ssr_fft_first_index(int tid, int reg) {
  using P = SsrFftPlan<LOGN, PPT>;
  const int b = reg / P::R0, q = reg % P::R0;
  return tid + b * P::NT + q * (P::N / P::R0);
}

template <typename T, int LOGN, int PASS, int PPT = 8>
SSR_DEV void ssr_fft_load(int tid, const T* re, const T* im, cx<T>* v) {
  using P = SsrFftPlan<LOGN, PPT>;
  constexpr int R = P::radix(PASS), NB = PPT / R;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int j = tid + b * P::NT;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int idx = ssr_pad(j + q * (P::N / R));
      v[b * R + q] = {re[idx], im[idx]};
    }
  }
}

// The three table twiddles (w^1, w^2, w^4; radix 4: w^1, w^2) of butterfly b of pass PASS: thread-constant
// addresses, so they can be requested a phase early (ssr_fft_mid_passes<..., PF = true>).
// table access through a plain pointer or through an SsrView (raw buffer resource)
template <typename T> SSR_DEV cx<T> ssr_tw_at(const cx<T>* tw, unsigned i) { return tw[i]; }
template <typename T> SSR_DEV cx<T> ssr_tw_at(const SsrView<cx<T>>& tw, unsigned i) { return tw.at(i); }

template <typename T, int LOGN, int PASS, int PPT = 8, typename TW>
SSR_DEV void ssr_fft_load_tw(int tid, const TW& tw, cx<T>* w) {
  using P = SsrFftPlan<LOGN, PPT>;
  constexpr int R = P::radix(PASS), NB = PPT / R, NS = P::ns(PASS);
  if constexpr (NS > 1) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int j = tid + b * P::NT;
      const unsigned ub = (unsigned)((j & (NS - 1)) * (P::N / (NS * R)));   // scalar table base + 32-bit lane offset
      w[3 * b] = ssr_tw_at<T>(tw, ub);
      w[3 * b + 1] = ssr_tw_at<T>(tw, 2 * ub);
      if constexpr (R == 8) w[3 * b + 2] = ssr_tw_at<T>(tw, 4 * ub);
    }
  }
}

// butterflies of pass PASS with the table twiddles already in registers
template <typename T, int LOGN, int PASS, int PPT = 8>
SSR_DEV void ssr_fft_compute_tw(cx<T>* v, const cx<T>* w) {
  using P = SsrFftPlan<LOGN, PPT>;
  constexpr int R = P::radix(PASS), NB = PPT / R, NS = P::ns(PASS);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if constexpr (NS > 1) {
      static_assert(R == P::RW || NS == 1, "only the leading pass may have a smaller radix");
      cx<T>* x = v + b * R;
      const cx<T> w1 = w[3 * b], w2 = w[3 * b + 1];
      if constexpr (R == 4) {
        x[1] = cmul(x[1], w1);
        x[2] = cmul(x[2], w2);
        x[3] = cmul(x[3], cmul(w1, w2));
      } else {
        const cx<T> w4 = w[3 * b + 2];
        // twiddle powers are formed just before use to keep few of them live (register pressure)
        x[1] = cmul(x[1], w1);
        x[2] = cmul(x[2], w2);
        x[4] = cmul(x[4], w4);
        const cx<T> w3 = cmul(w1, w2);
        x[3] = cmul(x[3], w3);
        x[5] = cmul(x[5], cmul(w1, w4));
        x[6] = cmul(x[6], cmul(w2, w4));
        x[7] = cmul(x[7], cmul(w3, w4));
      }
    }
    ssr_bfly<R>(v + b * R);
  }
}

template <typename T, int LOGN, int PASS, int PPT = 8, typename TW>
SSR_DEV void ssr_fft_compute(int tid, cx<T>* v, const TW& tw) {
  using P = SsrFftPlan<LOGN, PPT>;
  cx<T> w[3 * (PPT / P::radix(PASS))];
  ssr_fft_load_tw<T, LOGN, PASS, PPT>(tid, tw, w);
  ssr_fft_compute_tw<T, LOGN, PASS, PPT>(v, w);
}

// natural output index of register `reg` after pass PASS
template <int LOGN, int PASS, int PPT = 8> SSR_DEV int ssr_fft_out_index(int tid, int reg) {
  using P = SsrFftPlan<LOGN, PPT>;
  constexpr int R = P::radix(PASS), NS = P::ns(PASS);
  const int b = reg / R, q = reg % R;
  const int j = tid + b * P::NT;
  const int k = j & (NS - 1);
  return (j - k) * R + k + q * NS;
}

template <typename T, int LOGN, int PASS, int PPT = 8>
SSR_DEV void ssr_fft_store(int tid, T* re, T* im, const cx<T>* v) {
#pragma unroll
  for (int r = 0; r < PPT; ++r) {
    const int idx = ssr_pad(ssr_fft_out_index<LOGN, PASS, PPT>(tid, r));
    re[idx] = v[r].x;
    im[idx] = v[r].y;
  }
}

// Passes PASS..NPASS-2 complete (load, compute, store); the LAST pass is left loaded+computed in
// registers so the caller can fuse its own epilogue into the final store
// (ssr_fft_out_index<LOGN, NPASS-1> gives each register's natural frequency index).
// Pre-condition: pass 0 results already stored to (re, im) and a barrier passed.
// Regs must expose `cx<T> v[PPT]`.
// PF = true: the table twiddles of a pass are requested one phase early - in the store phase of the previous pass,
// whose data registers are dead by then - so their L1/L2 latency overlaps the barrier and the LDS reads.  Regs must
// then expose `cx<T> twp[3 * PPT / 8]`, and the CALLER has requested pass PASS's twiddles (ssr_fft_load_tw into
// R.twp) in the phase that stored pass PASS - 1.
template <typename T, int LOGN, int PASS, int PPT, bool PF = false, typename BLK, typename REGS, typename TW>
SSR_BODY void ssr_fft_mid_passes(BLK& blk, REGS& regs, T* re, T* im, const TW& tw) {
  using P = SsrFftPlan<LOGN, PPT>;
  if constexpr (PF) {
    SSR_PHASE(blk, regs, ssr_fft_load<T, LOGN, PASS, PPT>(tid, re, im, R.v);
              ssr_fft_compute_tw<T, LOGN, PASS, PPT>(R.v, R.twp));
    if constexpr (PASS < P::NPASS - 1) {
      SSR_PHASE(blk, regs, ssr_fft_store<T, LOGN, PASS, PPT>(tid, re, im, R.v);
                ssr_fft_load_tw<T, LOGN, PASS + 1, PPT>(tid, tw, R.twp));
      ssr_fft_mid_passes<T, LOGN, PASS + 1, PPT, true>(blk, regs, re, im, tw);
    }
  } else if constexpr (PASS < P::NPASS - 1) {
    SSR_PHASE(blk, regs, ssr_fft_load<T, LOGN, PASS, PPT>(tid, re, im, R.v);
              ssr_fft_compute<T, LOGN, PASS, PPT>(tid, R.v, tw));
    SSR_PHASE(blk, regs, ssr_fft_store<T, LOGN, PASS, PPT>(tid, re, im, R.v));
    ssr_fft_mid_passes<T, LOGN, PASS + 1, PPT>(blk, regs, re, im, tw);
  } else {
    SSR_PHASE(blk, regs, ssr_fft_load<T, LOGN, PASS, PPT>(tid, re, im, R.v);
              ssr_fft_compute<T, LOGN, PASS, PPT>(tid, R.v, tw));
  }
}
