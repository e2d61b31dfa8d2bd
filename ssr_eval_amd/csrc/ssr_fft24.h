// 1536-point complex transform on ONE wave, 24 points per lane: the wave engine of ssr_stft_wave.h with an in-register
// radix-24 first pass (3 x 8) instead of radix-32.  Used by the chirp-z sub-transforms of ssr_stft_rn_wave.h: a sub-sequence of
// q <= 768 samples needs M >= 2 q - 1 = 1535, and M = 1536 = 24 x 64 does a quarter less arithmetic (and moves a quarter
// fewer bytes through LDS) than M = 2048.
//
// Index algebra (N = 64 P, P = 24; lane l holds x[l + 64 r], r < P):
//   pass 0:  A_l[q]   = sum_r x[l + 64 r] W_P^(r q)                                  (in registers: ssr_dft24)
//   X[q + P m]        = sum_t W_N^(t q) W_64^(t m) A_t[q],   t = t_low + 8 t_high,  m = m0 + 8 m1
//   pass 1 (butterfly j = P t_low + q, radix 8 over t_high, twiddle W_N^(8 q t_high) = W_192^(q t_high)) -> m0
//   pass 2 (butterfly j2 = q + P m0,   radix 8 over t_low,  twiddle W_N^(t_low j2))                     -> m1
//   result: register 8 b + m1 of lane l holds Z[k], k = l + 64 b + 192 m1 = l + 64 (b + 3 m1)   (k = l mod 64, as the 2048 engine)
// A lane runs butterflies j = l + 64 b, b < 3, in both passes.
// LDS slots (one array of SSR_W24_PN values per wave; re and im parts go through it one after the other):
//   exchange 1: value (t, q) at 25 t + q          (lane stride 25: conflict-free 64-bit writes);
//               butterfly j reads input t_high at pad(j) + 200 t_high,  pad(j) = j + j / 24
//   exchange 2: pass-1 output m0 of butterfly j = (t_low, q) at q + 25 m0 + 200 t_low = pad(q + 24 m0) + 200 t_low:
//               the SAME read pattern as exchange 1 (butterfly j2 reads input t_low at pad(j2) + 200 t_low).
#pragma once
#include "ssr_stft_wave.h"

constexpr int SSR_W24_N = 1536, SSR_W24_P = 24, SSR_W24_PB = 3;
constexpr int SSR_W24_PN = 1600;                         // pad(191) + 200 * 7 + 1 = 1599
constexpr int SSR_W24_TWP = 2 * 9 * 64;                  // lane-ordered twiddle copies behind the table (= SSR_WAVE24_TWP)

// exp(-2 pi i m / 24), m = 0 .. 14 (the exponents n2 * k1 of the in-lane 3 x 8 decomposition)
template <typename T> SSR_DEV cx<T> ssr_w24(int m) {
  constexpr double c[15] = {1.0, 0.96592582628906828675, 0.86602540378443864676, 0.70710678118654752440, 0.5,
                            0.25881904510252076235, 0.0, -0.25881904510252076235, -0.5, -0.70710678118654752440,
                            -0.86602540378443864676, -0.96592582628906828675, -1.0, -0.96592582628906828675,
                            -0.86602540378443864676};
  constexpr double s[15] = {0.0, -0.25881904510252076235, -0.5, -0.70710678118654752440, -0.86602540378443864676,
                            -0.96592582628906828675, -1.0, -0.96592582628906828675, -0.86602540378443864676,
                            -0.70710678118654752440, -0.5, -0.25881904510252076235, 0.0, 0.25881904510252076235, 0.5};
  return {(T)c[m], (T)s[m]};
}

// v[r], r = 8 n1 + n2 (n1 < 3, n2 < 8)   ->   v[8 k1 + k2] = DFT24(v)[k1 + 3 k2]      (in place, registers only)
// NZ: only v[0 .. NZ) can be non-zero (a zero-padded chirp-z input: 8 < NZ <= 16, i.e. the third row group is zero and the
// second one from row NZ on); the radix-3 stage then skips the additions of zeros - the same values.
template <typename T, int NZ = 24> SSR_DEV void ssr_dft24(cx<T>* v) {
  const T h = (T)0.70710678118654752440, s3 = (T)0.86602540378443864676;
  static_assert(NZ == 24 || (NZ > 8 && NZ <= 16), "full, or two non-zero row groups");
  SSR_UNROLL for (int n2 = 0; n2 < 8; ++n2) {
    const cx<T> a = v[n2], b = v[8 + n2], c = v[16 + n2];
    cx<T> t[3];
    if (NZ == 24) {
      const cx<T> s = cadd(b, c), d = csub(b, c);
      const cx<T> m = {a.x - (T)0.5 * s.x, a.y - (T)0.5 * s.y};
      const cx<T> e = {s3 * d.y, -s3 * d.x};                    // -i sqrt(3)/2 (b - c)
      t[0] = cadd(a, s); t[1] = cadd(m, e); t[2] = csub(m, e);
    } else if (8 + n2 < NZ) {                                   // c = 0
      const cx<T> m = {a.x - (T)0.5 * b.x, a.y - (T)0.5 * b.y};
      const cx<T> e = {s3 * b.y, -s3 * b.x};
      t[0] = cadd(a, b); t[1] = cadd(m, e); t[2] = csub(m, e);
    } else {                                                    // only a
      t[0] = a; t[1] = a; t[2] = a;
    }
    SSR_UNROLL for (int k1 = 0; k1 < 3; ++k1) {
      const int m = n2 * k1;                       // compile-time after unrolling: the special angles cost no multiply
      cx<T> x = t[k1];
      if (m == 0) {
      } else if (m == 6) {
        x = cmul_negi(x);
      } else if (m == 12) {
        x = {-x.x, -x.y};
      } else if (m == 3) {
        x = {h * (x.x + x.y), h * (x.y - x.x)};
      } else if (m == 9) {
        x = {h * (x.y - x.x), -h * (x.x + x.y)};
      } else {
        x = cmul(x, ssr_w24<T>(m));
      }
      v[8 * k1 + n2] = x;
    }
  }
  SSR_UNROLL for (int k1 = 0; k1 < 3; ++k1) ssr_bfly8(v + 8 * k1);
}
// frequency held by register rho after ssr_dft24
SSR_DEV constexpr int ssr_dft24_freq(int rho) { return (rho >> 3) + 3 * (rho & 7); }

// per-lane slot bases (recomputed where they are used: three small divisions by 24)
struct SsrWave24Base { int st0; int ld8[3]; int st1[3]; };
SSR_DEV SsrWave24Base ssr_wave24_bases(int lane) {
  SsrWave24Base B;
  B.st0 = 25 * lane;
  SSR_UNROLL for (int b = 0; b < 3; ++b) {
    const int j = lane + 64 * b;
    const int tl = (j * 2731) >> 16;               // j / 24 for 0 <= j < 192
    const int q = j - 24 * tl;
    B.ld8[b] = j + tl;
    B.st1[b] = q + 200 * tl;
  }
  return B;
}

// One exchange through the wave's array (SPLIT: real parts, then imaginary parts).  WSLOT(B_, i) / RSLOT(B_, i): slot of
// register i given the lane's bases.  EXTRA: loads issued right after the first write phase (the next pass's twiddles).
#define SSR_W24_SLOT_ST0(B_, i) ((B_).st0 + ssr_dft24_freq(i))
#define SSR_W24_SLOT_LD8(B_, i) ((B_).ld8[(i) >> 3] + 200 * ((i) & 7))
#define SSR_W24_SLOT_ST1(B_, i) ((B_).st1[(i) >> 3] + 25 * ((i) & 7))
#define SSR_W24_EXCHANGE(blk, regs, L, WSLOT, RSLOT, EXTRA)                                                   \
  SSR_WPHASE(blk, regs, { const SsrWave24Base B_ = ssr_wave24_bases(tid & 63);                                 \
    SSR_UNROLL for (int i = 0; i < SSR_W24_P; ++i) (L).re[WSLOT(B_, i)] = R.v[i].x; });                       \
  SSR_WPHASE(blk, regs, { SSR_SCHED_BARRIER(); EXTRA; const SsrWave24Base B_ = ssr_wave24_bases(tid & 63);     \
    SSR_UNROLL for (int i = 0; i < SSR_W24_P; ++i) R.tx[i] = (L).re[RSLOT(B_, i)]; });                        \
  SSR_WPHASE(blk, regs, { const SsrWave24Base B_ = ssr_wave24_bases(tid & 63);                                 \
    SSR_UNROLL for (int i = 0; i < SSR_W24_P; ++i) (L).re[WSLOT(B_, i)] = R.v[i].y; });                       \
  SSR_WPHASE(blk, regs, { const SsrWave24Base B_ = ssr_wave24_bases(tid & 63);                                 \
    SSR_UNROLL for (int i = 0; i < SSR_W24_P; ++i) R.v[i] = {R.tx[i], (L).re[RSLOT(B_, i)]}; });

// (VT: a view of the plan's twiddle table INCLUDING the lane-ordered copies behind it, SSR_W24_N + SSR_W24_TWP entries)
#define SSR_W24_LOAD_TW1 { const unsigned l_ = SSR_UIDX(tid & 63); \
                           SSR_UNROLL for (int i = 0; i < 9; ++i) R.tw1[i] = VT.at(l_ + (SSR_W24_N + 64 * i)); }
#define SSR_W24_LOAD_TW2 { const unsigned l_ = SSR_UIDX(tid & 63); \
                           SSR_UNROLL for (int i = 0; i < 9; ++i) R.tw2[i] = VT.at(l_ + (SSR_W24_N + 576 + 64 * i)); }
// a radix-8 pass of the lane's three butterflies with twiddles w^(c t), t = 1..7, from the three loaded powers w^c, w^2c, w^4c
#define SSR_W24_PASS(TW)                                                                                      \
  SSR_UNROLL for (int b = 0; b < SSR_W24_PB; ++b) {                                                           \
    cx<T>* x = R.v + 8 * b;                                                                                   \
    const cx<T> w1 = R.TW[3 * b], w2 = R.TW[3 * b + 1], w4 = R.TW[3 * b + 2];                                 \
    x[1] = cmul(x[1], w1);                                                                                    \
    x[2] = cmul(x[2], w2);                                                                                    \
    x[4] = cmul(x[4], w4);                                                                                    \
    const cx<T> w3 = cmul(w1, w2);                                                                            \
    x[3] = cmul(x[3], w3);                                                                                    \
    x[5] = cmul(x[5], cmul(w1, w4));                                                                          \
    x[6] = cmul(x[6], cmul(w2, w4));                                                                          \
    x[7] = cmul(x[7], cmul(w3, w4));                                                                          \
    ssr_bfly8(x);                                                                                             \
  }
// The rest of the transform after the in-register radix-24 pass (ssr_dft24 applied to R.v).  On exit register 8 b + m1 holds
// Z[lane + 64 b + 192 m1].  EXTRA2: further table loads to issue with the last pass's twiddles.
#define SSR_W24_FFT_TAIL(blk, BLK0, regs, L, EXTRA2)                                                          \
  blk = BLK0; ssr_launder(blk);                                                                               \
  SSR_W24_EXCHANGE(blk, regs, L, SSR_W24_SLOT_ST0, SSR_W24_SLOT_LD8, SSR_W24_LOAD_TW1);                       \
  SSR_WPHASE(blk, regs, { SSR_W24_PASS(tw1) });                                                               \
  blk = BLK0; ssr_launder(blk);                                                                               \
  SSR_W24_EXCHANGE(blk, regs, L, SSR_W24_SLOT_ST1, SSR_W24_SLOT_LD8, SSR_W24_LOAD_TW2 EXTRA2);                \
  SSR_WPHASE(blk, regs, { SSR_W24_PASS(tw2) });
