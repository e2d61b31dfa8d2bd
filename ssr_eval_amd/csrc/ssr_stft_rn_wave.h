// Kernel body K1-K4 for n_fft = R q (R = 1, 2, 3; q <= 1024) on the wave engine: the AudioMetrics(rate) sizes that are not a
// power of two - 2229 = 3 * 743 (48 kHz: what every user of the reference's default API hits), 1486 = 2 * 743 (32 kHz),
// 1114 = 2 * 557 (24 kHz), 743 (16 kHz) - ssr_eval/metrics.py:16-19.
//
// One radix-R decimation-in-time step over R length-q Bluestein transforms with M = 2048,
//     X[k + q j] = sum_{r<R} W_R^{r j} * ( W_n^{r k} * DFT_q{ x[R m + r] }[k] )        (R = 1: the plain chirp-z transform).
// The block engines walk the sub-sequences in lock step on four waves (ssr_stft_r3.h: six barrier-phased 2048-point
// transforms per unit on a 35 KB buffer plus 36 KB of parked sub-spectra; R = 2 and 1 ran as ONE Bluestein transform of 4096 /
// 2048 points, ssr_stft.h).  Here a workgroup is R AUTONOMOUS WAVES, wave r running the whole chirp-z of sub-sequence r
// (forward transform -> * filter -> inverse transform, the inverse starting from the registers the forward one ended in, as in
// ssr_lowpass_wave.h) on its own 17 KB exchange array with no barrier; the arrays are free once the transforms are done, so
// the sub-spectra (2 R q values) are parked IN them for the epilogue.  Three workgroup barriers per unit (transforms done /
// parked / epilogue done), 17 KB per wave -> eight or nine waves per CU whatever R.
#pragma once
#include "ssr_stft_r3.h"
#include "ssr_stft_wave.h"
#include "ssr_fft24.h"

// NQ: rows of 64 sub-sequence samples / sub-spectrum bins a wave handles, in fours: 3 for q <= 768 (every AudioMetrics
// size: 743, 557), 4 for q <= 1024.  Twelve rows instead of sixteen is eight fewer prefetch registers and eight fewer
// conditions per phase (the sixteen-row variants spill 18-40 VGPRs).
// P: points per lane of the chirp-z transforms: 32 (M = 2048, ssr_stft_wave.h) or 24 (M = 1536, ssr_fft24.h; q <= 768)
// SA: sample type of the first signal (double for a float64 estimate - ssr_stft_r3_rot.h's IN64 variant)
template <typename T, bool SUMS, int NQ, int P = 32, typename SA = float, typename SB = float> struct SsrRnWaveRegs {
  cx<T> v[P];
  T tx[P];
  SA pa[4 * NQ];                  // the next unit's decimated samples m = lane + 64 i, requested a unit ahead
  SB pb[4 * NQ];
  cx<T> tw1[P == 32 ? 7 : 9];
  cx<T> tw2[P == 32 ? 12 : 9];
};
template <int P> SSR_HD constexpr int ssr_rn_wave_pn() { return P == 32 ? SSR_W_PN : SSR_W24_PN; }

template <typename T, int NW, int P = 32> struct SsrRnWaveLds {
  // scratch (doubles first), then NW split-exchange arrays; the parked sub-spectra alias the arrays
  static constexpr size_t bytes() { return sizeof(double) * (4 + 6 * 4 + 2) + sizeof(int) * 16 + sizeof(T) * NW * ssr_rn_wave_pn<P>(); }
  double* sc1; double* wacc; double* res; int* nz; T* x;
  SSR_MEMBER explicit SsrRnWaveLds(char* base) {
    sc1 = reinterpret_cast<double*>(base);          // [3] per-wave LSD sums of the current unit (+1 pad)
    wacc = sc1 + 4;                                 // [6][4] per-wave SISpec sums at the chunk end
    res = wacc + 6 * 4;                             // [0] running LSD of the chunk
    nz = reinterpret_cast<int*>(res + 2);           // [2 signals][2 flag sets][3 waves] (+ pad)
    x = reinterpret_cast<T*>(nz + 16);
  }
};

// LDS of one workgroup: scratch + exchange arrays, and for the variants with SISpec / log-SISpec running sums one float64
// accumulator per sum and lane ([6][64 NW], updated with ds_add_f64; see ssr_stft_wave_lds_bytes)
template <typename T, int NW, int P = 32> constexpr size_t ssr_stft_rn_wave_sums_offset() { return (SsrRnWaveLds<T, NW, P>::bytes() + 7) & ~(size_t)7; }
template <typename T, int NW, bool SUMS, int P = 32> constexpr size_t ssr_stft_rn_wave_lds_bytes() {
  return SUMS ? ssr_stft_rn_wave_sums_offset<T, NW, P>() + 6 * 64 * NW * sizeof(double) : SsrRnWaveLds<T, NW, P>::bytes();
}

// decimated samples of unit u (frame u of both signals), sub-sequence r: sample NW m + r of the frame, m = lane + 64 i
template <typename T, int NW, int NQ, typename REGS, typename SA = float, typename SB = float>
SSR_DEV void ssr_rn_wave_prefetch(REGS& R, int lane, int r, const SsrView<SA>& va, const SsrView<SB>& vb, int u, int hop,
                                  int n_fft, int q, int n, int n_frames) {
  const int t_c = (u < n_frames) ? u : n_frames - 1;
  const int base = t_c * hop - n_fft / 2;
  const bool interior = base >= 0 && base + n_fft <= n;
  SSR_UNROLL for (int i = 0; i < 4 * NQ; ++i) {
    const int m = lane + 64 * i;
    const int mc = (m < q) ? m : q - 1;
    const int s3 = base + NW * mc + r;
    const unsigned idx = SSR_UIDX(interior ? s3 : ssr_reflect(s3, n));
    R.pa[i] = va.at(idx);
    R.pb[i] = vb.at(idx);
  }
}

// X[K] (already carrying the factor 1/2) from the NW parked sub-spectra
template <typename T, int NW> SSR_DEV cx<T> ssr_rn_combine(const T* yre, const T* yim, int q, int K) {
  if constexpr (NW == 3) return ssr_r3_combine<T>(yre, yim, q, K);
  else if constexpr (NW == 1) return {yre[K], yim[K]};
  else {
    const int j = (K >= q) ? 1 : 0, k = K - j * q;                  // X[k + q j] = y0[k] + (-1)^j y1[k]
    const cx<T> y0 = {yre[k], yim[k]}, y1 = {yre[q + k], yim[q + k]};
    return j ? cx<T>{y0.x - y1.x, y0.y - y1.y} : cx<T>{y0.x + y1.x, y0.y + y1.y};
  }
}

// grid = n_items * n_chunks workgroups of 64 NW threads; PAIR mode, float32 signals, M = 2048, q = n_fft / NW <= 256 NQ.
template <typename T, bool SUMS, int NW, int NQ, int P = 32, typename BLK>
SSR_BODY void ssr_stft_rn_wave_body(const SsrStftParams<T>& p, BLK& blk, int chunk, int item, char* lds_base) {
  constexpr bool SPLIT = true;
  constexpr int NT = 64 * NW, NI = 4 * NQ;                           // input rows i < NI can be non-zero (q <= 64 NI)
  constexpr int PB = P / 8, M = 64 * P, PN = ssr_rn_wave_pn<P>();   // butterflies per lane and pass; transform length; array length
  constexpr int NQO = (P == 32) ? NQ : 4;                            // output rows b + PB qq, qq < NQO, cover k < q
  static_assert(NQ == 3 || NQ == 4, "q <= 768 or q <= 1024");
  static_assert(P == 32 || (P == 24 && NQ == 3), "M = 1536 holds the chirp-z of q <= 768 only");
  using Regs = SsrRnWaveRegs<T, SUMS, NQ, P>;
  SsrRnWaveLds<T, NW, P> L(lds_base);
  const int n_fft = p.n_fft, hop = p.hop, F = n_fft / 2 + 1, q = n_fft / NW;
  T* yre = L.x;                       // parked sub-spectra [NW q] re, [NW q] im: alias the exchange arrays (2113 >= 2 q)
  T* yim = L.x + NW * q;
  const int n = p.len[item];
  const int n_frames = ssr_num_frames_dev(n, n_fft, hop);
  const int u0 = chunk * p.units_per_chunk;
  const int u1 = (u0 + p.units_per_chunk < n_frames) ? u0 + p.units_per_chunk : n_frames;
  const int64_t row0 = p.frame_off[item];
  double* part = p.part ? p.part + ((int64_t)item * p.n_chunks + chunk) * SSR_NPART : nullptr;
  const int mask = SUMS ? p.metric_mask : (p.metric_mask & SSR_M_LSD);
  const bool want_lsd = mask & SSR_M_LSD;
  const SsrView<float> va(p.a + p.a_off[item], n), vb(p.b + p.b_off[item], n);
  const SsrView<cx<T>> vbf(p.bfilt, M), vch(p.chirp, n_fft), vt(p.tw, M + (P == 32 ? SSR_W_TWP : SSR_W24_TWP));

  double* lsum = reinterpret_cast<double*>(lds_base + ssr_stft_rn_wave_sums_offset<T, NW, P>());   // [6][NT], SUMS only
  SSR_REGS(Regs, regs, blk);
  SSR_PHASE(blk, regs, {
    for (int i = tid; i < 6 * 4; i += NT) L.wacc[i] = 0.0;
    if (tid == 0) L.res[0] = 0.0;
    if constexpr (SUMS) for (int i = 0; i < 6; ++i) lsum[NT * i + tid] = 0.0;
    if (u0 < u1) ssr_rn_wave_prefetch<T, NW, NQ>(R, tid & 63, ssr_wave_of(tid), va, vb, u0, hop, n_fft, q, n, n_frames);
  });

  BLK blk0 = blk;
  for (int u = u0; u < u1; ++u) {
    blk = blk0; ssr_launder(blk);
#define SSR_R3_L (SsrWaveBuf<T>{L.x + ssr_wave_of(tid) * PN, L.x + ssr_wave_of(tid) * PN})
    // ---- wave r: decimated frame * (window * chirp) -> registers (only m < q is non-zero: NI of the 32 points), first pass
    SSR_WPHASE(blk, regs, {
      const int lane = tid & 63, r = ssr_wave_of(tid);
      // this wave's q table values as a view of their own: rows m >= q are out of its range and load 0, which zeroes the
      // padded points without a lane mask per row (their sample registers hold a repeat of the frame's sample m = q - 1)
      const SsrView<cx<T>> vwr(p.wchirp + (int64_t)r * q, q);
      unsigned ora = 0u, orb = 0u;
      SSR_UNROLL for (int i = 0; i < P; ++i) {
        if (i < NI) {
          R.v[i] = cmul(cx<T>{(T)R.pa[i], (T)R.pb[i]}, vwr.at_or_zero(SSR_UIDX(lane + 64 * i)));
          // silent-frame vote: every register holds a sample of this frame (the repeats included); frame sample 0 carries
          // window weight exactly 0 (periodic Hann) and does not count
          const bool counts = i > 0 || lane + r != 0;
          ora |= counts ? ssr_mag_bits(R.pa[i]) : 0u;
          orb |= counts ? ssr_mag_bits(R.pb[i]) : 0u;
        } else {
          R.v[i] = cx<T>{(T)0, (T)0};
        }
      }
      // non-zero flags of the frame: one slot per wave (three decimated thirds make a frame), two sets (written a unit ahead
      // of the epilogue that reads them would need no double buffering, but the epilogue of unit u runs before these of u + 1)
      SSR_WAVE_ANY_STORE(lane, ora != 0u, L.nz + ((u - u0) & 1) * 3 + r);
      SSR_WAVE_ANY_STORE(lane, orb != 0u, L.nz + 8 + ((u - u0) & 1) * 3 + r);
      if constexpr (P == 32) ssr_dft32<T, NI>(R.v); else ssr_dft24<T, NI>(R.v);
      if (want_lsd && u > u0 && tid == 0) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += L.sc1[w];
        L.res[0] += sqrt(s / (double)F);
      }
    });
#define VT vt
    if constexpr (P == 32) { SSR_W_FFT_TAIL(blk, blk0, regs, SSR_R3_L, ); } else { SSR_W24_FFT_TAIL(blk, blk0, regs, SSR_R3_L, ); }
    // spectrum * filter; the inverse transform's input register i takes swap(.) of k = lane + 64 i, i = b + PB qq
    SSR_WPHASE(blk, regs, {
      const int lane = tid & 63;
      cx<T> z[P];
      SSR_UNROLL for (int b = 0; b < PB; ++b) SSR_UNROLL for (int qq = 0; qq < 8; ++qq) {
        const cx<T> y = cmul(R.v[8 * b + qq], vbf.at(SSR_UIDX(lane + 64 * (b + PB * qq))));
        z[b + PB * qq] = {y.y, y.x};
      }
      SSR_UNROLL for (int i = 0; i < P; ++i) R.v[i] = z[i];
      if constexpr (P == 32) ssr_dft32(R.v); else ssr_dft24(R.v);
    });
    if constexpr (P == 32) { SSR_W_FFT_TAIL(blk, blk0, regs, SSR_R3_L, ); } else { SSR_W24_FFT_TAIL(blk, blk0, regs, SSR_R3_L, ); }
#undef VT
    // registers hold swap(IFFT * M): true real part = .y, imaginary = .x, at k = lane + 64 (b + 4 qq); k < q is wanted:
    // post-multiply (chirp * W_n^{r k} / 2) in place
    SSR_WPHASE(blk, regs, {
      const int lane = tid & 63, r = ssr_wave_of(tid);
      SSR_UNROLL for (int b = 0; b < PB; ++b) SSR_UNROLL for (int qq = 0; qq < NQO; ++qq) {
        const int k = lane + 64 * (b + PB * qq);
        const cx<T> c = vch.at(SSR_UIDX(k < q ? k : q - 1), (int64_t)r * q);
        R.v[8 * b + qq] = cmul(cx<T>{R.v[8 * b + qq].y, R.v[8 * b + qq].x}, c);
      }
    });
    // ---- all waves are done with their exchange arrays: park the sub-spectra in them
    SSR_PHASE(blk, regs, {});
    SSR_PHASE(blk, regs, {
      const int lane = tid & 63, r = ssr_wave_of(tid);
      SSR_UNROLL for (int b = 0; b < PB; ++b) SSR_UNROLL for (int qq = 0; qq < NQO; ++qq) {
        const int k = lane + 64 * (b + PB * qq);
        if (k < q) { yre[r * q + k] = R.v[8 * b + qq].x; yim[r * q + k] = R.v[8 * b + qq].y; }
      }
    });
    // ---- epilogue: X[K] and X[n - K] from the parked thirds, separate, emit, accumulate
    blk = blk0; ssr_launder(blk);
    const int64_t OP = p.out_pitch ? p.out_pitch : F;      // floats between output rows
    float* ra0 = p.out_a ? p.out_a + (row0 + u) * OP : nullptr;
    float* rb0 = p.out_b ? p.out_b + (row0 + u) * OP : nullptr;
    SSR_PHASE(blk, regs, {
      ssr_rn_wave_prefetch<T, NW, NQ>(R, tid & 63, ssr_wave_of(tid), va, vb, u + 1, hop, n_fft, q, n, n_frames);   // (unconditional)
      SSR_SCHED_BARRIER();
      double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const int par = ((u - u0) & 1) * 3;
      bool a_nz = false, b_nz = false;
      for (int w = 0; w < NW; ++w) { a_nz = a_nz || L.nz[par + w] != 0; b_nz = b_nz || L.nz[8 + par + w] != 0; }
      const bool store = p.out_kind == SSR_OUT_MAG;
      // (two bins at a time with packed float32 arithmetic, as in ssr_stft_wave.h's epilogue, measured SLOWER here: 13.07 against
      // 12.74 ms per 1024 pairs - four radix-3 combines in flight instead of two; profiles/r03_notes.md)
      for (int K = tid; K < F; K += NT) {
        const int Kn = (K == 0) ? 0 : n_fft - K;
        const cx<T> zk = ssr_rn_combine<T, NW>(yre, yim, q, K);
        const cx<T> zn = ssr_rn_combine<T, NW>(yre, yim, q, Kn);
        float e, t;
        ssr_pair_bin<T, 0, true>(mask, acc, zk, zn, a_nz, b_nz, e, t);
        if (store) { ra0[K] = e; if (rb0 != nullptr) rb0[K] = t; }
      }
      if (want_lsd) SSR_WAVE_SUM_STORE(tid, NT, acc[0], L.sc1);
      if constexpr (SUMS)
        for (int i = 0; i < 6; ++i) SSR_LDS_ACCUM(lsum + NT * i + tid, acc[1 + i]);
    });
  }
#undef SSR_R3_L

  if (part == nullptr) return;
  if constexpr (SUMS) {
    SSR_PHASE(blk, regs, for (int i = 0; i < 6; ++i) SSR_WAVE_SUM_ADD(tid, NT, lsum[NT * i + tid], L.wacc + i * 4));
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
      for (int w = 0; w < NW; ++w) s += L.wacc[i * 4 + w];
      part[1 + i] = s;
    }
    part[7] = 0.0;
  });
}
