// Kernel body K1-K4 for n_fft = 3 q (q <= 768) - 2229 = 3 * 743, what AudioMetrics(48000) means (ssr_eval/metrics.py:16-19) - on
// FOUR autonomous waves per workgroup that ROTATE through the sub-sequence transforms of consecutive units (round 3).
//
// ssr_stft_rn_wave.h runs a unit (one frame of both signals) as three waves, one per decimated sub-sequence.  At 218 VGPRs a
// CU holds eight waves - two three-wave workgroups use six of the eight slots, and two of the four SIMDs run one wave where the
// others run two.  Here the chirp-z transforms of a chunk are a plain list of jobs, job j = (unit j / 3, sub-sequence j mod 3),
// and wave w of round rho takes job 4 rho + w:
//     round 0:  A0 A1 A2 B0      -> unit A complete
//     round 1:  B1 B2 C0 C1      -> unit B complete
//     round 2:  C2 D0 D1 D2      -> units C and D complete                    (then the pattern repeats)
// Two four-wave workgroups fill the CU, every SIMD runs two waves.
//   * A wave whose unit completes in its round parks its sub-spectrum (2 q values) in its OWN exchange array - free once its
//     transform is done, no barrier needed for that - and after one workgroup barrier all four waves run the unit's epilogue
//     (radix-3 combine, magnitudes, LSD / SISpec terms) over the bins.
//   * A wave whose unit completes in a LATER round (B0 of round 0; C0, C1 of round 1) keeps its sub-spectrum in registers
//     through the epilogue and parks it afterwards in one of two dedicated slots (the slot it needs may still be read by that
//     epilogue), where it survives the next round's transforms.
//   * Two barriers per round (three in the rounds that complete two units): 7 per four units, the three-wave engine has 12.
//     (Both units of such a round in ONE phase - 6 barriers - measured no faster: 10.51-10.55 against 10.49-10.51 ms.)
// LDS: 4 x 12.8 KB exchange arrays + 2 x 2 q values of dedicated slots + scratch = 75.4 KB per workgroup (q = 743), two per CU.
// The SISpec / log-SISpec sums are register accumulators here (the 1536-point engine has the room; lane-private LDS
// accumulators for 256 lanes would not fit next to the second workgroup).
#pragma once
#include "ssr_stft_rn_wave.h"

constexpr int SSR_R3ROT_WAVES = 4;

template <typename T, int P> struct SsrR3RotLds {
  static constexpr int PN = ssr_rn_wave_pn<P>();
  static size_t bytes(int q) { return sizeof(double) * (8 + 24 + 2) + sizeof(int) * 32 + sizeof(T) * (4 * (size_t)PN + 4 * (size_t)q); }
  double* sc1; double* wacc; double* res; int* nz; T* x; T* park;
  SSR_MEMBER explicit SsrR3RotLds(char* base) {
    sc1 = reinterpret_cast<double*>(base);          // [2 units of a round][4 waves] LSD sums of a unit
    wacc = sc1 + 8;                                 // [6][4] per-wave SISpec sums at the chunk end
    res = wacc + 24;                                // [0] running LSD of the chunk
    nz = reinterpret_cast<int*>(res + 2);           // [2 signals][unit & 3][sub-sequence (+ pad)]
    x = reinterpret_cast<T*>(nz + 32);              // four split-exchange arrays
    park = x + 4 * PN;                              // [2 slots][re q | im q]
  }
};

template <typename T, bool SUMS, int NQ, int P, typename SA = float, typename SB = float> struct SsrR3RotRegs : SsrRnWaveRegs<T, SUMS, NQ, P, SA, SB> {
  double sums[6];
};

// X[K] (already carrying the factor 1/2) from three sub-spectra parked at y0 / y1 / y2 (re at [k], im at [q + k])
template <typename T> SSR_DEV cx<T> ssr_r3_combine3(const T* y0p, const T* y1p, const T* y2p, int q, int K) {
  const T c = (T)-0.5, s = (T)0.86602540378443864676;   // W3 = exp(-2 pi i / 3) = c - i s
  const int j = (K >= q ? 1 : 0) + (K >= 2 * q ? 1 : 0), k = K - j * q;       // K < 3 q: no division
  const cx<T> y0 = {y0p[k], y0p[q + k]}, y1 = {y1p[k], y1p[q + k]}, y2 = {y2p[k], y2p[q + k]};
  if (j == 0) return {y0.x + y1.x + y2.x, y0.y + y1.y + y2.y};
  const cx<T> a = (j == 1) ? y1 : y2, b = (j == 1) ? y2 : y1;     // a * W3 + b * conj(W3)   (as ssr_r3_combine)
  const T sr = a.x + b.x, si = a.y + b.y;
  const T dr = a.x - b.x, di = a.y - b.y;
  return {y0.x + c * sr + s * di, y0.y + c * si - s * dr};
}

// round in which the last job of unit U runs
SSR_DEV int ssr_r3_rot_done_round(int U) { return (3 * U + 2) >> 2; }

// grid = n_items * n_chunks workgroups of 256 threads; PAIR mode, n_fft = 3 q.  IN64 = 0: float32 signals; SSR_IN_EST64 (round 5): the
// estimate as float64 samples (p.a64) with the float64 estimate arithmetic of ssr_pair_bin<T, SSR_IN_EST64> - what an IIR-degraded
// input carries into the metrics (ssr_eval/eval.py:138-150); the target stays float32.  SSR_IN_EST64X2 (round 6): TWO float64 estimates
// (p.a64, p.b64) ride one complex transform and leave as two float32 magnitude rows - |.| of the unrounded float64 spectrum, rounded
// once - with no metric term: ssr_pair_metrics_multi_est64 reduces them against the target image that key 0's pass stored.
template <typename T, bool SUMS, int NQ, int P, int IN64 = 0, typename BLK>
SSR_BODY void ssr_stft_r3_rot_body(const SsrStftParams<T>& p, BLK& blk, int chunk, int item, char* lds_base) {
  static_assert(IN64 == 0 || IN64 == SSR_IN_EST64 || (IN64 == SSR_IN_EST64X2 && !SUMS && sizeof(T) == 8),
                "float32 pairs, a float64 estimate against a float32 target, or two float64 estimates (images only)");
  using SA = typename SsrSample<(IN64 & 1) != 0>::type;
  using SB = typename SsrSample<(IN64 & 2) != 0>::type;
  constexpr bool SPLIT = true;
  // float64 estimate WITH running sums: the next job's samples (the estimate's as doubles: 24 more registers than floats) are
  // requested after the round's epilogue(s), not before - held across them they put 24 values into scratch (52 B per lane, round 5)
  constexpr bool LATE_PREFETCH = (IN64 & 1) != 0 && SUMS;
  constexpr int NT = 64 * SSR_R3ROT_WAVES, NI = 4 * NQ;
  constexpr int PB = P / 8, M = 64 * P, PN = ssr_rn_wave_pn<P>();
  constexpr int NQO = (P == 32) ? NQ : 4;
  static_assert(NQ == 3 || NQ == 4, "q <= 768 or q <= 1024");
  static_assert(P == 32 || (P == 24 && NQ == 3), "M = 1536 holds the chirp-z of q <= 768 only");
  using Regs = SsrR3RotRegs<T, SUMS, NQ, P, SA, SB>;
  SsrR3RotLds<T, P> L(lds_base);
  const int n_fft = p.n_fft, hop = p.hop, F = n_fft / 2 + 1, q = n_fft / 3;
  const int n = p.len[item];
  const int n_frames = ssr_num_frames_dev(n, n_fft, hop);
  const int u0 = chunk * p.units_per_chunk;
  const int u1 = (u0 + p.units_per_chunk < n_frames) ? u0 + p.units_per_chunk : n_frames;
  const int n_units = (u1 > u0) ? u1 - u0 : 0, n_jobs = 3 * n_units, n_rounds = (n_jobs + 3) >> 2;
  const int64_t row0 = p.frame_off[item];
  double* part = p.part ? p.part + ((int64_t)item * p.n_chunks + chunk) * SSR_NPART : nullptr;
  const int mask = SUMS ? p.metric_mask : (p.metric_mask & SSR_M_LSD);
  const bool want_lsd = mask & SSR_M_LSD;
  const SA* sa;
  if constexpr (IN64 & 1) sa = p.a64 + p.a_off[item]; else sa = p.a + p.a_off[item];
  const SB* sb;
  if constexpr (IN64 & 2) sb = p.b64 + p.b_off[item]; else sb = p.b + p.b_off[item];
  const SsrView<SA> va(sa, n);
  const SsrView<SB> vb(sb, n);
  const SsrView<cx<T>> vbf(p.bfilt, M), vch(p.chirp, n_fft), vt(p.tw, M + (P == 32 ? SSR_W_TWP : SSR_W24_TWP));
  const int64_t OP = p.out_pitch ? p.out_pitch : F;      // floats between output rows
  const bool store = p.out_kind == SSR_OUT_MAG;

  SSR_REGS(Regs, regs, blk);
  SSR_PHASE(blk, regs, {
    for (int i = tid; i < 6 * 4; i += NT) L.wacc[i] = 0.0;
    if (tid == 0) L.res[0] = 0.0;
    for (int i = 0; i < 6; ++i) R.sums[i] = 0.0;
    if (n_jobs > 0) {                                    // the wave's first job (clamped: a wave without one repeats the last)
      const int w = ssr_wave_of(tid), jc = (w < n_jobs) ? w : n_jobs - 1, U = jc / 3;
      ssr_rn_wave_prefetch<T, 3, NQ>(R, tid & 63, jc - 3 * U, va, vb, u0 + U, hop, n_fft, q, n, n_frames);
    }
  });

  BLK blk0 = blk;
  for (int rho = 0; rho < n_rounds; ++rho) {
    blk = blk0; ssr_launder(blk);
#define SSR_ROT_L (SsrWaveBuf<T>{L.x + ssr_wave_of(tid) * PN, L.x + ssr_wave_of(tid) * PN})
    // ---- the wave's job: decimated frame * (window * chirp) -> registers, first pass.  A wave past the last job of the chunk
    // repeats that job's transform (its samples were prefetched clamped) and neither votes nor parks.
    SSR_WPHASE(blk, regs, {
      const int lane = tid & 63, j = 4 * rho + ssr_wave_of(tid);
      const bool valid = j < n_jobs;
      const int jc = valid ? j : n_jobs - 1, U = jc / 3, r = jc - 3 * U;
      const SsrView<cx<T>> vwr(p.wchirp + (int64_t)r * q, q);       // rows m >= q are out of range and load 0 (ssr_stft_rn_wave.h)
      unsigned ora = 0u, orb = 0u;
      SSR_UNROLL for (int i = 0; i < P; ++i) {
        if (i < NI) {
          R.v[i] = cmul(cx<T>{(T)R.pa[i], (T)R.pb[i]}, vwr.at_or_zero(SSR_UIDX(lane + 64 * i)));
          const bool counts = i > 0 || lane + r != 0;               // frame sample 0 carries window weight exactly 0
          ora |= counts ? ssr_mag_bits(R.pa[i]) : 0u;
          orb |= counts ? ssr_mag_bits(R.pb[i]) : 0u;
        } else {
          R.v[i] = cx<T>{(T)0, (T)0};
        }
      }
      // non-zero flags of the frame, one slot per (unit & 3, sub-sequence): at most three units are in flight
      if (valid) {
        SSR_WAVE_ANY_STORE(lane, ora != 0u, L.nz + (U & 3) * 4 + r);
        SSR_WAVE_ANY_STORE(lane, orb != 0u, L.nz + 16 + (U & 3) * 4 + r);
      }
      if constexpr (P == 32) ssr_dft32<T, NI>(R.v); else ssr_dft24<T, NI>(R.v);
    });
#define VT vt
    if constexpr (P == 32) { SSR_W_FFT_TAIL(blk, blk0, regs, SSR_ROT_L, ); } else { SSR_W24_FFT_TAIL(blk, blk0, regs, SSR_ROT_L, ); }
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
    if constexpr (P == 32) { SSR_W_FFT_TAIL(blk, blk0, regs, SSR_ROT_L, ); } else { SSR_W24_FFT_TAIL(blk, blk0, regs, SSR_ROT_L, ); }
#undef VT
    // registers hold swap(IFFT * M) at k = lane + 64 (b + PB qq); k < q is wanted: post-multiply (chirp * W_n^{r k} / 2) in
    // place, request the wave's next job's samples, and - if the unit completes in this round - park in the wave's own array
    SSR_WPHASE(blk, regs, {
      const int lane = tid & 63, w = ssr_wave_of(tid), j = 4 * rho + w;
      const bool valid = j < n_jobs;
      const int jc = valid ? j : n_jobs - 1, U = jc / 3, r = jc - 3 * U;
      SSR_UNROLL for (int b = 0; b < PB; ++b) SSR_UNROLL for (int qq = 0; qq < NQO; ++qq) {
        const int k = lane + 64 * (b + PB * qq);
        const cx<T> c = vch.at(SSR_UIDX(k < q ? k : q - 1), (int64_t)r * q);
        R.v[8 * b + qq] = cmul(cx<T>{R.v[8 * b + qq].y, R.v[8 * b + qq].x}, c);
      }
      if constexpr (!LATE_PREFETCH) {
        const int jn = (j + 4 < n_jobs) ? j + 4 : n_jobs - 1, Un = jn / 3;
        ssr_rn_wave_prefetch<T, 3, NQ>(R, lane, jn - 3 * Un, va, vb, u0 + Un, hop, n_fft, q, n, n_frames);   // (unconditional)
      }
      if (valid && ssr_r3_rot_done_round(U) == rho) {
        T* own = L.x + w * PN;
        SSR_UNROLL for (int b = 0; b < PB; ++b) SSR_UNROLL for (int qq = 0; qq < NQO; ++qq) {
          const int k = lane + 64 * (b + PB * qq);
          if (k < q) { own[k] = R.v[8 * b + qq].x; own[q + k] = R.v[8 * b + qq].y; }
        }
      }
    });
    SSR_PHASE(blk, regs, {});                            // every wave's sub-spectrum of this round is parked (or held in registers)
    // ---- the units that complete in this round: U with 4 rho <= 3 U + 2 <= 4 rho + 3
    blk = blk0; ssr_launder(blk);
    const int Ulo = (4 * rho) / 3, Uhi = (4 * rho + 1) / 3;          // ceil((4 rho - 2) / 3) .. floor((4 rho + 1) / 3)
    for (int Uc = Ulo; Uc <= Uhi && Uc < n_units; ++Uc) {
      float* ra0 = p.out_a ? p.out_a + (row0 + u0 + Uc) * OP : nullptr;
      float* rb0 = p.out_b ? p.out_b + (row0 + u0 + Uc) * OP : nullptr;
      const T* yb[3];
      for (int r = 0; r < 3; ++r) {
        const int jr = 3 * Uc + r;
        yb[r] = ((jr >> 2) == rho) ? L.x + (jr & 3) * PN : L.park + r * 2 * q;
      }
      const int e = Uc - Ulo;
      SSR_PHASE(blk, regs, {
        double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        bool a_nz = false, b_nz = false;
        for (int r = 0; r < 3; ++r) { a_nz = a_nz || L.nz[(Uc & 3) * 4 + r] != 0; b_nz = b_nz || L.nz[16 + (Uc & 3) * 4 + r] != 0; }
        for (int K = tid; K < F; K += NT) {
          const int Kn = (K == 0) ? 0 : n_fft - K;
          const cx<T> zk = ssr_r3_combine3<T>(yb[0], yb[1], yb[2], q, K);
          const cx<T> zn = ssr_r3_combine3<T>(yb[0], yb[1], yb[2], q, Kn);
          float ev, tv;
          if constexpr (IN64 == SSR_IN_EST64X2) {               // numpy.abs(complex128) of both estimates, rounded once (ssr_pair_bin<T, SSR_IN_EST64>'s e)
            ev = a_nz ? (float)ssr_cabs_d((double)zk.x + (double)zn.x, (double)zk.y - (double)zn.y) : 0.0f;
            tv = b_nz ? (float)ssr_cabs_d((double)zk.y + (double)zn.y, (double)zn.x - (double)zk.x) : 0.0f;
          } else {
            ssr_pair_bin<T, IN64, IN64 == 0>(mask, acc, zk, zn, a_nz, b_nz, ev, tv);
          }
          if (store) { ra0[K] = ev; if (rb0 != nullptr) rb0[K] = tv; }
        }
        if (want_lsd) SSR_WAVE_SUM_STORE(tid, NT, acc[0], L.sc1 + 4 * e);
        if constexpr (SUMS)
          for (int i = 0; i < 6; ++i) R.sums[i] += acc[1 + i];
      });
    }
    // ---- after the epilogue(s): the sub-spectra of units that complete later leave the registers for the dedicated slots (the
    // epilogue above may have read them), and thread 0 folds the completed units' LSD.  Wave-scope: the next reader of the
    // slots is an epilogue behind a barrier.
    SSR_WPHASE(blk, regs, {
      const int lane = tid & 63, j = 4 * rho + ssr_wave_of(tid);
      const bool valid = j < n_jobs;
      const int jc = valid ? j : n_jobs - 1, U = jc / 3, r = jc - 3 * U;
      if (valid && ssr_r3_rot_done_round(U) > rho) {
        T* slot = L.park + r * 2 * q;
        SSR_UNROLL for (int b = 0; b < PB; ++b) SSR_UNROLL for (int qq = 0; qq < NQO; ++qq) {
          const int k = lane + 64 * (b + PB * qq);
          if (k < q) { slot[k] = R.v[8 * b + qq].x; slot[q + k] = R.v[8 * b + qq].y; }
        }
      }
      if constexpr (LATE_PREFETCH) {
        const int jn = (j + 4 < n_jobs) ? j + 4 : n_jobs - 1, Un = jn / 3;
        ssr_rn_wave_prefetch<T, 3, NQ>(R, lane, jn - 3 * Un, va, vb, u0 + Un, hop, n_fft, q, n, n_frames);   // (unconditional)
      }
      if (want_lsd && tid == 0)
        for (int Uc = Ulo; Uc <= Uhi && Uc < n_units; ++Uc) {
          double s = 0.0;
          for (int w = 0; w < SSR_R3ROT_WAVES; ++w) s += L.sc1[4 * (Uc - Ulo) + w];
          L.res[0] += sqrt(s / (double)F);
        }
    });
  }
#undef SSR_ROT_L

  if (part == nullptr) return;
  if constexpr (SUMS) {
    SSR_PHASE(blk, regs, for (int i = 0; i < 6; ++i) SSR_WAVE_SUM_STORE(tid, NT, R.sums[i], L.wacc + i * 4));
  }
  SSR_PHASE(blk, regs, if (tid == 0) {
    part[0] = L.res[0];
    for (int i = 0; i < 6; ++i) {
      double s = 0.0;
      for (int w = 0; w < SSR_R3ROT_WAVES; ++w) s += L.wacc[i * 4 + w];
      part[1 + i] = s;
    }
    part[7] = 0.0;
  });
}
