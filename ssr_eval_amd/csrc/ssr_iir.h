// Kernel N1: zero-phase IIR filtering = scipy.signal.sosfiltfilt(sos, x) (padtype="odd", default padlen) for a
// ragged batch of float32 signals, float64 arithmetic, float64 output - the arithmetic behind
// lowpass_filter / bandpass_filter (ssr_eval/lowpass.py:54-131).
//
// SciPy's algorithm, reproduced operation for operation (no fused multiply-add, same evaluation order):
//   ext  = odd extension of x by `edge` samples on both sides, formed in FLOAT32 (2*x[0] - x[k]), then widened;
//   fwd  = sosfilt(sos, ext,        zi * ext[0]);      bwd = sosfilt(sos, reverse(fwd), zi * fwd[-1]);
//   y    = reverse(bwd)[edge : -edge]
//   sosfilt, per sample and per section s (direct form II transposed):
//       y  = b0*x + z0;   z0 = (b1*x - a1*y) + z1;   z1 = b2*x - a2*y;   x <- y
// so the result is bit-identical to SciPy's (tests/test_gpu_parity.py::test_sosfiltfilt_bit_exact).
//
// Parallelisation: the recurrence is sequential in n inside a section and sequential in s inside a sample,
// so each utterance is a systolic WAVEFRONT: lane s of a G-lane group owns section s and at step t filters
// sample n = t - s, taking its input from lane s-1 through one DPP row shift.  A wave carries 64/G utterances;
// the critical path per step is the 4-operation float64 state update, independent of the number of sections.
// Inputs and outputs move in chunks of sixteen samples between HBM and registers (device schedule: below).
#pragma once
#include "ssr_block.h"

// X: input sample type (float, or double for a float64 signal - SciPy then extends and filters the float64 values)
template <typename X> struct SsrIirParamsT {
  const X* x;              // signals
  const int64_t* off;      // [n_items] element offset (also used for y)
  const int32_t* len;      // [n_items]
  const double* sos;       // [n_sections, 6]  b0 b1 b2 a0(=1) a1 a2
  const double* zi;        // [n_sections, 2]  sosfilt_zi(sos)
  int n_sections, edge, n_items;
  double* fwd;             // workspace: forward pass output, item i at off[i] + 2*edge*i, length len[i] + 2*edge
  double* y;               // [same layout as x] float64 result
  double* trash;           // 256 bytes of the workspace nobody reads a value from: where lanes without a sample load from and store to
                           // (device schedule below: no vector-memory instruction of the chunk loop sits under a lane condition)
};
typedef SsrIirParamsT<float> SsrIirParams;

// odd-extended input sample n of [0, len + 2*edge), in the signal's own arithmetic as numpy does it
template <typename X> SSR_DEV double ssr_iir_ext(const X* x, int len, int edge, int n) {
#ifndef SSR_HOST_EMU
#pragma clang fp contract(off)
#endif
  if (n < edge) return (double)((X)2 * x[0] - x[edge - n]);
  if (n < edge + len) return (double)x[n - edge];
  return (double)((X)2 * x[len - 1] - x[len - 2 - (n - edge - len)]);
}

// one section, one sample: SciPy's _sosfilt inner statement sequence
SSR_DEV double ssr_iir_step(double xin, double b0, double b1, double b2, double a1, double a2, double& z0, double& z1) {
#ifndef SSR_HOST_EMU
#pragma clang fp contract(off)
#endif
  const double yo = b0 * xin + z0;
  z0 = (b1 * xin - a1 * yo) + z1;
  z1 = b2 * xin - a2 * yo;
  return yo;
}

#ifdef SSR_HOST_EMU
// sequential statement of the same computation (what every lane schedule must reproduce)
template <typename X> static inline void ssr_iir_item_host(const SsrIirParamsT<X>& p, int item) {
  const int len = p.len[item], edge = p.edge, S = p.n_sections, ne = len + 2 * edge;
  const X* x = p.x + p.off[item];
  double* fwd = p.fwd + p.off[item] + (int64_t)2 * edge * item;
  double* y = p.y + p.off[item];
  std::vector<double> z0(S), z1(S);
  const double x0 = ssr_iir_ext(x, len, edge, 0);
  for (int s = 0; s < S; ++s) { z0[s] = p.zi[2 * s] * x0; z1[s] = p.zi[2 * s + 1] * x0; }
  for (int n = 0; n < ne; ++n) {
    double v = ssr_iir_ext(x, len, edge, n);
    for (int s = 0; s < S; ++s) v = ssr_iir_step(v, p.sos[6 * s], p.sos[6 * s + 1], p.sos[6 * s + 2], p.sos[6 * s + 4], p.sos[6 * s + 5], z0[s], z1[s]);
    fwd[n] = v;
  }
  const double y0 = fwd[ne - 1];
  for (int s = 0; s < S; ++s) { z0[s] = p.zi[2 * s] * y0; z1[s] = p.zi[2 * s + 1] * y0; }
  for (int n = 0; n < ne; ++n) {
    double v = fwd[ne - 1 - n];
    for (int s = 0; s < S; ++s) v = ssr_iir_step(v, p.sos[6 * s], p.sos[6 * s + 1], p.sos[6 * s + 2], p.sos[6 * s + 4], p.sos[6 * s + 5], z0[s], z1[s]);
    const int m = (ne - 1 - n) - edge;
    if (m >= 0 && m < len) y[m] = v;
  }
}
#else
// ---------------------------------------------------------------------------------------------------------------------------
// Device schedule (round 6: PACKED wavefront, no LDS).
//
// A launch of this kernel lasts as long as the longest utterance's recurrence - steps x (cycles per step) - however many
// recurrences run beside it; with the designs of an evaluation side by side (ssr_sosfiltfilt_multi) the waves do not even fill the
// SIMDs.  So what counts is the LATENCY of one step.  The round-4/5 schedule (16-lane groups, inputs and outputs staged through LDS,
// an exec-masked LDS store per step) ran ~20 instructions and ~195 cycles per step.  Here:
//   * G lanes per utterance = the smallest power of two >= n_sections (1, 2, 4, 8, 16): every lane of a group owns a section, a wave
//     carries 64 / G utterances (an order-2 design: 64) - the same recurrences on a sixth of the waves;
//   * a chunk is SSR_IIR_CK = 16 steps: lane 0's sixteen input samples sit in registers, loaded TWO chunks ahead with 16-byte
//     vector loads (SSR_IIR_NB = 3 rotating buffers), the last section's sixteen outputs leave with 16-byte vector stores - a step is the nine
//     float64 operations of SciPy's statement sequence plus (G > 1) the DPP hand-off from the lane below: no LDS, no exec masking;
//   * a chunk in which some lane starts or ends its signal, or that touches the odd extension, takes the general per-step path
//     (a few chunks per utterance).  Lanes whose utterance has ended (or that have none) run on: nothing they compute is stored;
//   * NO vector-memory instruction of the regular chunk sits under a lane condition: a lane without a sample loads from / stores to
//     256 bytes of scratch workspace (`trash`).  Under a condition the loaded registers meet their old values at the join and the
//     compiler waits for the load right there (vmcnt(0) behind every chunk's requests: 2,000 cycles per 16 steps, measured - the
//     whole prefetch distance), and a conditional store makes every later wait count as if no store were in flight, i.e. wait for
//     the stores' acknowledgements (the memory counter retires in order).
// Same operations in the same order on the same values: the outputs are bit-identical to the round-5 kernel's and to SciPy's.
#include <type_traits>

constexpr int SSR_IIR_CK = 16;   // steps per chunk
#ifndef SSR_IIR_NB
#define SSR_IIR_NB 3             // rotating input buffers: requests run NB - 1 chunks ahead (5: 242 VGPRs, the same time - measured)
#endif

typedef double ssr_d2u __attribute__((ext_vector_type(2), aligned(8)));    // 16-byte accesses at the signals' own alignment
typedef float ssr_f4u __attribute__((ext_vector_type(4), aligned(4)));

// value of `v` held by the lane one position lower in the same 16-lane row (DPP row_shr:1), two 32-bit halves
SSR_DEV double ssr_dpp_from_lower_lane(double v) {
  union { double d; int i[2]; } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x111, 0xf, 0xf, true);     // (bound_ctrl: a row's lane 0 reads 0 - no `old` value to
  b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x111, 0xf, 0xf, true);     // preset per step; that lane is an s == 0 lane and takes x_in)
  return b.d;
}

// Input of lane s at a step: lane 0 of a group takes the sample `x_in`, the others lane s-1's output of the previous step.
// G == 16 (a group = a whole DPP row): ONE update_dpp per half whose out-of-row value is x_in; G == 1: no hand-off at all.
template <int G> SSR_DEV double ssr_iir_xin(double yout_prev, double x_in, int s) {
  if constexpr (G == 1) {
    return x_in;
  } else if constexpr (G == 16) {
    union { double d; int i[2]; } a, o, b;
    a.d = yout_prev; o.d = x_in;
    b.i[0] = __builtin_amdgcn_update_dpp(o.i[0], a.i[0], 0x111, 0xf, 0xf, false);
    b.i[1] = __builtin_amdgcn_update_dpp(o.i[1], a.i[1], 0x111, 0xf, 0xf, false);
    return b.d;
  } else {
    const double from_lower = ssr_dpp_from_lower_lane(yout_prev);
    return (s == 0) ? x_in : from_lower;
  }
}

// The utterance of a lane's group (ne = 0: none - the lanes past the end of the batch)
template <typename X> struct SsrIirLane {
  const X* x;
  double* fwd;
  double* y;
  double* trash;
  int len, ne;
};

// input sample n of a pass, general form: the odd extension (forward) or the reversed forward output (backward); 0 past the end
template <bool BACKWARD, typename X> SSR_DEV double ssr_iir_load_in(const SsrIirLane<X>& q, int edge, int n) {
  if (n >= q.ne) return 0.0;
  return BACKWARD ? q.fwd[q.ne - 1 - n] : ssr_iir_ext(q.x, q.len, edge, n);
}

// One pass (forward or backward) of the wavefront for the lane's group.
template <int G, bool BACKWARD, typename X>
SSR_DEV void ssr_iir_pass(const SsrIirLane<X>& q, int s, int S, int edge, double b0, double b1, double b2, double a1, double a2,
                          double zi0, double zi1) {
  constexpr int CK = SSR_IIR_CK;
  using XT = typename std::conditional<BACKWARD, double, X>::type;   // the forward pass keeps the signal's own type (widening is exact)
  int ne_max = q.ne;
  for (int o = 32; o > 0; o >>= 1) {
    const int a_ = __shfl_xor(ne_max, o);
    ne_max = a_ > ne_max ? a_ : ne_max;
  }
  const int n_chunks = (ne_max + G - 1 + CK - 1) / CK;              // the last section lags G - 1 steps at most
  const bool last = s == S - 1;
  // initial state: zi * (first input sample of this pass)
  const double first = ssr_iir_load_in<BACKWARD>(q, edge, 0);
  double z0 = zi0 * first, z1 = zi1 * first, yout = 0.0;

  // ---- the REGULAR chunk: every lane of the wave either filters sixteen samples of its signal's interior or has finished
  // the sixteen inputs of chunk c (consumed by lane 0 of the group; every lane of the group requests them), 16-byte loads.
  // A lane whose chunk is not interior reads the scratch block: values nobody uses (the chunk is then not run as a regular one).
  auto load_regular = [&](int c, XT (&b)[CK]) __attribute__((always_inline)) {
#ifdef SSR_IIR_DEV_NO_LOAD             /* developer experiment (wrong results): what the sixteen-sample loads cost */
    SSR_UNROLL for (int k = 0; k < CK; ++k) b[k] = (XT)(0.001 * (c + k));
    return;
#endif
    const int t0 = c * CK;
    const bool inner = BACKWARD ? (t0 + CK <= q.ne) : (t0 >= edge && t0 + CK <= edge + q.len);
    if constexpr (BACKWARD) {
      const ssr_d2u* src = reinterpret_cast<const ssr_d2u*>(inner ? q.fwd + (q.ne - CK - t0) : q.trash);   // fwd[ne - 1 - t0 - k] = block[CK - 1 - k]
      SSR_UNROLL for (int j = 0; j < CK / 2; ++j) { const ssr_d2u v = src[j]; b[CK - 1 - 2 * j] = v.x; b[CK - 2 - 2 * j] = v.y; }
    } else if constexpr (sizeof(X) == 4) {
      const ssr_f4u* src = reinterpret_cast<const ssr_f4u*>(inner ? q.x + (t0 - edge) : reinterpret_cast<const X*>(q.trash));
      SSR_UNROLL for (int j = 0; j < CK / 4; ++j) { const ssr_f4u v = src[j]; b[4 * j] = v.x; b[4 * j + 1] = v.y; b[4 * j + 2] = v.z; b[4 * j + 3] = v.w; }
    } else {
      const ssr_d2u* src = reinterpret_cast<const ssr_d2u*>(inner ? q.x + (t0 - edge) : reinterpret_cast<const X*>(q.trash));
      SSR_UNROLL for (int j = 0; j < CK / 2; ++j) { const ssr_d2u v = src[j]; b[2 * j] = (XT)v.x; b[2 * j + 1] = (XT)v.y; }
    }
  };
  // is chunk c regular?  (wave-uniform)
  auto regular = [&](int c) -> bool {
    const int t0 = c * CK, n0 = t0 - s;                          // the lane filters samples n0 .. n0 + CK - 1
    const bool inner = BACKWARD ? (t0 + CK <= q.ne) : (t0 >= edge && t0 + CK <= edge + q.len), done = t0 >= q.ne;
    const bool full = n0 >= 0 && n0 + CK <= q.ne, idle = n0 >= q.ne;
    // what the last section's lane stores: forward every sample; backward sample n is y[(ne - 1 - n) - edge], kept inside [0, len)
    const bool st_all = BACKWARD ? (n0 >= edge && n0 + CK <= q.ne - edge) : full;
    const bool st_none = BACKWARD ? (n0 + CK <= edge || n0 >= q.ne - edge) : idle;
    return __all((inner || done) && (full || idle) && (!last || st_all || st_none));
  };
  // the sixteen steps of a regular chunk.  ssr_iir_step's statements in PIPELINED order: a wave issues in order, and the only values
  // the next step waits for are y (through the hand-off from the lane below and b0 * x) and z0 - so the next step's hand-off and product
  // are issued right behind y, ahead of this step's state update, and the two dependent chains overlap instead of queueing (same
  // operations on the same operands).  16-byte stores; every lane but a last section's with sixteen samples to keep writes the scratch block.
  auto run_regular = [&](int c, const XT (&b)[CK]) __attribute__((always_inline)) {
#pragma clang fp contract(off)
    const int n0 = c * CK - s;
    double yv[CK];
    double xin = ssr_iir_xin<G>(yout, (double)b[0], s);
    double p0 = b0 * xin;
    SSR_UNROLL for (int k = 0; k < CK; ++k) {
      const double yo = p0 + z0;
      double xin_n = 0.0, p0_n = 0.0;
      if (k + 1 < CK) { xin_n = ssr_iir_xin<G>(yo, (double)b[k + 1], s); p0_n = b0 * xin_n; }
      z0 = (b1 * xin - a1 * yo) + z1;
      z1 = b2 * xin - a2 * yo;
      yv[k] = yo;
      xin = xin_n; p0 = p0_n;
    }
    yout = yv[CK - 1];
    const bool full = n0 >= 0 && n0 + CK <= q.ne;
    const bool keep = last && full && (BACKWARD ? (n0 >= edge && n0 + CK <= q.ne - edge) : true);
#ifdef SSR_IIR_DEV_NO_STORE            /* developer experiment (wrong results): what the sixteen-sample stores cost */
    if (yv[0] == 1.2345e300) q.trash[0] = yv[CK - 1];
    return;
#endif
    if constexpr (BACKWARD) {
      ssr_d2u* dst = reinterpret_cast<ssr_d2u*>(keep ? q.y + ((q.ne - 1 - edge - n0) - (CK - 1)) : q.trash);
      SSR_UNROLL for (int j = 0; j < CK / 2; ++j) { ssr_d2u v; v.x = yv[CK - 1 - 2 * j]; v.y = yv[CK - 2 - 2 * j]; dst[j] = v; }
    } else {
      ssr_d2u* dst = reinterpret_cast<ssr_d2u*>(keep ? q.fwd + n0 : q.trash);
      SSR_UNROLL for (int j = 0; j < CK / 2; ++j) { ssr_d2u v; v.x = yv[2 * j]; v.y = yv[2 * j + 1]; dst[j] = v; }
    }
  };
  // NB regular chunks on NB rotating buffers: chunk c + NB - 1's inputs are requested before chunk c runs.  Straight-line: four
  // loads, sixteen steps, eight stores, NB times - the compiler's memory-counter waits are then exact (vmcnt = the requests issued
  // since), and a wait for inputs requested NB - 1 chunks ago never waits for a store or a younger load.  (NB = 5 - 64 steps between
  // a request and its use - runs at the same 34-39 ns per step as NB = 3: the loop is not waiting for its inputs any more.)
  constexpr int NB = SSR_IIR_NB;
  XT buf[NB][CK];
  auto nb_regular = [&](int c) __attribute__((always_inline)) {
    SSR_UNROLL for (int j = 0; j < NB; ++j) {
      load_regular(c + j + NB - 1, buf[(j + NB - 1) % NB]);        // (the buffer the previous sub-step consumed)
      run_regular(c + j, buf[j]);
    }
  };
  auto nb_are_regular = [&](int c) -> bool {
    bool ok = true;
    SSR_UNROLL for (int j = 0; j < NB; ++j) ok = ok && regular(c + j);
    return ok;
  };

  // ---- any other chunk (a lane starts or ends in it, or it touches the odd extension): general loads, per-step predicates
  auto run_general = [&](int c) {
    const int t0 = c * CK, n0 = t0 - s;
    XT b[CK];
    SSR_UNROLL for (int k = 0; k < CK; ++k) b[k] = (XT)ssr_iir_load_in<BACKWARD>(q, edge, t0 + k);   // (the extension is formed in X: exact)
    SSR_UNROLL for (int k = 0; k < CK; ++k) {
      const int n = n0 + k;
      const double xin = ssr_iir_xin<G>(yout, (double)b[k], s);
      double nz0 = z0, nz1 = z1;
      const double yo = ssr_iir_step(xin, b0, b1, b2, a1, a2, nz0, nz1);
      const bool on = (s < S) && (n >= 0) && (n < q.ne);
      z0 = on ? nz0 : z0;
      z1 = on ? nz1 : z1;
      yout = on ? yo : yout;
      if (on && last) {
        if constexpr (BACKWARD) {
          const int m = (q.ne - 1 - n) - edge;
          if (m >= 0 && m < q.len) q.y[m] = yo;
        } else {
          q.fwd[n] = yo;
        }
      }
    }
  };

  int c = 0;
  while (c < n_chunks) {                                          // (wave-uniform conditions throughout)
    if (nb_are_regular(c)) {
      // a run of regular chunks.  The first NB are peeled so that the loop is entered with the memory requests of a full
      // iteration behind it: preheader and back edge then agree on how many requests follow each buffer's loads
      SSR_UNROLL for (int j = 0; j < NB - 1; ++j) load_regular(c + j, buf[j]);
      nb_regular(c);
      c += NB;
      while (c < n_chunks && nb_are_regular(c)) {
        nb_regular(c);
        c += NB;
      }
      // (the buffers requested ahead are dropped: whoever runs those chunks loads them again)
    } else {
      run_general(c);
      c += 1;
    }
  }
}

// grid = ceil(n_items / (64 / G)) workgroups of ONE wave
template <int G, typename X>
SSR_DEV void ssr_iir_wave(const SsrIirParamsT<X>& p, int wg, int lane) {
  constexpr int GROUPS = 64 / G;
  const int g = lane / G, s = lane % G, S = p.n_sections;
  const int item = wg * GROUPS + g;
  const bool active = item < p.n_items;
  const int it = active ? item : 0;
  SsrIirLane<X> q;
  q.len = p.len[it];
  q.ne = active ? q.len + 2 * p.edge : 0;
  q.x = p.x + p.off[it];
  q.fwd = p.fwd + p.off[it] + (int64_t)2 * p.edge * it;
  q.y = p.y + p.off[it];
  q.trash = p.trash;
  const int sc = s < S ? s : 0;
  const double b0 = p.sos[6 * sc], b1 = p.sos[6 * sc + 1], b2 = p.sos[6 * sc + 2], a1 = p.sos[6 * sc + 4], a2 = p.sos[6 * sc + 5];
  const double zi0 = p.zi[2 * sc], zi1 = p.zi[2 * sc + 1];
  ssr_iir_pass<G, false>(q, s, S, p.edge, b0, b1, b2, a1, a2, zi0, zi1);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the backward pass re-reads `fwd` written by other lanes of this wave
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  ssr_iir_pass<G, true>(q, s, S, p.edge, b0, b1, b2, a1, a2, zi0, zi1);
}

// lanes per utterance for a design of S sections
SSR_HD constexpr int ssr_iir_group(int S) { return S <= 1 ? 1 : S <= 2 ? 2 : S <= 4 ? 4 : S <= 8 ? 8 : 16; }
template <typename X> SSR_DEV void ssr_iir_wave_any(const SsrIirParamsT<X>& p, int wg, int lane) {
  switch (ssr_iir_group(p.n_sections)) {                      // (wave-uniform)
    case 1: ssr_iir_wave<1, X>(p, wg, lane); break;
    case 2: ssr_iir_wave<2, X>(p, wg, lane); break;
    case 4: ssr_iir_wave<4, X>(p, wg, lane); break;
    case 8: ssr_iir_wave<8, X>(p, wg, lane); break;
    default: ssr_iir_wave<16, X>(p, wg, lane); break;
  }
}
#endif
