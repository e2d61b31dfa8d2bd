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
// Input / output are staged through LDS in chunks (next chunk's loads are issued a whole chunk early).
#pragma once
#include "ssr_block.h"

#define SSR_IIR_CH 128   // samples per staging chunk (> max group size)

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
// value of `v` held by the lane one position lower in the same 16-lane row (DPP row_shr:1), two 32-bit halves
SSR_DEV double ssr_dpp_from_lower_lane(double v) {
  union { double d; int i[2]; } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x111, 0xf, 0xf, false);
  b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x111, 0xf, 0xf, false);
  return b.d;
}

// Input of lane s at a step: lane 0 of a group takes the staged sample `x_in`, the others lane s-1's output of the previous step.
// G == 16 (a group = a whole DPP row): ONE update_dpp per half whose out-of-row value is x_in - no select on the step's dependent chain.
template <int G> SSR_DEV double ssr_iir_xin(double yout_prev, double x_in, int s) {
  if constexpr (G == 16) {
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

// Per-lane view of one utterance slot.  A lane can serve U utterances at once (independent recurrences
// interleaved in one instruction stream); measured on MI355X the step time grows almost linearly with U, so
// U = 1 is the default.
#define SSR_IIR_U 1   /* utterances per lane slot: 2 measured 1.67x the latency for 2x the work - only pays beyond ~8k utterances */
template <int G, typename X> struct SsrIirSlot {
  bool active;
  const X* x;
  int len, ne;
  double* fwd;
  double* y;
  double* in_buf;    // LDS ring of 2*CH inputs of this (group, slot)
  double* out_buf;   // LDS ring of 2*CH outputs
  double z0, z1, yout;
  double pre[SSR_IIR_CH / G];
};

template <int G, bool BACKWARD, typename X>
SSR_DEV double ssr_iir_load_in(const SsrIirSlot<G, X>& q, int edge, int n) {
  if (n >= q.ne) return 0.0;
  return BACKWARD ? q.fwd[q.ne - 1 - n] : ssr_iir_ext(q.x, q.len, edge, n);
}

// One pass (forward or backward) of the wavefront for the lane's group.  G: lanes per utterance (8 or 16).
template <int G, bool BACKWARD, typename X>
SSR_DEV void ssr_iir_pass(const SsrIirParamsT<X>& p, int s, SsrIirSlot<G, X> (&sl)[SSR_IIR_U], double b0, double b1, double b2,
                          double a1, double a2, double zi0, double zi1) {
  constexpr int CH = SSR_IIR_CH, PER = CH / G, U = SSR_IIR_U;
  const int edge = p.edge, S = p.n_sections;
  // wave-uniform trip count (longest utterance of the wave) and the shortest ACTIVE one: blocks of steps inside
  // [S-1, ne_min) need no per-lane predicates
  int ne_max = 0, ne_min = 0x7fffffff;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    ne_max = sl[u].ne > ne_max ? sl[u].ne : ne_max;
    if (sl[u].active) ne_min = sl[u].ne < ne_min ? sl[u].ne : ne_min;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int a_ = __shfl_xor(ne_max, o), b_ = __shfl_xor(ne_min, o);
    ne_max = a_ > ne_max ? a_ : ne_max;
    ne_min = b_ < ne_min ? b_ : ne_min;
  }
  const int n_chunks = (ne_max + S - 1 + CH - 1) / CH + 1;      // +1: flush of the last outputs

#pragma unroll
  for (int u = 0; u < U; ++u) {
    // initial state: zi * (first input sample of this pass); chunk 0 staged directly
    const double first = sl[u].active ? ssr_iir_load_in<G, BACKWARD>(sl[u], edge, 0) : 0.0;
    sl[u].z0 = zi0 * first; sl[u].z1 = zi1 * first; sl[u].yout = 0.0;
    for (int i = 0; i < PER; ++i) sl[u].in_buf[s * PER + i] = ssr_iir_load_in<G, BACKWARD>(sl[u], edge, s * PER + i);
  }
  for (int c = 0; c < n_chunks; ++c) {
    // issue the NEXT chunk's loads now; they land while this chunk is being filtered
#pragma unroll
    for (int u = 0; u < U; ++u)
      for (int i = 0; i < PER; ++i) sl[u].pre[i] = ssr_iir_load_in<G, BACKWARD>(sl[u], edge, (c + 1) * CH + s * PER + i);
    const int ring = (c & 1) * CH;
    for (int tb = 0; tb < CH; tb += 8) {
      double x8[U][8];                                           // lane 0's next eight inputs: all LDS reads in flight at once
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) x8[u][k] = sl[u].in_buf[ring + tb + k];
      const int t0 = c * CH + tb;
      if (t0 >= S - 1 && t0 + 8 <= ne_min) {
        // interior block (wave-uniform): every lane s < S has a valid sample at every step -> no predicates;
        // lanes s >= S compute on garbage that nobody reads
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const double xin = ssr_iir_xin<G>(sl[u].yout, x8[u][k], s);       // lane s-1's output of step t-1 (lane 0: the sample)
            sl[u].yout = ssr_iir_step(xin, b0, b1, b2, a1, a2, sl[u].z0, sl[u].z1);
            if (s == S - 1) sl[u].out_buf[(t0 + k - s) & (2 * CH - 1)] = sl[u].yout;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int n = t0 + k - s;                                // sample this lane filters at this step
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const double xin = ssr_iir_xin<G>(sl[u].yout, x8[u][k], s);
            double nz0 = sl[u].z0, nz1 = sl[u].z1;
            const double yo = ssr_iir_step(xin, b0, b1, b2, a1, a2, nz0, nz1);
            const bool on = (s < S) && (n >= 0) && (n < sl[u].ne);
            sl[u].z0 = on ? nz0 : sl[u].z0;
            sl[u].z1 = on ? nz1 : sl[u].z1;
            sl[u].yout = on ? yo : sl[u].yout;
            if (on && s == S - 1) sl[u].out_buf[n & (2 * CH - 1)] = yo;
          }
        }
      }
    }
    // flush outputs of chunk c-1 (the last section lags by S-1 < CH steps, so they are complete now)
    if (c >= 1) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        for (int i = 0; i < PER; ++i) {
          const int n = (c - 1) * CH + s * PER + i;
          if (n < sl[u].ne) {
            const double v = sl[u].out_buf[n & (2 * CH - 1)];
            if (BACKWARD) {
              const int m = (sl[u].ne - 1 - n) - edge;
              if (m >= 0 && m < sl[u].len) sl[u].y[m] = v;
            } else {
              sl[u].fwd[n] = v;
            }
          }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      for (int i = 0; i < PER; ++i) sl[u].in_buf[((c + 1) & 1) * CH + s * PER + i] = sl[u].pre[i];
  }
}

// grid = ceil(n_items / (U * 64/G)) workgroups of ONE wave; LDS: (64/G) groups * U slots * 4*CH doubles
template <int G, typename X>
SSR_DEV void ssr_iir_wave(const SsrIirParamsT<X>& p, int wg, int lane, char* lds_base) {
  constexpr int CH = SSR_IIR_CH, GROUPS = 64 / G, U = SSR_IIR_U;
  const int g = lane / G, s = lane % G, S = p.n_sections;
  SsrIirSlot<G, X> sl[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int item = (wg * GROUPS + g) * U + u;
    sl[u].active = item < p.n_items;
    const int it = sl[u].active ? item : 0;
    sl[u].len = p.len[it];
    sl[u].ne = sl[u].active ? sl[u].len + 2 * p.edge : 0;
    sl[u].x = p.x + p.off[it];
    sl[u].fwd = p.fwd + p.off[it] + (int64_t)2 * p.edge * it;
    sl[u].y = p.y + p.off[it];
    sl[u].in_buf = reinterpret_cast<double*>(lds_base) + ((size_t)g * U + u) * 4 * CH;
    sl[u].out_buf = sl[u].in_buf + 2 * CH;
  }
  const int sc = s < S ? s : 0;
  const double b0 = p.sos[6 * sc], b1 = p.sos[6 * sc + 1], b2 = p.sos[6 * sc + 2], a1 = p.sos[6 * sc + 4], a2 = p.sos[6 * sc + 5];
  const double zi0 = p.zi[2 * sc], zi1 = p.zi[2 * sc + 1];
  ssr_iir_pass<G, false>(p, s, sl, b0, b1, b2, a1, a2, zi0, zi1);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the backward pass re-reads `fwd` written by other lanes of this wave
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  ssr_iir_pass<G, true>(p, s, sl, b0, b1, b2, a1, a2, zi0, zi1);
}
#endif
