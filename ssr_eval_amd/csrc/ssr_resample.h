// Kernel body K7: polyphase FIR resampler with the arithmetic of scipy.signal.resample_poly /
// upfirdn (mode="constant"), which is what librosa.resample(res_type="polyphase") runs
// (ssr_eval/eval.py:145-150) and what ssr_eval/lowpass.py:137-140 calls directly.
//
// out[m] = sum_i  x[q - i] * h[ph + i*up],   t = (m + n_pre_remove) * down,  q = t / up, ph = t % up,
// with h the Kaiser-windowed sinc (already scaled by `up`, already prefixed by n_pre_pad zeros) and
// x taken as zero outside [0, n_in).  Every output is accumulated in ASCENDING input index with a
// separately rounded float32 multiply and add per tap - the order and rounding of SciPy's
// _upfirdn_apply loop - so the result is bit-identical to the reference's.
//
// Register blocking: outputs m, m + up, m + 2 up, ... share the phase ph (their q advance by `down`), so a
// work item (residue r, group g) walks the taps of ONE phase once and feeds J = 8 independent accumulators:
// 1 tap read + J input reads + J multiply-adds per tap, i.e. (1 + 1/J) LDS reads per multiply-add instead
// of 2, and J independent dependency chains per thread instead of one.
// LDS: the tap table in natural order (when it fits; otherwise taps come from HBM/L2) + the block's input window.
#pragma once
#include "ssr_block.h"

#define SSR_RESAMPLE_NT 256
#define SSR_RESAMPLE_J 8

// S: sample type (float, or double for a float64 signal: SciPy then designs float64 taps and accumulates in float64)
template <typename S> struct SsrResampleParamsT {
  const S* in;
  const int64_t* in_off;   // [n_items]
  const int32_t* in_len;   // [n_items]
  const int64_t* out_off;  // [n_items]
  const int32_t* out_len;  // [n_items]
  int up, down;
  const S* taps;           // [n_taps] = zeros(n_pre_pad) ++ h*up ++ zeros(n_post_pad)
  int n_taps, n_pre_remove;
  int groups;              // G: outputs per block = up * SSR_RESAMPLE_J * G
  int taps_in_lds;         // 0: tap table too large for LDS, read it through L2
  S* out;
};
typedef SsrResampleParamsT<float> SsrResampleParams;

template <typename S> SSR_HD int ssr_resample_hpp(const SsrResampleParamsT<S>& p) { return (p.n_taps + p.up - 1) / p.up; }
template <typename S> SSR_HD int ssr_resample_opb(const SsrResampleParamsT<S>& p) { return p.up * SSR_RESAMPLE_J * p.groups; }
template <typename S> SSR_HD int ssr_resample_win(const SsrResampleParamsT<S>& p) {
  return (int)(((int64_t)ssr_resample_opb(p) * p.down) / p.up) + ssr_resample_hpp(p) + 2;
}
template <typename S> SSR_HD size_t ssr_resample_lds_bytes(const SsrResampleParamsT<S>& p) {
  const size_t taps = p.taps_in_lds ? (size_t)ssr_resample_hpp(p) * p.up : 0;
  return sizeof(S) * (taps + ssr_resample_win(p) + 8);
}
// host-side geometry.  A block's work items are (phase r < up, group g < G), J outputs each, dealt to the NT threads in
// rounds.  G is chosen for (lane efficiency of the rounds) x (workgroups a CU holds at the resulting LDS footprint, at most
// `max_wg_per_cu`, the register-file limit): 441 phases x 4 groups fill 98 % of 7 rounds but their 20 KB window next to the
// 37 KB tap table lets only two workgroups share a CU; x 3 fills 86 % of 6 rounds and three fit.  Caps: the input window
// must fit the prefetch registers, a block at most 16384 outputs.
SSR_HD int ssr_resample_max_wg_per_cu(size_t elem_size) { return elem_size == 4 ? 3 : 2; }   // k_resample's launch bounds
SSR_HD int ssr_resample_pick_groups(int up, int down, int n_taps = 0, size_t elem_size = 4) {
  const int hpp = n_taps > 0 ? (n_taps + up - 1) / up : 21, cap = ssr_resample_max_wg_per_cu(elem_size);
  int best = 1;
  double best_score = 0.0;
  for (int g = 1; g <= 2048; ++g) {                                     // (small `up`: many groups make a full round)
    const int64_t opb = (int64_t)up * SSR_RESAMPLE_J * g;
    const int64_t win = opb * down / up + hpp + 2;
    if (g > 1 && (win + 62 > 10240 || opb > 16384)) break;
    const int64_t items = (int64_t)up * g, rounds = (items + SSR_RESAMPLE_NT - 1) / SSR_RESAMPLE_NT;
    const double eff = (double)items / (double)(rounds * SSR_RESAMPLE_NT);
    size_t lds = elem_size * (size_t)((int64_t)hpp * up + win + 8);
    if (lds > 96 * 1024) lds = elem_size * (size_t)(win + 8);          // tap table left in HBM / L2 (resample_poly_t)
    int per_cu = (int)((160 * 1024) / (lds + 512));
    per_cu = per_cu < 1 ? 1 : (per_cu > cap ? cap : per_cu);
    const double score = eff * per_cu;
    if (score > best_score * 1.02) { best_score = score; best = g; }
    else if (score >= best_score * 0.999) best = g;                     // as good: the larger block (less halo per output)
  }
  return best;
}
#define SSR_RESAMPLE_MAXPF 44   // prefetch registers per thread: ceil(max window / NT)

// Per-block geometry: outputs [m0, m1) of an item and the input window [q_lo, q_lo + win) they read.
struct SsrResampleBlock { int64_t m0, m1, q_lo; int win; bool live; };
template <typename S>
SSR_DEV SsrResampleBlock ssr_resample_block(const SsrResampleParamsT<S>& p, int item, int block) {
  SsrResampleBlock b;
  const int hpp = ssr_resample_hpp(p), opb = ssr_resample_opb(p);
  const int n_out = p.out_len[item];
  b.m0 = (int64_t)block * opb;
  b.live = b.m0 < n_out;
  b.m1 = (b.m0 + opb < n_out) ? b.m0 + opb : n_out;
  if (!b.live) { b.m1 = b.m0; b.q_lo = 0; b.win = 0; return b; }
  const int64_t q_first = ((b.m0 + p.n_pre_remove) * p.down) / p.up;
  b.q_lo = q_first - (hpp - 1);
  const int64_t q_hi = ((b.m1 - 1 + p.n_pre_remove) * p.down) / p.up;
  b.win = (int)(q_hi - b.q_lo + 1);
  return b;
}

// the multiply-adds of one block: window in xw, taps in h
template <typename S>
SSR_DEV void ssr_resample_compute(const SsrResampleParamsT<S>& p, int tid, int item, const SsrResampleBlock& b, const S* xw,
                                  const S* h, int h_len) {
  constexpr int NT = SSR_RESAMPLE_NT, J = SSR_RESAMPLE_J;
  const int hpp = ssr_resample_hpp(p), up = p.up, down = p.down, G = p.groups;
  S* y = p.out + p.out_off[item];
  const int64_t m0 = b.m0, m1 = b.m1, q_lo = b.q_lo;
  // (m + n_pre_remove) * down = Q0 * up + P0 for the block's first output (one 64-bit division per block, uniform); a work
  // item adds r * down + g * J * up * down to it: a 32-bit division of a number below up * (down + 1) per work item instead
  // of a 64-bit one (~100 instructions, a quarter of the loop's arithmetic for 21 taps x 8 outputs)
  const int64_t T0 = (m0 + p.n_pre_remove) * down;
  const int64_t Q0 = T0 / up;
  const unsigned P0 = (unsigned)(T0 - Q0 * up);
  for (int it = tid; it < up * G; it += NT) {
    const int r = it % up, g = it / up;
    const int64_t mf = m0 + r + (int64_t)g * J * up;          // first output of this work item
    if (mf < m1) {
      const unsigned a = P0 + (unsigned)r * (unsigned)down;
      const unsigned qa = a / (unsigned)up;
      const int ph = (int)(a - qa * (unsigned)up);
      const int base = (int)(Q0 - q_lo) + (int)qa + g * J * down;
      S acc[J];
      int xb[J];   // window slot of the OLDEST input sample of output j (outputs past m1 alias output 0, never stored)
      SSR_UNROLL for (int j = 0; j < J; ++j) {
        acc[j] = (S)0;
        xb[j] = (mf + (int64_t)j * up < m1) ? base + j * down - (hpp - 1) : base - (hpp - 1);
      }
      // k ascending = input index ascending (tap index descending): SciPy's accumulation order.
      // Partial unroll keeps several taps' worth of LDS reads in flight per wait.
      int hi = ph + (hpp - 1) * up;
      SSR_UNROLL4 for (int k = 0; k < hpp; ++k) {
        const S hv = (hi < h_len) ? h[hi] : (S)0;
        hi -= up;
        SSR_UNROLL for (int j = 0; j < J; ++j) acc[j] = ssr_fadd_rn(acc[j], ssr_fmul_rn(xw[xb[j] + k], hv));
      }
      SSR_UNROLL for (int j = 0; j < J; ++j) {
        const int64_t m = mf + (int64_t)j * up;
        if (m < m1) y[m] = acc[j];
      }
    }
  }
}

template <typename S> struct SsrResampleRegs { S nx[SSR_RESAMPLE_MAXPF]; };

// PERSISTENT workgroup: stages the tap table ONCE, then walks the (item, block) pairs idx = first, first + stride, ...
// (idx = item * blocks_per_item + block).  The next block's input window is requested into registers before the current
// block's multiply-adds and written to LDS after them, so the HBM latency of a window hides behind a block of arithmetic
// (the one-block-per-workgroup launch re-staged 37 KB of taps per 6144 outputs and waited out every window: 74 % of the
// wave cycles were waits, profiles/r02_notes.md).
template <typename S, typename BLK>
SSR_BODY void ssr_resample_persistent_body(const SsrResampleParamsT<S>& p, BLK& blk, int first, int stride, int total,
                                           int blocks_per_item, char* lds_base) {
  constexpr int NT = SSR_RESAMPLE_NT, PF = SSR_RESAMPLE_MAXPF;
  using Regs = SsrResampleRegs<S>;
  const int hpp = ssr_resample_hpp(p), up = p.up;
  S* hl = reinterpret_cast<S*>(lds_base);
  S* xw = hl + (p.taps_in_lds ? hpp * up : 0);
  const S* h = p.taps_in_lds ? hl : p.taps;
  const int h_len = p.taps_in_lds ? hpp * up : p.n_taps;
  if (first >= total) return;

  SSR_REGS(Regs, regs, blk);
  SSR_PHASE(blk, regs, {
    if (p.taps_in_lds)
      for (int i = tid; i < hpp * up; i += NT) hl[i] = (i < p.n_taps) ? p.taps[i] : (S)0;
    const int item = first / blocks_per_item;
    const SsrResampleBlock b = ssr_resample_block(p, item, first % blocks_per_item);
    const S* x = p.in + p.in_off[item];
    const int n_in = p.in_len[item];
    for (int i = tid; i < b.win; i += NT) {
      const int64_t j = b.q_lo + i;
      xw[i] = (j >= 0 && j < n_in) ? x[j] : (S)0;
    }
  });
  for (int idx = first; idx < total; idx += stride) {
    const int item = idx / blocks_per_item;
    const SsrResampleBlock b = ssr_resample_block(p, item, idx % blocks_per_item);
    const int nidx = idx + stride;
    const bool more = nidx < total;
    const int nitem = more ? nidx / blocks_per_item : item;
    SsrResampleBlock nb = b;
    if (more) nb = ssr_resample_block(p, nitem, nidx % blocks_per_item);
    else nb.win = 0;
    SSR_PHASE(blk, regs, {
      const int n_in_n = p.in_len[nitem];
      const SsrView<S> vn(p.in + p.in_off[nitem], n_in_n);       // (scalar base + 32-bit lane offset per load)
      // next window -> registers (in flight across the arithmetic).  No per-lane condition: an index outside [0, n_in) - a
      // negative one included, as an unsigned offset - is out of the view's range and loads 0, which IS the signal's zero
      // extension; slots beyond the window load something that is never written to LDS.  (44 lane masks kept in scalar
      // registers across the multiply-adds - for the select after the load - were 98 spilled SGPRs.)
      // (round 3: in groups of four behind ONE block-uniform scalar test each - the window of 441/160 needs 16 of the 44 slots,
      // and the other 28 requests per thread fetched 5.7 GB per 12,500 utterances that nobody read: PMC 8.9 GB for 3.2 GB of input)
      const int q_lo_n = (int)nb.q_lo + tid;
      SSR_UNROLL for (int u0 = 0; u0 < PF; u0 += 4)
        if (u0 * NT < nb.win) {
          SSR_UNROLL for (int u = u0; u < u0 + 4 && u < PF; ++u) R.nx[u] = vn.at_or_zero((unsigned)(q_lo_n + u * NT));
        }
      if (b.live) ssr_resample_compute<S>(p, tid, item, b, xw, h, h_len);
    });
    SSR_PHASE(blk, regs, {
      SSR_UNROLL for (int u = 0; u < PF; ++u) {
        const int i = tid + u * NT;
        if (i < nb.win) xw[i] = R.nx[u];
      }
    });
  }
}

// Fallback for rate pairs whose reduced `up` is so large that even the smallest block of the kernel above (up * J
// outputs, all phases once) needs an input window beyond the LDS - e.g. the reference's subsampling of a 16 kHz input at
// cutoff 8000 Hz (low_rate == sr -> -1 quirk: 7349 / 7350, ssr_eval/lowpass.py:134-140, eval.py:404-405).  One output per
// thread, consecutive outputs on consecutive lanes; samples and taps come through L1 / L2 (the tap index advances by
// `down mod up` per output, the input index by about down / up).  Same arithmetic: ascending input index, separately
// rounded multiply and add -> the same bits as SciPy.
template <typename S>
SSR_DEV void ssr_resample_direct_output(const SsrResampleParamsT<S>& p, int item, int64_t m) {
  const int n_in = p.in_len[item], n_out = p.out_len[item];
  if (m >= n_out) return;
  const int hpp = ssr_resample_hpp(p), up = p.up;
  const S* x = p.in + p.in_off[item];
  const int64_t t0 = (m + p.n_pre_remove) * p.down;
  const int64_t q0 = t0 / up;
  const int ph = (int)(t0 - q0 * up);
  S acc = (S)0;
  int64_t hi = ph + (int64_t)(hpp - 1) * up;
  for (int k = 0; k < hpp; ++k) {
    const int64_t i = q0 - (hpp - 1) + k;
    const S hv = (hi < p.n_taps) ? p.taps[hi] : (S)0;
    const S xv = (i >= 0 && i < n_in) ? x[i] : (S)0;
    hi -= up;
    acc = ssr_fadd_rn(acc, ssr_fmul_rn(xv, hv));
  }
  p.out[p.out_off[item] + m] = acc;
}
