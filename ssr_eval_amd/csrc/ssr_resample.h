// Kernel body K7: polyphase FIR resampler with the arithmetic of scipy.signal.resample_poly /
// upfirdn (mode="constant"), which is what librosa.resample(res_type="polyphase") runs
// (ssr_eval/eval.py:145-150) and what ssr_eval/lowpass.py:137-140 calls directly.
//
// out[m] = sum_i  x[q - i] * h[ph + i*up],   t = (m + n_pre_remove) * down,  q = t / up, ph = t % up,
// with h the Kaiser-windowed sinc (already scaled by `up`, already prefixed by n_pre_pad zeros) and
// x taken as zero outside [0, n_in).  The sum runs in ASCENDING input index with a separately rounded
// float32 multiply and add per tap - the order and rounding of SciPy's _upfirdn_apply loop - so the
// output is bit-identical to the reference's.
//
// LDS: the whole tap table in natural order (lane stride `down mod up` is odd for every rate pair in
// use, so tap reads are conflict-free) plus the block's input window.
#pragma once
#include "ssr_block.h"

#define SSR_RESAMPLE_NT 256

struct SsrResampleParams {
  const float* in;
  const int64_t* in_off;   // [n_items]
  const int32_t* in_len;   // [n_items]
  const int64_t* out_off;  // [n_items]
  const int32_t* out_len;  // [n_items]
  int up, down;
  const float* taps;       // [n_taps] = zeros(n_pre_pad) ++ h*up ++ zeros(n_post_pad)
  int n_taps, n_pre_remove, outs_per_block;
  float* out;
};

SSR_HD int ssr_resample_hpp(const SsrResampleParams& p) { return (p.n_taps + p.up - 1) / p.up; }
SSR_HD int ssr_resample_win(const SsrResampleParams& p) {
  return (int)(((int64_t)p.outs_per_block * p.down) / p.up) + ssr_resample_hpp(p) + 2;
}
SSR_HD size_t ssr_resample_lds_bytes(const SsrResampleParams& p) {
  return sizeof(float) * ((size_t)ssr_resample_hpp(p) * p.up + ssr_resample_win(p) + 8);
}

// grid = (n_blocks, n_items), block = SSR_RESAMPLE_NT
template <typename BLK>
SSR_BODY void ssr_resample_body(const SsrResampleParams& p, BLK& blk, int block, int item, char* lds_base) {
  constexpr int NT = SSR_RESAMPLE_NT;
  struct Regs { int unused; };
  const int hpp = ssr_resample_hpp(p), up = p.up, down = p.down;
  const int n_in = p.in_len[item], n_out = p.out_len[item];
  const int m0 = block * p.outs_per_block;
  if (m0 >= n_out) return;
  const int m1 = (m0 + p.outs_per_block < n_out) ? m0 + p.outs_per_block : n_out;
  float* h = reinterpret_cast<float*>(lds_base);
  float* xw = h + hpp * up;
  const int64_t q_first = ((int64_t)(m0 + p.n_pre_remove) * down) / up;
  const int64_t q_lo = q_first - (hpp - 1);
  const int64_t q_hi = ((int64_t)(m1 - 1 + p.n_pre_remove) * down) / up;
  const int win = (int)(q_hi - q_lo + 1);
  const float* x = p.in + p.in_off[item];
  float* y = p.out + p.out_off[item];

  SSR_REGS(Regs, regs, blk);
  SSR_PHASE(blk, regs, {
    for (int i = tid; i < hpp * up; i += NT) h[i] = (i < p.n_taps) ? p.taps[i] : 0.0f;
    for (int i = tid; i < win; i += NT) {
      const int64_t j = q_lo + i;
      xw[i] = (j >= 0 && j < n_in) ? x[j] : 0.0f;
    }
  });
  SSR_PHASE(blk, regs, {
    for (int m = m0 + tid; m < m1; m += NT) {
      const int64_t t = (int64_t)(m + p.n_pre_remove) * down;
      const int64_t q = t / up;
      const int ph = (int)(t - q * up);
      const int base = (int)(q - q_lo);
      float acc = 0.0f;
      for (int i = hpp - 1; i >= 0; --i) acc = ssr_fadd_rn(acc, ssr_fmul_rn(xw[base - i], h[ph + i * up]));
      y[m] = acc;
    }
  });
}
