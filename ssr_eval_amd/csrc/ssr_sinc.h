// Kernel body N2: band-limited (windowed-sinc) resampler with the arithmetic of resampy.resample(filter="kaiser_best"),
// which is what librosa.load(file, sr) / librosa.resample(res_type="kaiser_best") run when a file's rate differs from the
// requested one (ssr_eval/eval.py:242, ssr_eval/metrics.py:22-23).  resampy is a third-party package absent from the
// reference tree and from the image: this follows its published algorithm (resampy 0.2.x interpn.resample_f; J.O. Smith's
// bandlimited interpolation) - parity with the real package is unpinned.
//
// For output sample t (time register tr[t] = t / ratio, accumulated by the host exactly as resampy's loop does):
//   n = int(tr), frac = scale * (tr - n), scale = min(1, ratio)
//   left wing : offset = int(frac * num_table), eta = frac * num_table - offset,
//               y += (win[offset + i*step] + eta * delta[offset + i*step]) * x[n - i],  i = 0 .. min(n + 1, (nwin - offset) / step) - 1
//   right wing: frac = scale - frac, same offset / eta, y += (...) * x[n + k + 1],      k = 0 .. min(n_in - n - 1, (nwin - offset) / step) - 1
// with float64 weights and products and the running sum y rounded to float32 after every tap (resampy accumulates
// into the float32 output array).  No fused multiply-add: every product and sum is rounded separately, so the result is
// bit-identical to the NumPy restatement (oracle/resampy.py) given the same tables.
#pragma once
#include "ssr_block.h"

struct SsrSincParams {
  const float* in;
  const int64_t* in_off;    // [n_items]
  const int32_t* in_len;    // [n_items]
  const int64_t* out_off;   // [n_items]
  const int32_t* out_len;   // [n_items]  int(n_in * ratio)
  const double* time_reg;   // [>= max_out_len] time register of output t (independent of the item)
  const double* win;        // [nwin]  interp_win (already scaled by ratio when ratio < 1)
  const double* delta;      // [nwin]  interp_delta: win[i + 1] - win[i], last entry 0
  int nwin, num_table, index_step;
  double scale;             // min(1, ratio)
  float* out;
};

SSR_DEV void ssr_sinc_output(const SsrSincParams& p, int item, int64_t t) {
  const int n_in = p.in_len[item], n_out = p.out_len[item];
  if (t >= n_out) return;
  const float* x = p.in + p.in_off[item];
  const double tr = p.time_reg[t];
  const int n = (int)tr;
  double frac = ssr_fmul_rn(p.scale, ssr_fadd_rn(tr, -(double)n));
  float y = 0.0f;
  for (int wing = 0; wing < 2; ++wing) {
    const double index_frac = ssr_fmul_rn(frac, (double)p.num_table);
    const int offset = (int)index_frac;
    const double eta = ssr_fadd_rn(index_frac, -(double)offset);
    const int room = (p.nwin - offset) / p.index_step;
    const int avail = wing == 0 ? n + 1 : n_in - n - 1;
    const int cnt = room < avail ? room : avail;
    for (int i = 0; i < cnt; ++i) {
      const int idx = offset + i * p.index_step;
      const double weight = ssr_fadd_rn(p.win[idx], ssr_fmul_rn(eta, p.delta[idx]));
      const double xv = (double)(wing == 0 ? x[n - i] : x[n + i + 1]);
      y = (float)ssr_fadd_rn((double)y, ssr_fmul_rn(weight, xv));
    }
    frac = ssr_fadd_rn(p.scale, -frac);           // "invert P"
  }
  p.out[p.out_off[item] + t] = y;
}
