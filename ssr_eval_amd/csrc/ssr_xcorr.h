// Kernel body N4: argmax of the full cross-correlation of two equal-length signals,
//   z[k] = sum_l a[l] * b[l - k + n - 1],  k = 0 .. 2n-2      (scipy.signal.correlate(a, b, "full")),
// which is how the reference aligns an mp3-decoded signal with its source (ssr_eval/eval.py:319:
// shift = argmax(correlate(decoded, x)) - len(x)).  Only the position of the maximum is needed, so the 2n-1
// correlation values are never written: every workgroup reduces its own block of lags to (value, index) and a
// second tiny kernel picks the first maximum per item (numpy.argmax tie rule: lowest index).
//
// Direct form, register blocked: thread t owns SSR_XC_J consecutive lags, so one pass over l needs one new b sample
// and one (broadcast) a sample per SSR_XC_J multiply-adds.  Products are accumulated in float32 over one l-block
// (SSR_XC_LB terms) and those partial sums in float64.  A 4 s / 48 kHz utterance is 3.7e10 multiply-adds.
// LDS: the a block + the b window of the workgroup's lags, the window index-padded (one spare word per 8) because
// neighbouring threads read it 8 words apart.
#pragma once
#include "ssr_block.h"

#define SSR_XC_NT 256
#define SSR_XC_J 8
#define SSR_XC_LB 512
#define SSR_XC_LAGS (SSR_XC_NT * SSR_XC_J)   // lags per workgroup

struct SsrXcorrParams {
  const float* a; const int64_t* a_off;
  const float* b; const int64_t* b_off;
  const int32_t* len;        // [n_items] n (both signals)
  int n_lag_blocks;          // gridDim.x per item = ceil((2 * max_len - 1) / SSR_XC_LAGS)
  double* best_val;          // [n_items, n_lag_blocks]
  int64_t* best_idx;         // [n_items, n_lag_blocks]
};

SSR_HD int ssr_xc_pad(int q) { return q + (q >> 3); }
struct SsrXcorrLds {
  static constexpr int WIN = SSR_XC_LB + SSR_XC_LAGS;          // b-window samples (one spare at the top)
  static constexpr size_t bytes() { return sizeof(float) * (SSR_XC_LB + WIN + WIN / 8 + 8) + sizeof(double) * 8 + sizeof(int64_t) * 8; }
  float* as; float* bs; double* wv; int64_t* wi;
  SSR_MEMBER explicit SsrXcorrLds(char* base) {
    wv = reinterpret_cast<double*>(base);
    wi = reinterpret_cast<int64_t*>(wv + 8);
    as = reinterpret_cast<float*>(wi + 8);
    bs = as + SSR_XC_LB;
  }
};
struct SsrXcorrRegs { double acc[SSR_XC_J]; double bv; int64_t bi; };

// (value, index) maximum with the lowest index winning ties
SSR_DEV void ssr_xc_better(double v, int64_t i, double& bv, int64_t& bi) {
  if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

// grid = (n_lag_blocks, n_items); block = SSR_XC_NT
template <typename BLK>
SSR_BODY void ssr_xcorr_body(const SsrXcorrParams& p, BLK& blk, int lag_block, int item, char* lds_base) {
  constexpr int NT = SSR_XC_NT, J = SSR_XC_J, LB = SSR_XC_LB, LAGS = SSR_XC_LAGS, WIN = SsrXcorrLds::WIN;
  SsrXcorrLds L(lds_base);
  const int n = p.len[item];
  const int64_t K = 2 * (int64_t)n - 1;
  const int64_t k0 = (int64_t)lag_block * LAGS;
  double* out_v = p.best_val + (int64_t)item * p.n_lag_blocks + lag_block;
  int64_t* out_i = p.best_idx + (int64_t)item * p.n_lag_blocks + lag_block;
  const float* a = p.a + p.a_off[item];
  const float* b = p.b + p.b_off[item];
  SSR_REGS(SsrXcorrRegs, regs, blk);
  if (k0 >= K) {                                   // block of a shorter item in a ragged batch: no lags
    SSR_PHASE(blk, regs, if (tid == 0) { *out_v = -INFINITY; *out_i = 0; });
    return;
  }
  // lag d = k - (n - 1);  z[k] = sum over l in [max(0, d), min(n, n + d)) of a[l] * b[l - d]
  const int64_t dmin = k0 - (n - 1), dmax = dmin + LAGS - 1;
  const int64_t l_lo = dmin > 0 ? dmin : 0;
  const int64_t l_hi = (n + dmax < n) ? n + dmax : n;          // exclusive
  SSR_PHASE(blk, regs, SSR_UNROLL for (int j = 0; j < J; ++j) R.acc[j] = 0.0);
  for (int64_t l0 = l_lo; l0 < l_hi; l0 += LB) {
    const int64_t b0 = l0 - dmax;                              // b index held in window slot 0
    SSR_PHASE(blk, regs, {
      for (int m = tid; m < LB; m += NT) {
        const int64_t l = l0 + m;
        L.as[m] = (l < n) ? a[l] : 0.0f;
      }
      for (int q = tid; q < WIN; q += NT) {
        const int64_t i = b0 + q;
        L.bs[ssr_xc_pad(q)] = (i >= 0 && i < n) ? b[i] : 0.0f;
      }
    });
    SSR_PHASE(blk, regs, {
      // slot of (l = l0 + m, lag j of this thread): q = m + (LAGS - 1) - J * tid - j
      const int q0 = (LAGS - 1) - J * tid;
      float w[J];        // w[j] = b sample for lag j at the current m;  w[j] at step m == w[0] at step m - j
      SSR_UNROLL for (int j = 1; j < J; ++j) w[j] = L.bs[ssr_xc_pad(q0 - j)];
      float s[J];
      SSR_UNROLL for (int j = 0; j < J; ++j) s[j] = 0.0f;
      for (int m = 0; m < LB; m += J) {
        SSR_UNROLL for (int u = 0; u < J; ++u) {
          // rotate: register (j + u) % J plays the role of w[j] at step m + u
          w[(J - u) % J] = L.bs[ssr_xc_pad(q0 + m + u)];
          const float av = L.as[m + u];
          SSR_UNROLL for (int j = 0; j < J; ++j) s[j] += av * w[(j + J - u) % J];
        }
      }
      SSR_UNROLL for (int j = 0; j < J; ++j) R.acc[j] += (double)s[j];
    });
  }
  // per-thread, per-wave, per-workgroup best (lowest index on ties)
  SSR_PHASE(blk, regs, {
    R.bv = -INFINITY; R.bi = 0;
    SSR_UNROLL for (int j = 0; j < J; ++j) {
      const int64_t k = k0 + (int64_t)J * tid + j;
      if (k < K) ssr_xc_better(R.acc[j], k, R.bv, R.bi);
    }
  });
#ifdef SSR_HOST_EMU
  SSR_PHASE(blk, regs, if (tid == 0) {
    double bv = -INFINITY; int64_t bi = 0;
    for (int t = 0; t < NT; ++t) ssr_xc_better(regs[t].bv, regs[t].bi, bv, bi);
    *out_v = bv; *out_i = bi;
  });
#else
  SSR_PHASE(blk, regs, {
    double bv = R.bv; int64_t bi = R.bi;
    SSR_UNROLL for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(bv, o, 64);
      const long long oi = __shfl_xor((long long)bi, o, 64);
      ssr_xc_better(ov, (int64_t)oi, bv, bi);
    }
    if ((tid & 63) == 0) { L.wv[tid >> 6] = bv; L.wi[tid >> 6] = bi; }
  });
  SSR_PHASE(blk, regs, if (tid == 0) {
    double bv = L.wv[0]; int64_t bi = L.wi[0];
    for (int w = 1; w < NT / 64; ++w) ssr_xc_better(L.wv[w], L.wi[w], bv, bi);
    *out_v = bv; *out_i = bi;
  });
#endif
}

// one thread per item: first maximum over the item's lag blocks
SSR_DEV void ssr_xcorr_pick(const double* best_val, const int64_t* best_idx, int n_lag_blocks, int item, int64_t* argmax_out) {
  double bv = -INFINITY; int64_t bi = 0;
  for (int c = 0; c < n_lag_blocks; ++c)
    ssr_xc_better(best_val[(int64_t)item * n_lag_blocks + c], best_idx[(int64_t)item * n_lag_blocks + c], bv, bi);
  argmax_out[item] = bi;
}
