// K3 / K4 on magnitude images that already sit in HBM, one wave per run of rows: the LSD / SISpec / log-SISpec terms of
// AudioMetrics.lsd / .sispec (ssr_eval/metrics.py:109-121, ssr_eval/utils.py:43-44,79-92) for the estimates of
// ssr_pair_metrics_multi whose spectra were computed two per complex transform (their target's image was written once, by the
// pass that transformed the target).  The per-bin arithmetic is ssr_accumulate_metrics<true> - the very sequences the fused
// epilogue of k_stft_wave runs - on the float32 magnitudes the transform kernel stored, so a value differs from the fused
// pipeline's only through the order of the float64 sums and through the estimate's own magnitudes (paired with another estimate
// instead of the target: the two-for-one separation leaves ~1e-16 of the partner's spectrum, i.e. a last-bit flip of a float32
// magnitude now and then).
// Rows are padded to a multiple of four floats (the pair pipeline's pitch): a lane reads bins 4 l + 256 j .. + 3 with one 16-byte
// load per image, row and j; the per-row LSD is a wave reduction, the six SISpec sums stay lane-private float64 registers until
// the end of the chunk.  Device code only.
#pragma once
#include "ssr_stft.h"

struct SsrSpecWaveParams {
  const float* x;              // estimate planes: key k at x + k * x_plane
  const float* y;              // the shared target image
  const int64_t* frame_off;    // [vi_n]
  const int32_t* n_rows;       // [vi_n]
  int F, pitch, metric_mask, rows_per_chunk, n_chunks, vi_n;
  int64_t x_plane;
  double* part;                // [n_keys * vi_n, n_chunks, SSR_NPART]
};

// KG keys per wave: the target row is fetched once for KG estimates (2 KG images -> KG + 1), and the target-only terms
// (t * t, log10(t + 1e-12), the two target-only sums) are formed once per bin.
//
// The kernel is bound by VALU issue, not by HBM (round 4: ~50 issue slots per element with scalar float32 sequences; the fused
// epilogue of k_stft_wave spends ~26 on the same terms): with the full metric mask the float32 arithmetic runs on (bin, bin + 1)
// register pairs - ssr_divf_fast2 / ssr_log10f_fast2, component for component the scalar sequences - and the sums that depend
// on the target alone (sum t^2, sum log10(t)^2) are accumulated once per group instead of once per key.  Every value and the
// order in which a lane adds its terms are those of the scalar path (ssr_accumulate_metrics<true>, element after element): same bits.
struct SsrSpecTgt2 { f2 t, tt, lt; };                  // two target bins: t, t * t, log10(t + 1e-12)

__device__ __forceinline__ SsrSpecTgt2 ssr_spec_tgt2(float t0, float t1) {
  const f2 t = f2_make(t0, t1);
  return {t, t * t, ssr_log10f_fast2(t + f2_splat(1e-12f))};
}
// target-only sums of two bins: tsum[0] += t^2, tsum[1] += log10(t)^2 (the scalar path's acc[2], acc[5])
__device__ __forceinline__ void ssr_spec_tgt_sums2(const SsrSpecTgt2& y, double* tsum) {
  const double t0 = (double)y.t.x, l0 = (double)y.lt.x, t1 = (double)y.t.y, l1 = (double)y.lt.y;
  tsum[0] += t0 * t0; tsum[1] += l0 * l0;
  tsum[0] += t1 * t1; tsum[1] += l1 * l1;
}
// two estimate bins against two target bins, full mask: acc[0] LSD term, acc[1] sum d^2, acc[3] sum d t, acc[4] sum l^2, acc[6] sum l lt
__device__ __forceinline__ void ssr_spec_est2(float e0, float e1, const SsrSpecTgt2& y, double* acc) {
  const f2 EPS2 = f2_splat(1e-12f);
  const f2 e = f2_make(e0, e1);
  const f2 ee = e + EPS2;
  const f2 d = ssr_log10f_fast2(ssr_divf_fast2(y.tt, ee * ee) + EPS2);
  const f2 dd = d * d;
  const f2 le = ssr_log10f_fast2(ee);
  {
    acc[0] += (double)dd.x;
    const double td = (double)y.t.x, d0 = (double)e.x - td;
    acc[1] += d0 * d0; acc[3] += d0 * td;
    const double ld = (double)y.lt.x, l0 = (double)le.x - ld;
    acc[4] += l0 * l0; acc[6] += l0 * ld;
  }
  {
    acc[0] += (double)dd.y;
    const double td = (double)y.t.y, d0 = (double)e.y - td;
    acc[1] += d0 * d0; acc[3] += d0 * td;
    const double ld = (double)y.lt.y, l0 = (double)le.y - ld;
    acc[4] += l0 * l0; acc[6] += l0 * ld;
  }
}

template <int KG>
__device__ __forceinline__ void ssr_specred_wave_body(const SsrSpecWaveParams& p, int chunk, int group_v) {
  const int lane = (int)threadIdx.x;
  const int key0 = (group_v / p.vi_n) * KG, item = group_v % p.vi_n;
  const int T = p.n_rows[item];
  const int t0 = chunk * p.rows_per_chunk;
  const int t1 = (t0 + p.rows_per_chunk < T) ? t0 + p.rows_per_chunk : T;
  const float* x = p.x + (int64_t)key0 * p.x_plane + p.frame_off[item] * (int64_t)p.pitch;
  const float* y = p.y + p.frame_off[item] * (int64_t)p.pitch;
  const int mask = p.metric_mask;
  const bool want_lsd = mask & SSR_M_LSD;
  constexpr int FULL = SSR_M_LSD | SSR_M_SISPEC | SSR_M_LOG_SISPEC;
  const bool fast = (mask & FULL) == FULL;          // wave-uniform
  double acc[KG][7];
  double lsd_sum[KG];
  double tsum[2] = {0.0, 0.0};                      // fast path: the target-only sums, shared by the KG keys
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    lsd_sum[g] = 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) acc[g][q] = 0.0;
  }
  const int nq = (p.F + 3) / 4;                      // quads per row (the last one may be partial)
  const int nfull = p.F / 4;                         // whole quads
  for (int t = t0; t < t1; ++t) {
    const float4* yr = reinterpret_cast<const float4*>(y + (int64_t)t * p.pitch);
#pragma unroll
    for (int g = 0; g < KG; ++g) acc[g][0] = 0.0;
    if (fast) {
      for (int q = lane; q < nfull; q += 64) {
        const float4 yv = yr[q];
        float4 xv[KG];
#pragma unroll
        for (int g = 0; g < KG; ++g) xv[g] = reinterpret_cast<const float4*>(x + (int64_t)g * p.x_plane + (int64_t)t * p.pitch)[q];
        const SsrSpecTgt2 y01 = ssr_spec_tgt2(yv.x, yv.y), y23 = ssr_spec_tgt2(yv.z, yv.w);
        ssr_spec_tgt_sums2(y01, tsum);
        ssr_spec_tgt_sums2(y23, tsum);
#pragma unroll
        for (int g = 0; g < KG; ++g) {
          ssr_spec_est2(xv[g].x, xv[g].y, y01, acc[g]);
          ssr_spec_est2(xv[g].z, xv[g].w, y23, acc[g]);
        }
      }
      if (nfull < nq && lane == nfull % 64) {        // the partial quad (1025 bins: one element), scalar
        const float4 yv = yr[nfull];
        const int k = 4 * nfull;
        double a7[7];
#pragma unroll
        for (int g = 0; g < KG; ++g) {
          const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)g * p.x_plane + (int64_t)t * p.pitch)[nfull];
          a7[0] = acc[g][0]; a7[1] = acc[g][1]; a7[2] = tsum[0]; a7[3] = acc[g][3]; a7[4] = acc[g][4]; a7[5] = tsum[1]; a7[6] = acc[g][6];
          ssr_accumulate_metrics<true>(xv.x, yv.x, mask, a7);
          if (k + 1 < p.F) ssr_accumulate_metrics<true>(xv.y, yv.y, mask, a7);
          if (k + 2 < p.F) ssr_accumulate_metrics<true>(xv.z, yv.z, mask, a7);
          acc[g][0] = a7[0]; acc[g][1] = a7[1]; acc[g][3] = a7[3]; acc[g][4] = a7[4]; acc[g][6] = a7[6];
          if (g == KG - 1) { tsum[0] = a7[2]; tsum[1] = a7[5]; }      // (every key adds the same target terms: keep one copy)
        }
      }
    } else {
      for (int q = lane; q < nq; q += 64) {
        const float4 yv = yr[q];
        const int k = 4 * q;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
          const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)g * p.x_plane + (int64_t)t * p.pitch)[q];
          ssr_accumulate_metrics<true>(xv.x, yv.x, mask, acc[g]);
          if (k + 1 < p.F) ssr_accumulate_metrics<true>(xv.y, yv.y, mask, acc[g]);
          if (k + 2 < p.F) ssr_accumulate_metrics<true>(xv.z, yv.z, mask, acc[g]);
          if (k + 3 < p.F) ssr_accumulate_metrics<true>(xv.w, yv.w, mask, acc[g]);
        }
      }
    }
    if (want_lsd) {
#pragma unroll
      for (int g = 0; g < KG; ++g) lsd_sum[g] += sqrt(ssr_wave_sum<64>(acc[g][0]) / (double)p.F);
    }
  }
  if (fast) {
#pragma unroll
    for (int g = 0; g < KG; ++g) { acc[g][2] = tsum[0]; acc[g][5] = tsum[1]; }
  }
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    double tot[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) tot[q] = ssr_wave_sum<64>(acc[g][1 + q]);
    if (lane == 0) {
      double* part = p.part + (((int64_t)(key0 + g) * p.vi_n + item) * p.n_chunks + chunk) * SSR_NPART;
      part[0] = lsd_sum[g];
#pragma unroll
      for (int q = 0; q < 6; ++q) part[1 + q] = tot[q];
      part[7] = 0.0;
    }
  }
}
