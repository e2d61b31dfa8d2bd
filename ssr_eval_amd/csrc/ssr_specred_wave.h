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

// KG keys per wave: the target row is fetched once for KG estimates (the kernel is HBM-bound: 2 KG images -> KG + 1), and the
// target-only float32 terms (t * t, log10(t + 1e-12)) are formed once per bin.
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
  double acc[KG][7];
  double lsd_sum[KG];
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    lsd_sum[g] = 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) acc[g][q] = 0.0;
  }
  const int nq = (p.F + 3) / 4;                      // quads per row (the last one may be partial)
  for (int t = t0; t < t1; ++t) {
    const float4* yr = reinterpret_cast<const float4*>(y + (int64_t)t * p.pitch);
#pragma unroll
    for (int g = 0; g < KG; ++g) acc[g][0] = 0.0;
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
    if (want_lsd) {
#pragma unroll
      for (int g = 0; g < KG; ++g) lsd_sum[g] += sqrt(ssr_wave_sum<64>(acc[g][0]) / (double)p.F);
    }
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
