// libssrhip.so translation unit: degradation-side kernels and entry points - polyphase resampler (K7),
// zero-phase IIR (N1), cross-correlation alignment (N4).
#include "ssr_host.h"
#include "ssr_iir.h"
#include "ssr_xcorr.h"
#include "ssr_resample.h"
#include "ssr_resample_mfma.h"
#include "ssr_resample_rc.h"
#include "ssr_resample_chain.h"
#include "ssr_sinc.h"

__global__ __launch_bounds__(SSR_XC_NT) void k_xcorr(SsrXcorrParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  ssr_xcorr_body(p, blk, blockIdx.x % p.n_lag_blocks, blockIdx.x / p.n_lag_blocks, smem);
}

__global__ __launch_bounds__(64) void k_xcorr_pick(const double* best_val, const int64_t* best_idx, int n_lag_blocks,
                                                   int n_items, int64_t* argmax_out) {
  const int item = blockIdx.x * 64 + threadIdx.x;
  if (item < n_items) ssr_xcorr_pick(best_val, best_idx, n_lag_blocks, item, argmax_out);
}

#ifndef SSR_RESAMPLE_WPE
#define SSR_RESAMPLE_WPE 3
#endif
// float32 signals: 3 workgroups per CU (<= 168 VGPRs; the LDS footprint allows three) - float64 accumulators need the 256
template <typename S>
__global__ __launch_bounds__(SSR_RESAMPLE_NT, sizeof(S) == 4 ? SSR_RESAMPLE_WPE : 2) void k_resample(SsrResampleParamsT<S> p, int blocks_per_item, int total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  ssr_resample_persistent_body<S>(p, blk, (int)blockIdx.x, (int)gridDim.x, total, blocks_per_item, smem);
}

template <typename S>
__global__ __launch_bounds__(256) void k_resample_direct(SsrResampleParamsT<S> p, int blocks_per_item) {
  const int item = blockIdx.x / blocks_per_item;
  ssr_resample_direct_output<S>(p, item, (int64_t)(blockIdx.x % blocks_per_item) * 256 + threadIdx.x);
}

template <bool PAD>
__global__ __launch_bounds__(SSR_SINC_NT) void k_resample_sinc(SsrSincParams p, int blocks_per_item) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  ssr_sinc_block_body<PAD>(p, blk, blockIdx.x / blocks_per_item, blockIdx.x % blocks_per_item, smem);
}

__global__ __launch_bounds__(256) void k_sinc_table(SsrSincParams p) {
  ssr_sinc_table_body(p, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

template <typename X>
__global__ __launch_bounds__(64) void k_sosfiltfilt(SsrIirParamsT<X> p) {
  ssr_iir_wave_any<X>(p, blockIdx.x, threadIdx.x);
}

// K designs in one launch (ssr_sosfiltfilt_multi): workgroup -> (design, block of 64 / G utterances), G = the design's lanes per
// utterance (ssr_iir_group: 1, 2, 4 or 8); the design's parameters are wave-uniform values read from the kernel arguments.
constexpr int SSR_IIR_MAXD = 48;
struct SsrIirMultiParams {
  SsrIirParams base;                     // x, off, len, n_items; sos / zi / fwd / y = the first design's
  int n_designs;
  int64_t y_stride;                      // doubles between the designs' outputs
  int n_sections[SSR_IIR_MAXD], edge[SSR_IIR_MAXD];
  int wg_start[SSR_IIR_MAXD + 1];        // first workgroup of design d (exclusive prefix sum of ceil(n_items / (64 / G_d)))
  int64_t fwd_off[SSR_IIR_MAXD];         // doubles
};
__global__ __launch_bounds__(64) void k_sosfiltfilt_multi(SsrIirMultiParams mp) {
  int d = 0;
  while (d + 1 < mp.n_designs && (int)blockIdx.x >= mp.wg_start[d + 1]) ++d;
  const int wg = (int)blockIdx.x - mp.wg_start[d];
  SsrIirParams p = mp.base;
  p.sos += (int64_t)d * 8 * 6;
  p.zi += (int64_t)d * 8 * 2;
  p.n_sections = mp.n_sections[d];
  p.edge = mp.edge[d];
  p.fwd += mp.fwd_off[d];
  p.y += (int64_t)d * mp.y_stride;
  ssr_iir_wave_any<float>(p, wg, threadIdx.x);
}

// ----------------------------------------------------------------------------------------------------
static int64_t gcd64(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }

extern "C" int ssr_resample_plan(int64_t n_in, int up, int down, int* up_r, int* down_r, int64_t* n_out, int* half_len,
                                 int* n_pre_pad, int* n_pre_remove) {
  if (up < 1 || down < 1 || n_in < 0) return ssr_fail(SSR_ERR_INVALID_ARG, "up and down must be >= 1");
  const int g = (int)gcd64(up, down);
  up /= g; down /= g;
  const int64_t prod = n_in * up;
  const int mx = up > down ? up : down;
  const int hl = 10 * mx;
  const int pre_pad = down - hl % down;
  if (up_r) *up_r = up;
  if (down_r) *down_r = down;
  if (n_out) *n_out = prod / down + ((prod % down) ? 1 : 0);
  if (half_len) *half_len = hl;
  if (n_pre_pad) *n_pre_pad = pre_pad;
  if (n_pre_remove) *n_pre_remove = (hl + pre_pad) / down;
  return SSR_OK;
}

// Residue-class kernel (ssr_resample_rc.h): float32, 21 taps per phase (every up-sampling plan of resample_poly), 33 <= up <= 1024.
template <int HPP, int NTMAX> __global__ __launch_bounds__(NTMAX, 4) void k_resample_rc(SsrResampleRcParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ssr_resample_rc_body<HPP>(p, smem);
}
// lane groups per workgroup.  Measured (round 4, 160/147: 2 x 160 = 320 lanes, five full waves, against one group on three waves at
// 83 %): 1.74 against 1.71 ms per 4096 utterances - the kernel waits (barrier, LDS-DMA), it does not lack lanes, and a larger
// workgroup waits longer - so several groups are taken only where one group leaves more than a quarter of its lanes idle (up < 48
// or so: 33 residues on a 64-lane wave).
static int ssr_rc_groups(int up) {
  int best = 1;
  double best_eff = (double)up / (ssr_ceil_div(up, 64) * 64);
  if (best_eff >= 0.75) return 1;
  for (int g = 2; g * up <= 512 && g <= 8; ++g) {
    const double eff = (double)(g * up) / (ssr_ceil_div(g * up, 64) * 64);
    if (eff > best_eff + 0.02) { best_eff = eff; best = g; }
  }
  return best;
}
static bool resample_rc_eligible(int up, int down, int n_taps) {
#ifdef SSR_DEV_KNOBS
  static const int off = getenv("SSR_NO_RC") ? atoi(getenv("SSR_NO_RC")) : 0;
  if (off) return false;
#endif
  if (up < 33 || up > 1024 || (n_taps + up - 1) / up != 21) return false;
  return (size_t)4 * (ssr_rc_pairs(up, down, 21, ssr_rc_groups(up)) + 32) * sizeof(float) <= 64 * 1024;      // two stages of (x[i], x[i + down]) pairs
}
static int resample_rc_launch(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off, const int32_t* out_len,
                              int n_items, int max_out_len, int up, int down, const float* taps, int n_taps, int n_pre_remove, float* out,
                              hipStream_t s) {
  SsrResampleRcParams p{in, in_off, in_len, out_off, out_len, up, down, n_taps, n_pre_remove, taps, 1, 1, 0, 1, out};
  p.groups = ssr_rc_groups(up);
  // whole 32-pair deposits (a wave's LDS-DMA instruction) must stay inside a stage
  p.stage_floats = 2 * (((ssr_rc_pairs(up, down, 21, p.groups) + 31) / 32) * 32);
  const int steps = ssr_ceil_div(max_out_len, up), blocks = ssr_ceil_div(steps, SSR_RC_JB * p.groups);
  int n_chunks = ssr_ceil_div(4096, n_items);                   // >= ~4 k workgroups per launch, whole items where the batch is large
  if (n_chunks > blocks) n_chunks = blocks;
  if (n_chunks < 1) n_chunks = 1;
  p.blocks_per_chunk = ssr_ceil_div(blocks, n_chunks);
  p.n_chunks = ssr_ceil_div(blocks, p.blocks_per_chunk);
  const int64_t grid = (int64_t)n_items * p.n_chunks;
  if (grid > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  const size_t lds = (size_t)2 * p.stage_floats * sizeof(float);      // two stages
  const int nt = ssr_ceil_div(p.groups * up, 64) * 64;
  static thread_local SsrLdsSlot slot;
  if (nt <= 512) {                                              // (the launch bound sets the register budget: 8 waves -> 256 VGPRs)
    if (int rc = ssr_allow_lds((const void*)k_resample_rc<21, 512>, lds, &slot)) return rc;
    hipLaunchKernelGGL((k_resample_rc<21, 512>), dim3((unsigned)grid), dim3(nt), lds, s, p);
  } else {
    static thread_local SsrLdsSlot slot2;
    if (int rc = ssr_allow_lds((const void*)k_resample_rc<21, 1024>, lds, &slot2)) return rc;
    hipLaunchKernelGGL((k_resample_rc<21, 1024>), dim3((unsigned)grid), dim3(nt), lds, s, p);
  }
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// Two residue-class stages in one kernel, the intermediate signal in LDS (ssr_resample_chain.h).
template <int HPP> __global__ __launch_bounds__(SSR_RCC_NT, 6) void k_resample_chain(SsrResampleChainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ssr_resample_chain_body<HPP>(p, smem);
}
extern "C" int ssr_resample_poly_chain(const float* in, const int64_t* in_off, const int32_t* in_len, const int32_t* mid_len,
                                       const int64_t* out_off, const int32_t* out_len, int n_items, int max_out_len, int up1, int down1,
                                       const float* taps1, int n_taps1, int n_pre_remove1, int up2, int down2, const float* taps2,
                                       int n_taps2, int n_pre_remove2, float* out, void* stream) {
  if (!in || !in_off || !in_len || !mid_len || !out_off || !out_len || !taps1 || !taps2 || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (up1 < 1 || down1 < 1 || n_taps1 < 1 || up2 < 1 || down2 < 1 || n_taps2 < 1) return ssr_fail(SSR_ERR_INVALID_ARG, "bad resampling plan");
  if (n_items <= 0 || max_out_len <= 0) return SSR_OK;
  // the kernel forms 32-bit byte offsets into an item's output (and parks idle consumer lanes at byte 0x40000000): items of 2^28
  // samples or more take the two-call path (ssr_resample_poly rejects 2^29 itself)
  if (max_out_len >= (1 << 28)) return ssr_fail(SSR_ERR_UNSUPPORTED, "the fused chain handles outputs below 2^28 samples per item");
#ifdef SSR_DEV_KNOBS
  static const int off = getenv("SSR_NO_CHAIN") ? atoi(getenv("SSR_NO_CHAIN")) : 0;
  if (off) return ssr_fail(SSR_ERR_UNSUPPORTED, "fused chain switched off");
#endif
  // geometry of the fused kernel: 21 taps per phase in both plans, a block of 8 stage-1 steps = 24 stage-2 steps, 441 / 2 x 160 lanes
  const bool ok = (n_taps1 + up1 - 1) / up1 == 21 && (n_taps2 + up2 - 1) / up2 == 21 && up1 >= 33 && up1 <= SSR_RCC_NT1 &&
                  SSR_RCC_G2 * up2 <= SSR_RCC_NT2 && up2 >= 33 &&
                  (int64_t)SSR_RCC_JB1 * up1 == (int64_t)SSR_RCC_G2 * SSR_RCC_JB2 * down2;
  if (!ok) return ssr_fail(SSR_ERR_UNSUPPORTED, "the fused chain needs 21-tap phases and 8 up1 = 24 down2 (e.g. 441/160 then 160/147)");
  SsrResampleChainParams p{in, in_off, in_len, mid_len, out_off, out_len, up1, down1, n_taps1, n_pre_remove1, taps1,
                           up2, down2, n_taps2, n_pre_remove2, taps2, 1, 1, 0, 0, out};
  p.x_stage_floats = 2 * (((ssr_rc_pairs(up1, down1, 21, 1) + 31) / 32) * 32);
  p.y_pairs = ((SSR_RCC_JB1 * up1 + 20 + 7) / 8) * 8;
  const size_t lds = ((size_t)2 * p.x_stage_floats + (size_t)4 * p.y_pairs + 64) * sizeof(float);
  if (lds > 80 * 1024) return ssr_fail(SSR_ERR_UNSUPPORTED, "the fused chain's windows exceed half a CU's LDS");
  const int steps2 = ssr_ceil_div(max_out_len, up2), blocks = ssr_ceil_div(steps2, SSR_RCC_G2 * SSR_RCC_JB2);
  // whole items where the batch is large; a chunk pays two extra iterations (the block before its first, the pipeline's lag)
  int n_chunks = ssr_ceil_div(1024, n_items);
  if (n_chunks > blocks / 8) n_chunks = blocks / 8;
  if (n_chunks < 1) n_chunks = 1;
  p.blocks_per_chunk = ssr_ceil_div(blocks, n_chunks);
  p.n_chunks = ssr_ceil_div(blocks, p.blocks_per_chunk);
  const int64_t grid = (int64_t)n_items * p.n_chunks;
  if (grid > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  static thread_local SsrLdsSlot slot;
  if (int rc = ssr_allow_lds((const void*)k_resample_chain<21>, lds, &slot)) return rc;
  hipLaunchKernelGGL((k_resample_chain<21>), dim3((unsigned)grid), dim3(SSR_RCC_NT), lds, (hipStream_t)stream, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

template <typename S>
static int resample_poly_t(const S* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                           const int32_t* out_len, int n_items, int max_out_len, int up, int down, const S* taps,
                           int n_taps, int n_pre_remove, S* out, void* stream) {
  if (!in || !in_off || !in_len || !out_off || !out_len || !taps || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (up < 1 || down < 1 || n_taps < 1) return ssr_fail(SSR_ERR_INVALID_ARG, "bad resampling plan");
  if (n_items <= 0 || max_out_len <= 0) return SSR_OK;
  SsrResampleParamsT<S> p{in, in_off, in_len, out_off, out_len, up, down, taps, n_taps, n_pre_remove,
                          ssr_resample_pick_groups(up, down, n_taps, sizeof(S)), 1, out};
  if constexpr (sizeof(S) == 4) {
    if (resample_rc_eligible(up, down, n_taps))
      return resample_rc_launch((const float*)in, in_off, in_len, out_off, out_len, n_items, max_out_len, up, down, (const float*)taps, n_taps,
                                n_pre_remove, (float*)out, (hipStream_t)stream);
  }
  if (ssr_resample_lds_bytes(p) > 96 * 1024) p.taps_in_lds = 0;      // huge tap tables stay in HBM / L2
  const size_t lds = ssr_resample_lds_bytes(p);
  if (lds > 160 * 1024) {           // huge reduced `up`: the phase-blocked kernel's window does not fit LDS
    const int bpi = ssr_ceil_div(max_out_len, 256);
    if ((int64_t)n_items * bpi > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL((k_resample_direct<S>), dim3((unsigned)((int64_t)n_items * bpi)), dim3(256), 0, (hipStream_t)stream, p, bpi);
    HIP_TRY(hipGetLastError());
    return SSR_OK;
  }
  if ((size_t)ssr_resample_win(p) > (size_t)SSR_RESAMPLE_MAXPF * SSR_RESAMPLE_NT)
    return ssr_fail(SSR_ERR_UNSUPPORTED, "input window exceeds the prefetch registers");     // (geometry keeps it below)
  static thread_local SsrLdsSlot slot;
  if (int rc = ssr_allow_lds((const void*)k_resample<S>, lds, &slot)) return rc;
  const int bpi = ssr_ceil_div(max_out_len, ssr_resample_opb(p));
  const int64_t total = (int64_t)n_items * bpi;
  if (total > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  // persistent workgroups: as many as fit on the device at this LDS size (the tap table is staged once per workgroup)
  int dev = 0, n_cu = 256;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  int per_cu = (int)((160 * 1024) / (lds + 512));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > ssr_resample_max_wg_per_cu(sizeof(S))) per_cu = ssr_resample_max_wg_per_cu(sizeof(S));   // registers: launch bounds
  int64_t wgs = (int64_t)n_cu * per_cu;
  if (wgs > total) wgs = total;
  hipLaunchKernelGGL((k_resample<S>), dim3((unsigned)wgs), dim3(SSR_RESAMPLE_NT), lds, (hipStream_t)stream, p, bpi, (int)total);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_resample_poly(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                                 const int32_t* out_len, int n_items, int max_out_len, int up, int down,
                                 const float* taps, int n_taps, int n_pre_remove, float* out, void* stream) {
  return resample_poly_t<float>(in, in_off, in_len, out_off, out_len, n_items, max_out_len, up, down, taps, n_taps,
                                n_pre_remove, out, stream);
}

extern "C" int ssr_resample_poly_f64(const double* in, const int64_t* in_off, const int32_t* in_len,
                                     const int64_t* out_off, const int32_t* out_len, int n_items, int max_out_len, int up,
                                     int down, const double* taps, int n_taps, int n_pre_remove, double* out,
                                     void* stream) {
  return resample_poly_t<double>(in, in_off, in_len, out_off, out_len, n_items, max_out_len, up, down, taps, n_taps,
                                 n_pre_remove, out, stream);
}

// matrix-core variant (ssr_resample_mfma.h): float32 fused multiply-add evaluation of the same sums - NOT SciPy's bits
template <int MAXCH>
__global__ __launch_bounds__(SSR_RMF_NT, 2) void k_resample_mfma(SsrResampleMfmaParams p, int total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  ssr_resample_mfma_body<MAXCH>(p, blk, (int)blockIdx.x, (int)gridDim.x, total, smem);
}

extern "C" int ssr_resample_poly_mfma(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                                      const int32_t* out_len, int n_items, int max_out_len, int up, int down,
                                      const float* taps, int n_taps, int n_pre_remove, float* out, void* stream) {
  if (!in || !in_off || !in_len || !out_off || !out_len || !taps || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (up < 1 || down < 1 || n_taps < 1) return ssr_fail(SSR_ERR_INVALID_ARG, "bad resampling plan");
  if (n_items <= 0 || max_out_len <= 0) return SSR_OK;
  const SsrResampleMfmaGeom g = ssr_resample_mfma_geom(up, down, n_taps);
  if (!g.ok)       // (tap table or block window beyond the LDS / the prefetch registers: the exact kernel serves every plan)
    return ssr_fail(SSR_ERR_UNSUPPORTED, "this resampling plan does not fit the matrix-core kernel; use ssr_resample_poly");
  SsrResampleMfmaParams p{};
  p.rp = SsrResampleParams{in, in_off, in_len, out_off, out_len, up, down, taps, n_taps, n_pre_remove, 0, 1, out};
  p.n_items_total = n_items;
  p.max_out_len = max_out_len;
  p.n_groups = ssr_ceil_div(n_items, SSR_RMF_T);
  p.passes_per_group = ssr_ceil_div(max_out_len, 32 * g.NB);
  const int64_t total = (int64_t)p.n_groups * p.passes_per_group;
  if (total > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  const bool small = g.W <= 64 * 5;                       // window fits five 64-sample chunks per lane and utterance
  static thread_local SsrLdsSlot slot[2];
  if (int rc = ssr_allow_lds(small ? (const void*)k_resample_mfma<5> : (const void*)k_resample_mfma<SSR_RMF_MAXCH>, g.lds_bytes, &slot[small])) return rc;
  int dev = 0, n_cu = 256;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  int per_cu = (int)((160 * 1024) / (g.lds_bytes + 512));
  per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
  int64_t wgs = (int64_t)n_cu * per_cu;
  if (wgs > total) wgs = total;
  if (small) hipLaunchKernelGGL(k_resample_mfma<5>, dim3((unsigned)wgs), dim3(SSR_RMF_NT), g.lds_bytes, (hipStream_t)stream, p, (int)total);
  else hipLaunchKernelGGL(k_resample_mfma<SSR_RMF_MAXCH>, dim3((unsigned)wgs), dim3(SSR_RMF_NT), g.lds_bytes, (hipStream_t)stream, p, (int)total);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// ----------------------------------------------------------------------------------------------------
extern "C" int ssr_resample_sinc(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                                 const int32_t* out_len, int n_items, int max_out_len, const double* time_register,
                                 int64_t time_register_len, const double* interp_win, const double* interp_delta, int n_win,
                                 int num_table, int index_step, double scale, double ratio, int phase_period, float* out,
                                 void* stream) {
  if (!in || !in_off || !in_len || !out_off || !out_len || !time_register || !interp_win || !interp_delta || !out)
    return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_win < 1 || num_table < 1 || index_step < 1 || !(scale > 0.0) || scale > 1.0 || !(ratio > 0.0))
    return ssr_fail(SSR_ERR_INVALID_ARG, "bad interpolation filter description");
  if (n_items <= 0 || max_out_len <= 0) return SSR_OK;
  if (time_register_len < (int64_t)max_out_len)
    return ssr_fail(SSR_ERR_INVALID_ARG, "time_register holds fewer entries than max_out_len");
  if ((int64_t)n_win * 8 >= ((int64_t)1 << 31)) return ssr_fail(SSR_ERR_UNSUPPORTED, "interpolation table of 2 GiB or more");
  const SsrSincGeometry g = ssr_sinc_geometry(phase_period, ratio, n_win, index_step, 12288);   // <= 48 KB of input window
  SsrSincParams p{in, in_off, in_len, out_off, out_len, time_register, interp_win, interp_delta, n_win, num_table, index_step,
                  scale, out, g.period, g.pw, g.m, g.max_room, g.lds_floats};
  const int bpi = ssr_ceil_div(max_out_len, g.outputs_per_block);
  if ((int64_t)n_items * bpi > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  const size_t lds = (size_t)g.lds_floats * sizeof(float);
  if (lds > 160 * 1024) return ssr_fail(SSR_ERR_UNSUPPORTED, "input window of one block exceeds the LDS (extreme down-sampling ratio)");
  static thread_local SsrLdsSlot slot[2];
  if (int rc = ssr_allow_lds(g.pad ? (const void*)k_resample_sinc<true> : (const void*)k_resample_sinc<false>, lds, &slot[g.pad])) return rc;
  // the phase-major copy of the two tables (about the tables' own size: 0.5 MB for kaiser_best), stream-ordered scratch of this call
  p.tab_r = ssr_sinc_tab_r(g.max_room);
  p.tab_rows = index_step + 1;
  const int64_t entries = (int64_t)p.tab_rows * p.tab_r;
  if (entries * 16 >= ((int64_t)1 << 31)) return ssr_fail(SSR_ERR_UNSUPPORTED, "interpolation table of 2 GiB or more");
  hipStream_t s = (hipStream_t)stream;
  void* tab = nullptr;
  HIP_TRY(hipMallocAsync(&tab, (size_t)entries * 16, s));
  p.tab = (const double*)tab;
  hipLaunchKernelGGL(k_sinc_table, dim3((unsigned)ssr_ceil_div(entries, 256)), dim3(256), 0, s, p);
  if (g.pad) hipLaunchKernelGGL(k_resample_sinc<true>, dim3((unsigned)((int64_t)n_items * bpi)), dim3(SSR_SINC_NT), lds, s, p, bpi);
  else hipLaunchKernelGGL(k_resample_sinc<false>, dim3((unsigned)((int64_t)n_items * bpi)), dim3(SSR_SINC_NT), lds, s, p, bpi);
  const hipError_t launched = hipGetLastError();
  (void)hipFreeAsync(tab, s);
  HIP_TRY(launched);
  return SSR_OK;
}

// ----------------------------------------------------------------------------------------------------
static int xcorr_blocks(int max_len) { return max_len > 0 ? ssr_ceil_div(2 * (int64_t)max_len - 1, SSR_XC_LAGS) : 1; }

extern "C" size_t ssr_xcorr_workspace_bytes(int n_items, int max_len) {
  if (n_items <= 0) return 0;
  return 2 * ssr_align256((size_t)n_items * xcorr_blocks(max_len) * sizeof(double));
}

extern "C" int ssr_xcorr_argmax(const float* a, const int64_t* a_off, const float* b, const int64_t* b_off,
                                const int32_t* len, int n_items, int max_len, int64_t* argmax_out, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!a || !a_off || !b || !b_off || !len || !argmax_out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  if (max_len <= 0) return ssr_fail(SSR_ERR_INVALID_ARG, "empty signals");
  const int nb = xcorr_blocks(max_len);
  const size_t half = ssr_align256((size_t)n_items * nb * sizeof(double));
  if (!workspace || workspace_bytes < 2 * half) return ssr_fail(SSR_ERR_WORKSPACE, "workspace too small");
  if ((int64_t)n_items * nb > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  SsrXcorrParams p{a, a_off, b, b_off, len, nb, (double*)workspace, (int64_t*)((char*)workspace + half)};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_xcorr, dim3((unsigned)(n_items * nb)), dim3(SSR_XC_NT), SsrXcorrLds::bytes(), s, p);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_xcorr_pick, dim3(ssr_ceil_div(n_items, 64)), dim3(64), 0, s, p.best_val, p.best_idx, nb, n_items, argmax_out);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// ----------------------------------------------------------------------------------------------------
// workspace = [256 bytes of scratch (SsrIirParamsT::trash)] [forward-pass output of design 0] [of design 1] ...
static size_t iir_region_bytes(int64_t total_len, int n_items, int edge) {
  return ssr_align256(((size_t)total_len + (size_t)2 * edge * n_items) * sizeof(double));
}
extern "C" size_t ssr_sosfiltfilt_workspace_bytes(int64_t total_len, int n_items, int edge) {
  if (total_len <= 0 || n_items <= 0 || edge < 0) return 0;
  return 256 + iir_region_bytes(total_len, n_items, edge);
}

template <typename X>
static int sosfiltfilt_t(const X* x, const int64_t* off, const int32_t* len, int n_items, int64_t total_len,
                         const double* sos, const double* zi, int n_sections, int edge, double* y, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (!x || !off || !len || !sos || !zi || !y) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_sections < 1 || n_sections > 16) return ssr_fail(SSR_ERR_UNSUPPORTED, "n_sections must be in [1, 16]");
  if (edge < 0) return ssr_fail(SSR_ERR_INVALID_ARG, "negative edge");
  if (n_items <= 0) return SSR_OK;
  if (!workspace || workspace_bytes < ssr_sosfiltfilt_workspace_bytes(total_len, n_items, edge))
    return ssr_fail(SSR_ERR_WORKSPACE, "workspace too small");
  SsrIirParamsT<X> p{x, off, len, sos, zi, n_sections, edge, n_items, (double*)workspace + 32, y, (double*)workspace};
  hipStream_t s = (hipStream_t)stream;
  // G lanes per utterance = the smallest power of two >= n_sections, 64 / G utterances per one-wave workgroup (ssr_iir.h)
  const int per_wave = 64 / ssr_iir_group(n_sections);
  hipLaunchKernelGGL((k_sosfiltfilt<X>), dim3(ssr_ceil_div(n_items, per_wave)), dim3(64), 0, s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_sosfiltfilt(const float* x, const int64_t* off, const int32_t* len, int n_items, int64_t total_len,
                               const double* sos, const double* zi, int n_sections, int edge, double* y,
                               void* workspace, size_t workspace_bytes, void* stream) {
  return sosfiltfilt_t<float>(x, off, len, n_items, total_len, sos, zi, n_sections, edge, y, workspace, workspace_bytes, stream);
}

extern "C" size_t ssr_sosfiltfilt_multi_workspace_bytes(int64_t total_len, int n_items, const int32_t* edge, int n_designs) {
  if (total_len <= 0 || n_items <= 0 || !edge || n_designs <= 0) return 0;
  size_t b = 256;
  for (int d = 0; d < n_designs; ++d) b += iir_region_bytes(total_len, n_items, edge[d] < 0 ? 0 : edge[d]);
  return b;
}

extern "C" int ssr_sosfiltfilt_multi(const float* x, const int64_t* off, const int32_t* len, int n_items, int64_t total_len,
                                     const double* sos, const double* zi, const int32_t* n_sections, const int32_t* edge,
                                     int n_designs, double* y, int64_t y_stride, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  if (!x || !off || !len || !sos || !zi || !n_sections || !edge || !y) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_designs < 1 || n_designs > SSR_IIR_MAXD) return ssr_fail(SSR_ERR_UNSUPPORTED, "1 to 48 designs per call");
  if (y_stride < total_len) return ssr_fail(SSR_ERR_INVALID_ARG, "y_stride smaller than the batch");
  for (int d = 0; d < n_designs; ++d) {
    if (n_sections[d] < 1 || n_sections[d] > 8) return ssr_fail(SSR_ERR_UNSUPPORTED, "n_sections must be in [1, 8] (ssr_sosfiltfilt takes up to 16)");
    if (edge[d] < 0) return ssr_fail(SSR_ERR_INVALID_ARG, "negative edge");
  }
  if (n_items <= 0) return SSR_OK;
  if (!workspace || workspace_bytes < ssr_sosfiltfilt_multi_workspace_bytes(total_len, n_items, edge, n_designs))
    return ssr_fail(SSR_ERR_WORKSPACE, "workspace too small");
  SsrIirMultiParams mp;
  mp.base = SsrIirParams{x, off, len, sos, zi, 0, 0, n_items, (double*)workspace + 32, y, (double*)workspace};
  mp.n_designs = n_designs;
  mp.y_stride = y_stride;
  int64_t fo = 0, wgs = 0;
  for (int d = 0; d < n_designs; ++d) {
    mp.n_sections[d] = n_sections[d]; mp.edge[d] = edge[d]; mp.fwd_off[d] = fo; mp.wg_start[d] = (int)wgs;
    fo += (int64_t)(iir_region_bytes(total_len, n_items, edge[d]) / sizeof(double));
    wgs += ssr_ceil_div(n_items, 64 / ssr_iir_group(n_sections[d]));
    if (wgs > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  }
  for (int d = n_designs; d <= SSR_IIR_MAXD; ++d) mp.wg_start[d] = (int)wgs;
  hipLaunchKernelGGL(k_sosfiltfilt_multi, dim3((unsigned)wgs), dim3(64), 0, (hipStream_t)stream, mp);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_sosfiltfilt_f64(const double* x, const int64_t* off, const int32_t* len, int n_items, int64_t total_len,
                                   const double* sos, const double* zi, int n_sections, int edge, double* y,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  return sosfiltfilt_t<double>(x, off, len, n_items, total_len, sos, zi, n_sections, edge, y, workspace, workspace_bytes, stream);
}
