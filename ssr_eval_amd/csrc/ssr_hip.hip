// libssrhip.so - HIP (gfx950) kernels + C ABI.  See include/ssr_hip.h for the contract.
// Kernel bodies live in the ssr_*.h headers (shared with the host emulation used by the CPU tests);
// this file only instantiates them as __global__ kernels, owns plans/tables and launches.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ssr_hip.h"
#include "ssr_lowpass.h"
#include "ssr_metrics.h"
#include "ssr_iir.h"
#include "ssr_xcorr.h"
#include "ssr_resample.h"
#include "ssr_stft_r3.h"
#include "ssr_tables.h"

#define SSR_VERSION 100

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(SSR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

extern "C" const char* ssr_last_error(void) { return g_err.c_str(); }
extern "C" int ssr_version(void) { return SSR_VERSION; }

// ----------------------------------------------------------------------------------------------------
// kernels
// Minimum waves per SIMD asked of the register allocator.  The 2048-point direct kernel without running
// SISpec sums fits 128 VGPRs with no spill, which admits a 4th workgroup per CU (-4.5 % time, measured);
// the variant that carries the six sums would spill at 128, so it stays at 3.
#ifndef SSR_MINW
#define SSR_MINW 4
#endif
#ifndef SSR_MINW_SUMS
#define SSR_MINW_SUMS 0
#endif
constexpr int ssr_stft_min_waves(int logn, bool blu, bool sums) { return (logn == 11 && !blu && (!sums || SSR_MINW_SUMS)) ? SSR_MINW : 1; }
// E64 (SSR_IN_EST64 / SSR_IN_BOTH64): float64 signals (pair mode only); their float64 epilogue needs more registers,
// so those variants are left to the allocator (min waves 1).
template <typename T, int LOGN, bool BLU, int MODE, bool SUMS, int E64>
__global__ __launch_bounds__((1 << LOGN) / ssr_stft_ppt(LOGN, BLU), E64 ? 1 : ssr_stft_min_waves(LOGN, BLU, SUMS))
void k_stft(SsrStftParams<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  const int item = blockIdx.x / p.n_chunks, chunk = blockIdx.x % p.n_chunks;
  ssr_stft_body<T, LOGN, BLU, MODE, SUMS, ssr_stft_ppt(LOGN, BLU), E64>(p, blk, chunk, item, smem);
}

template <typename T, int LOGN, int MODE, bool SUMS, int E64>
__global__ __launch_bounds__((1 << LOGN) / 8) void k_stft_r3(SsrStftParams<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  const int item = blockIdx.x / p.n_chunks, chunk = blockIdx.x % p.n_chunks;
  ssr_stft_r3_body<T, LOGN, MODE, SUMS, E64>(p, blk, chunk, item, smem);
}

#ifndef SSR_LOWPASS_WAVES_PER_EU
#define SSR_LOWPASS_WAVES_PER_EU 3   /* 168 VGPRs, no spill: 3 workgroups per CU instead of 2 */
#endif
template <typename T, int LOGN>
__global__ __launch_bounds__((1 << LOGN) / 8, SSR_LOWPASS_WAVES_PER_EU) void k_lowpass_frames(SsrLowpassParams<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  const int item = blockIdx.x / p.n_chunks, chunk = blockIdx.x % p.n_chunks;
  ssr_lowpass_frames_body<T, LOGN>(p, blk, chunk, item, smem);
}

#ifndef SSR_SSIM_WAVES_PER_EU
#define SSR_SSIM_WAVES_PER_EU 1
#endif
template <int CPT>
__global__ __launch_bounds__(SSR_SSIM_NT, SSR_SSIM_WAVES_PER_EU) void k_ssim(SsrSsimParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  const int tiles = p.n_row_tiles * p.n_strips;
  ssr_ssim_body<CPT>(p, blk, blockIdx.x % tiles, blockIdx.x / tiles, smem);
}

__global__ __launch_bounds__(256) void k_specred(SsrSpecRedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  ssr_specred_body(p, blk, blockIdx.x % p.n_chunks, blockIdx.x / p.n_chunks, smem);
}

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

__global__ __launch_bounds__(64) void k_finalize(SsrFinalizeParams p) {
  const int item = blockIdx.x * 64 + threadIdx.x;
  if (item < p.n_items) ssr_finalize_item(p, item);
}

__global__ __launch_bounds__(256) void k_ola(SsrOlaParams p, int blocks_per_item) {
  const int item = blockIdx.x / blocks_per_item;
  const int s = (blockIdx.x % blocks_per_item) * 256 + threadIdx.x;
  ssr_ola_sample(p, item, s);
}

template <typename S>
__global__ __launch_bounds__(SSR_RESAMPLE_NT) void k_resample(SsrResampleParamsT<S> p, int blocks_per_item) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  ssr_resample_body<S>(p, blk, blockIdx.x % blocks_per_item, blockIdx.x / blocks_per_item, smem);
}

template <int G, typename X>
__global__ __launch_bounds__(64) void k_sosfiltfilt(SsrIirParamsT<X> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ssr_iir_wave<G, X>(p, blockIdx.x, threadIdx.x, smem);
}

__global__ __launch_bounds__(256) void k_magphase(const float* re, const float* im, int64_t n, float eps, float* mag,
                                                  float* cosv, float* sinv) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float r = re[i], q = im[i];
    float p = r * r + q * q;          // torch: clamp(real**2 + imag**2, eps, inf) ** 0.5   (dsp.py:78)
    p = p < eps ? eps : p;
    const float m = sqrtf(p);
    mag[i] = m;
    cosv[i] = r / m;
    sinv[i] = q / m;
  }
}

// ----------------------------------------------------------------------------------------------------
// plan
template <typename T> struct DevTables {
  T* window = nullptr;
  T* window_h = nullptr;
  cx<T>*tw = nullptr, *wchirp = nullptr, *bfilt = nullptr, *chirp = nullptr;
};

struct ssr_plan {
  int n_fft, hop, n_bins, precision, device;
  SsrEngine eng;
  DevTables<float> f32;
  DevTables<double> f64;
  double* window64 = nullptr;  // always present (OLA normalisation)
  std::vector<void*> allocs;
};

template <typename V> static int upload(ssr_plan* pl, const std::vector<V>& h, V** d) {
  *d = nullptr;
  if (h.empty()) return SSR_OK;
  HIP_TRY(hipMalloc((void**)d, h.size() * sizeof(V)));
  pl->allocs.push_back((void*)*d);
  HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(V), hipMemcpyHostToDevice));
  return SSR_OK;
}

template <typename T> static int build_dev_tables(ssr_plan* pl, DevTables<T>& d) {
  SsrTables<T> t;
  if (!ssr_build_tables<T>(pl->n_fft, t)) return fail(SSR_ERR_UNSUPPORTED, "unsupported n_fft");
  int rc;
  if ((rc = upload(pl, t.window, &d.window))) return rc;
  if ((rc = upload(pl, t.window_h, &d.window_h))) return rc;
  if ((rc = upload(pl, t.tw, &d.tw))) return rc;
  if ((rc = upload(pl, t.wchirp, &d.wchirp))) return rc;
  if ((rc = upload(pl, t.bfilt, &d.bfilt))) return rc;
  if ((rc = upload(pl, t.chirp, &d.chirp))) return rc;
  return SSR_OK;
}

template <typename T> static const DevTables<T>& tables_of(const ssr_plan* pl);
template <> const DevTables<float>& tables_of<float>(const ssr_plan* pl) { return pl->f32; }
template <> const DevTables<double>& tables_of<double>(const ssr_plan* pl) { return pl->f64; }

// kernel registry: (precision, logn, bluestein) -> launcher
typedef int (*stft_launcher)(const ssr_plan*, void* params, int grid, hipStream_t);

template <typename T, int LOGN, bool BLU, int MODE, bool SUMS, int E64 = 0>
static int launch_stft_mode(SsrStftParams<T>& p, int grid, hipStream_t s) {
  // SSR_LDS_PAD (bytes, developer knob): over-allocate LDS to cap workgroups per CU in occupancy experiments
  static const size_t lds_pad = getenv("SSR_LDS_PAD") ? (size_t)atol(getenv("SSR_LDS_PAD")) : 0;
  constexpr int PPT = ssr_stft_ppt(LOGN, BLU);
  const size_t lds = SsrStftLds<T, LOGN, PPT>::bytes() + lds_pad;
  static thread_local int attr_dev = -1;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (lds > 48 * 1024 && attr_dev != dev) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_stft<T, LOGN, BLU, MODE, SUMS, E64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_dev = dev;
  }
  hipLaunchKernelGGL((k_stft<T, LOGN, BLU, MODE, SUMS, E64>), dim3(grid), dim3((1 << LOGN) / PPT), lds, s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}
template <typename T, int LOGN, bool BLU> static int launch_stft_inst(SsrStftParams<T>& p, int grid, hipStream_t s) {
  if (p.mode != SSR_MODE_PAIR) return launch_stft_mode<T, LOGN, BLU, SSR_MODE_SINGLE, false>(p, grid, s);
  const bool sums = p.metric_mask & (SSR_M_SISPEC | SSR_M_LOG_SISPEC);
  if (p.a64 && p.b64)
    return sums ? launch_stft_mode<T, LOGN, BLU, SSR_MODE_PAIR, true, SSR_IN_BOTH64>(p, grid, s)
                : launch_stft_mode<T, LOGN, BLU, SSR_MODE_PAIR, false, SSR_IN_BOTH64>(p, grid, s);
  if (p.a64)
    return sums ? launch_stft_mode<T, LOGN, BLU, SSR_MODE_PAIR, true, SSR_IN_EST64>(p, grid, s)
                : launch_stft_mode<T, LOGN, BLU, SSR_MODE_PAIR, false, SSR_IN_EST64>(p, grid, s);
  return sums ? launch_stft_mode<T, LOGN, BLU, SSR_MODE_PAIR, true>(p, grid, s)
              : launch_stft_mode<T, LOGN, BLU, SSR_MODE_PAIR, false>(p, grid, s);
}

template <typename T, int LOGN, int MODE, bool SUMS, int E64 = 0>
static int launch_stft_r3_mode(SsrStftParams<T>& p, int grid, hipStream_t s) {
  const size_t lds = SsrStftR3Lds<T, LOGN>::bytes(p.n_fft / 3);
  HIP_TRY(hipFuncSetAttribute((const void*)k_stft_r3<T, LOGN, MODE, SUMS, E64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_stft_r3<T, LOGN, MODE, SUMS, E64>), dim3(grid), dim3((1 << LOGN) / 8), lds, s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}
template <typename T, int LOGN> static int launch_stft_r3(SsrStftParams<T>& p, int grid, hipStream_t s) {
  if (p.mode != SSR_MODE_PAIR) return launch_stft_r3_mode<T, LOGN, SSR_MODE_SINGLE, false>(p, grid, s);
  const bool sums = p.metric_mask & (SSR_M_SISPEC | SSR_M_LOG_SISPEC);
  if (p.a64 && p.b64)
    return sums ? launch_stft_r3_mode<T, LOGN, SSR_MODE_PAIR, true, SSR_IN_BOTH64>(p, grid, s)
                : launch_stft_r3_mode<T, LOGN, SSR_MODE_PAIR, false, SSR_IN_BOTH64>(p, grid, s);
  if (p.a64)
    return sums ? launch_stft_r3_mode<T, LOGN, SSR_MODE_PAIR, true, SSR_IN_EST64>(p, grid, s)
                : launch_stft_r3_mode<T, LOGN, SSR_MODE_PAIR, false, SSR_IN_EST64>(p, grid, s);
  return sums ? launch_stft_r3_mode<T, LOGN, SSR_MODE_PAIR, true>(p, grid, s)
              : launch_stft_r3_mode<T, LOGN, SSR_MODE_PAIR, false>(p, grid, s);
}

template <typename T> static int launch_stft_t(const ssr_plan* pl, SsrStftParams<T>& p, int grid, hipStream_t s) {
  const DevTables<T>& d = tables_of<T>(pl);
  p.window = d.window_h; p.tw = d.tw; p.wchirp = d.wchirp; p.bfilt = d.bfilt; p.chirp = d.chirp;
  if (pl->eng.radix == 3) {
    switch (pl->eng.logn) {
      case 8: return launch_stft_r3<T, 8>(p, grid, s);
      case 9: return launch_stft_r3<T, 9>(p, grid, s);
      case 10: return launch_stft_r3<T, 10>(p, grid, s);
      case 11: return launch_stft_r3<T, 11>(p, grid, s);
    }
    return fail(SSR_ERR_UNSUPPORTED, "no radix-3 kernel for this FFT length");
  }
#define CASE(L)                                                                   \
  case L:                                                                         \
    return pl->eng.bluestein ? launch_stft_inst<T, L, true>(p, grid, s) : launch_stft_inst<T, L, false>(p, grid, s);
  switch (pl->eng.logn) { CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
    case 13: return launch_stft_inst<T, 13, true>(p, grid, s);
  }
#undef CASE
  return fail(SSR_ERR_UNSUPPORTED, "no kernel for this FFT length");
}

template <typename T, int LOGN> static int launch_lowpass_inst(SsrLowpassParams<T>& p, int grid, hipStream_t s) {
  const size_t lds = SsrStftLds<T, LOGN>::bytes();
  static thread_local int attr_dev = -1;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (lds > 48 * 1024 && attr_dev != dev) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_lowpass_frames<T, LOGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_dev = dev;
  }
  hipLaunchKernelGGL((k_lowpass_frames<T, LOGN>), dim3(grid), dim3((1 << LOGN) / 8), lds, s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

template <typename T> static int launch_lowpass_t(const ssr_plan* pl, SsrLowpassParams<T>& p, int grid, hipStream_t s) {
  const DevTables<T>& d = tables_of<T>(pl);
  p.window = d.window; p.tw = d.tw;
  if (pl->eng.bluestein) return fail(SSR_ERR_UNSUPPORTED, "inverse STFT needs a power-of-two n_fft in [256, 4096]");
  switch (pl->eng.logn) {
    case 8: return launch_lowpass_inst<T, 8>(p, grid, s);
    case 9: return launch_lowpass_inst<T, 9>(p, grid, s);
    case 10: return launch_lowpass_inst<T, 10>(p, grid, s);
    case 11: return launch_lowpass_inst<T, 11>(p, grid, s);
    case 12: return launch_lowpass_inst<T, 12>(p, grid, s);
  }
  return fail(SSR_ERR_UNSUPPORTED, "no kernel for this FFT length");
}

extern "C" int ssr_plan_create(int n_fft, int hop, int precision, ssr_plan** out) {
  if (!out) return fail(SSR_ERR_INVALID_ARG, "plan output pointer is null");
  *out = nullptr;
  if (n_fft < 2 || hop < 1) return fail(SSR_ERR_INVALID_ARG, "n_fft must be >= 2 and hop >= 1");
  if (precision != SSR_F32 && precision != SSR_F64) return fail(SSR_ERR_INVALID_ARG, "precision must be SSR_F32 or SSR_F64");
  SsrEngine eng = ssr_pick_engine(n_fft);
  if (!eng.ok) return fail(SSR_ERR_UNSUPPORTED, "n_fft too large: Bluestein length would exceed 8192 (n_fft <= 4096)");
  ssr_plan* pl = new ssr_plan();
  pl->n_fft = n_fft; pl->hop = hop; pl->n_bins = n_fft / 2 + 1; pl->precision = precision; pl->eng = eng;
  int rc = SSR_OK;
  if (hipGetDevice(&pl->device) != hipSuccess) rc = fail(SSR_ERR_HIP, "hipGetDevice failed (no HIP device?)");
  if (!rc) rc = (precision == SSR_F64) ? build_dev_tables<double>(pl, pl->f64) : build_dev_tables<float>(pl, pl->f32);
  if (!rc) {
    SsrTables<double> t;
    ssr_build_tables<double>(n_fft, t);
    rc = upload(pl, t.window, &pl->window64);
  }
  if (rc) { ssr_plan_destroy(pl); return rc; }
  *out = pl;
  return SSR_OK;
}

extern "C" int ssr_plan_destroy(ssr_plan* pl) {
  if (!pl) return SSR_OK;
  for (void* p : pl->allocs) (void)hipFree(p);
  delete pl;
  return SSR_OK;
}

extern "C" int ssr_plan_query(const ssr_plan* pl, int* n_fft, int* hop, int* n_bins, int* fft_len, int* bluestein,
                              int* precision) {
  if (!pl) return fail(SSR_ERR_INVALID_ARG, "plan is null");
  if (n_fft) *n_fft = pl->n_fft;
  if (hop) *hop = pl->hop;
  if (n_bins) *n_bins = pl->n_bins;
  if (fft_len) *fft_len = 1 << pl->eng.logn;
  if (bluestein) *bluestein = pl->eng.bluestein ? 1 : 0;
  if (precision) *precision = pl->precision;
  return SSR_OK;
}

extern "C" int64_t ssr_num_frames(const ssr_plan* pl, int64_t n) {
  if (!pl) return -1;
  return 1 + (n + 2 * (int64_t)(pl->n_fft / 2) - pl->n_fft) / pl->hop;
}

// ----------------------------------------------------------------------------------------------------
// launch geometry (deterministic functions of the batch shape; also define the workspace layout)
static int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// workgroups per launch aimed for when chunking items (developer knob: SSR_TARGET_WGS)
static const int TARGET_WGS = getenv("SSR_TARGET_WGS") ? atoi(getenv("SSR_TARGET_WGS")) : 4096;

static int units_per_chunk_for(int max_units, int n_items) {
  int64_t u = ((int64_t)max_units * n_items + TARGET_WGS - 1) / TARGET_WGS;
  if (u < 4) u = 4;
  if (u > 128) u = 128;             // ragged batches: short workgroups keep the tail of a launch balanced
  if (u > max_units) u = max_units;
  if (u < 1) u = 1;
  return (int)u;
}

struct SsimGeom { int rows_per_tile, n_row_tiles, n_strips, cpt; };
static SsimGeom ssim_geom(int max_rows, int n_bins, int n_items) {
  SsimGeom g;
  const int out_rows = max_rows - 6 > 1 ? max_rows - 6 : 1;
  int64_t r = ((int64_t)out_rows * n_items + TARGET_WGS - 1) / TARGET_WGS;
  if (r < 8) r = 8;
  if (r > 128) r = 128;
  if (r > out_rows) r = out_rows;
  g.rows_per_tile = (int)r;
  g.n_row_tiles = ceil_div(out_rows, g.rows_per_tile);
  g.cpt = ssr_ssim_pick_cpt(n_bins);
  static const int cpt_env = getenv("SSR_SSIM_CPT") ? atoi(getenv("SSR_SSIM_CPT")) : 0;   // developer knob
  if (cpt_env >= 1 && cpt_env <= SSR_SSIM_MAXCPT) g.cpt = cpt_env;
  g.n_strips = n_bins > 6 ? ceil_div(n_bins - 6, ssr_ssim_strip_out(g.cpt)) : 1;
  return g;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct PairWs {
  size_t off_est, off_tgt, off_part, off_ssim, total;
  int units_per_chunk, n_chunks;
  SsimGeom sg;
};
static PairWs pair_ws(const ssr_plan* pl, int n_items, int max_len, int64_t total_rows) {
  PairWs w;
  const int max_T = (int)ssr_num_frames(pl, max_len);
  w.units_per_chunk = units_per_chunk_for(max_T, n_items);
  w.n_chunks = ceil_div(max_T, w.units_per_chunk);
  w.sg = ssim_geom(max_T, pl->n_bins, n_items);
  size_t o = 0;
  w.off_est = o; o += align256((size_t)total_rows * pl->n_bins * sizeof(float));
  w.off_tgt = o; o += align256((size_t)total_rows * pl->n_bins * sizeof(float));
  w.off_part = o; o += align256((size_t)n_items * w.n_chunks * SSR_NPART * sizeof(double));
  w.off_ssim = o; o += align256((size_t)n_items * w.sg.n_row_tiles * w.sg.n_strips * sizeof(double));
  w.total = o;
  return w;
}

// ----------------------------------------------------------------------------------------------------
template <typename T>
static int stft_single_t(const ssr_plan* pl, const float* wav, const int64_t* off, const int32_t* len,
                         const int64_t* frame_off, int n_items, int max_len, int out_kind, float* out_a, float* out_b,
                         hipStream_t s) {
  SsrStftParams<T> p{};
  p.a = wav; p.b = wav; p.a_off = off; p.b_off = off; p.len = len; p.frame_off = frame_off;
  p.mode = SSR_MODE_SINGLE; p.out_kind = out_kind; p.metric_mask = 0;
  p.n_fft = pl->n_fft; p.hop = pl->hop; p.n_bins = pl->n_bins;
  const int max_units = (int)((ssr_num_frames(pl, max_len) + 1) / 2);
  p.units_per_chunk = units_per_chunk_for(max_units, n_items);
  p.n_chunks = ceil_div(max_units, p.units_per_chunk);
  p.out_a = out_a; p.out_b = out_b; p.part = nullptr;
  return launch_stft_t<T>(pl, p, n_items * p.n_chunks, s);
}

extern "C" int ssr_stft(const ssr_plan* pl, const float* wav, const int64_t* wav_off, const int32_t* wav_len,
                        const int64_t* frame_off, int n_items, int max_len, int out_kind, float* out_a, float* out_b,
                        void* stream) {
  if (!pl || !wav || !wav_off || !wav_len || !frame_off || !out_a) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (out_kind != SSR_STFT_MAG && out_kind != SSR_STFT_COMPLEX) return fail(SSR_ERR_INVALID_ARG, "bad out_kind");
  if (out_kind == SSR_STFT_COMPLEX && !out_b) return fail(SSR_ERR_INVALID_ARG, "complex output needs out_b");
  if (n_items <= 0) return SSR_OK;
  if (max_len < 1) return fail(SSR_ERR_INVALID_ARG, "empty signals");
  if (max_len >= (1 << 29)) return fail(SSR_ERR_UNSUPPORTED, "signals of 2^29 samples or more (4 GiB buffer views)");
  hipStream_t s = (hipStream_t)stream;
  return pl->precision == SSR_F64
             ? stft_single_t<double>(pl, wav, wav_off, wav_len, frame_off, n_items, max_len, out_kind, out_a, out_b, s)
             : stft_single_t<float>(pl, wav, wav_off, wav_len, frame_off, n_items, max_len, out_kind, out_a, out_b, s);
}

extern "C" int ssr_magphase(const float* re, const float* im, int64_t n, float eps, float* mag, float* cosv,
                            float* sinv, void* stream) {
  if (!re || !im || !mag || !cosv || !sinv) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n <= 0) return SSR_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_magphase, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, re, im, n, eps, mag, cosv, sinv);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// ----------------------------------------------------------------------------------------------------
// A6: the tensor helpers of ssr_eval/utils.py as stand-alone calls (AudioMetrics.sispec has them fused in-kernel).
enum { SSR_EW_TO_LOG = 0, SSR_EW_FROM_LOG = 1 };
__global__ __launch_bounds__(256) void k_elementwise(int op, const float* x, int64_t n, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    if (op == SSR_EW_TO_LOG) out[i] = log10f(v + 1e-12f);               // utils.py:43-44
    else out[i] = powf(10.0f, v > 5.0f ? 5.0f : v);                     // utils.py:47-49 (clip(max=5), NaN passes through)
  }
}

// sums[item] = {sum a^2, sum b^2, sum a*b} over per_item contiguous elements; one workgroup per item, float64
// accumulation in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_energy_sums(const float* a, const float* b, int64_t per_item, double* sums) {
  __shared__ double sh[3][4];
  const float* pa = a + (int64_t)blockIdx.x * per_item;
  const float* pb = b + (int64_t)blockIdx.x * per_item;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int64_t i = threadIdx.x; i < per_item; i += 256) {
    const double u = (double)pa[i], v = (double)pb[i];
    s0 += u * u; s1 += v * v; s2 += u * v;
  }
  s0 = ssr_wave_sum<64>(s0); s1 = ssr_wave_sum<64>(s1); s2 = ssr_wave_sum<64>(s2);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; sh[2][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) sums[(int64_t)blockIdx.x * 3 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// out[item][j] = (x[item][j] * mul[item]) / div[item], two float32 roundings as in energy_unify (utils.py:79-82)
__global__ __launch_bounds__(256) void k_scale_items(const float* x, const float* mul, const float* div, int64_t per_item,
                                                     int64_t n, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t item = i / per_item;
    out[i] = ssr_fmul_rn(x[i], mul[item]) / div[item];
  }
}

static unsigned ew_blocks(int64_t n) {
  int64_t blocks = (n + 255) / 256;
  return (unsigned)(blocks > 8192 ? 8192 : blocks);
}

extern "C" int ssr_to_log(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n <= 0) return SSR_OK;
  hipLaunchKernelGGL(k_elementwise, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (int)SSR_EW_TO_LOG, x, n, out);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_from_log(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n <= 0) return SSR_OK;
  hipLaunchKernelGGL(k_elementwise, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (int)SSR_EW_FROM_LOG, x, n, out);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_energy_sums(const float* a, const float* b, int n_items, int64_t per_item, double* sums, void* stream) {
  if (!a || !b || !sums) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  if (per_item < 0) return fail(SSR_ERR_INVALID_ARG, "negative item size");
  hipLaunchKernelGGL(k_energy_sums, dim3((unsigned)n_items), dim3(256), 0, (hipStream_t)stream, a, b, per_item, sums);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_scale_items(const float* x, const float* mul, const float* div, int n_items, int64_t per_item,
                               float* out, void* stream) {
  if (!x || !mul || !div || !out) return fail(SSR_ERR_INVALID_ARG, "null argument");
  const int64_t n = (int64_t)n_items * per_item;
  if (n <= 0) return SSR_OK;
  hipLaunchKernelGGL(k_scale_items, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, mul, div, per_item, n, out);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// ----------------------------------------------------------------------------------------------------
extern "C" size_t ssr_pair_metrics_workspace_bytes(const ssr_plan* pl, int n_items, int max_len, int64_t total_rows) {
  if (!pl || n_items <= 0) return 0;
  return pair_ws(pl, n_items, max_len, total_rows).total + align256((size_t)n_items * sizeof(int32_t));
}

template <int CPT> static int launch_ssim_inst(const SsrSsimParams& p, int grid, hipStream_t s) {
  const size_t lds = SsrSsimLds<CPT>::bytes();
  static thread_local int attr_dev = -1;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (lds > 48 * 1024 && attr_dev != dev) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_ssim<CPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_dev = dev;
  }
  hipLaunchKernelGGL((k_ssim<CPT>), dim3(grid), dim3(SSR_SSIM_NT), lds, s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

static int launch_ssim(const float* x, const float* y, const int64_t* frame_off, const int32_t* n_rows, int n_items,
                       int F, const SsimGeom& g, double* part, hipStream_t s) {
  SsrSsimParams p{x, y, frame_off, n_rows, F, g.rows_per_tile, g.n_row_tiles, g.n_strips, part};
  const int grid = n_items * g.n_row_tiles * g.n_strips;
  switch (g.cpt) {
    case 1: return launch_ssim_inst<1>(p, grid, s);
    case 2: return launch_ssim_inst<2>(p, grid, s);
    case 3: return launch_ssim_inst<3>(p, grid, s);
    case 4: return launch_ssim_inst<4>(p, grid, s);
    case 5: return launch_ssim_inst<5>(p, grid, s);
    case 6: return launch_ssim_inst<6>(p, grid, s);
  }
  return fail(SSR_ERR_UNSUPPORTED, "bad SSIM geometry");
}

static int launch_finalize(const double* part, int n_chunks, const double* ssim_part, int n_tiles, const int32_t* n_rows,
                           int F, unsigned mask, int n_items, double* out, hipStream_t s) {
  SsrFinalizeParams p{part, n_chunks, ssim_part, n_tiles, n_rows, F, (int)mask, n_items, out};
  hipLaunchKernelGGL(k_finalize, dim3(ceil_div(n_items, 64)), dim3(64), 0, s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// n_rows (T_i) for the finalisation is derived on device from len: a tiny kernel fills it.
__global__ void k_rows_from_len(const int32_t* len, int n_items, int n_fft, int hop, int32_t* rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_items) rows[i] = ssr_num_frames_dev(len[i], n_fft, hop);
}

template <typename T>
static int pair_stage_stft(const ssr_plan* pl, const float* est, const double* est64, const int64_t* est_off, const float* tgt, const double* tgt64,
                           const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off, int n_items,
                           unsigned mask, bool need_mag, const PairWs& w, char* ws, hipStream_t s) {
  SsrStftParams<T> p{};
  p.a = est; p.a64 = est64; p.b = tgt; p.b64 = tgt64; p.a_off = est_off; p.b_off = tgt_off; p.len = len; p.frame_off = frame_off;
  p.mode = SSR_MODE_PAIR; p.out_kind = need_mag ? SSR_OUT_MAG : SSR_OUT_NONE; p.metric_mask = (int)mask;
  p.n_fft = pl->n_fft; p.hop = pl->hop; p.n_bins = pl->n_bins;
  p.units_per_chunk = w.units_per_chunk; p.n_chunks = w.n_chunks;
  p.out_a = (float*)(ws + w.off_est); p.out_b = (float*)(ws + w.off_tgt);
  p.part = (double*)(ws + w.off_part);
  return launch_stft_t<T>(pl, p, n_items * w.n_chunks, s);
}

// stages: 1 = STFT + LSD/SISpec epilogue, 2 = SSIM, 4 = finalise (bench.py times stages separately)
static int pair_metrics_impl(const ssr_plan* pl, const float* est, const double* est64, const int64_t* est_off,
                             const float* tgt, const double* tgt64, const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off,
                             int n_items, int max_len, int64_t total_rows, unsigned mask, double* out,
                             void* workspace, size_t workspace_bytes, void* stream, int stages) {
  if (!pl || (!est && !est64) || (!tgt && !tgt64) || !est_off || !tgt_off || !len || !frame_off || !out)
    return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  if (max_len < 1) return fail(SSR_ERR_INVALID_ARG, "empty signals");
  if (max_len >= (1 << 29)) return fail(SSR_ERR_UNSUPPORTED, "signals of 2^29 samples or more (4 GiB buffer views)");
  if ((mask & ~SSR_METRIC_ALL) || mask == 0) return fail(SSR_ERR_INVALID_ARG, "bad metric mask");
  const int max_T = (int)ssr_num_frames(pl, max_len);
  const bool want_ssim = mask & SSR_METRIC_SSIM;
  if (want_ssim && (int64_t)max_T * pl->n_bins >= ((int64_t)1 << 30))
    return fail(SSR_ERR_UNSUPPORTED, "spectrogram of 2^30 elements or more (4 GiB buffer views)");
  if (want_ssim && (max_T < 7 || pl->n_bins < 7)) return fail(SSR_ERR_INVALID_ARG, "win_size exceeds image extent");
  const PairWs w = pair_ws(pl, n_items, max_len, total_rows);
  // rows array lives at the tail of the ssim partial area's alignment slack: allocate it explicitly
  const size_t rows_bytes = align256((size_t)n_items * sizeof(int32_t));
  if (!workspace || workspace_bytes < w.total + rows_bytes) return fail(SSR_ERR_WORKSPACE, "workspace too small");
  char* ws = (char*)workspace;
  int32_t* rows = (int32_t*)(ws + w.total);
  hipStream_t s = (hipStream_t)stream;
  int rc = SSR_OK;
  if (stages & 1) {
    hipLaunchKernelGGL(k_rows_from_len, dim3(ceil_div(n_items, 256)), dim3(256), 0, s, len, n_items, pl->n_fft, pl->hop, rows);
    HIP_TRY(hipGetLastError());
    rc = pl->precision == SSR_F64
             ? pair_stage_stft<double>(pl, est, est64, est_off, tgt, tgt64, tgt_off, len, frame_off, n_items, mask, want_ssim, w, ws, s)
             : pair_stage_stft<float>(pl, est, est64, est_off, tgt, tgt64, tgt_off, len, frame_off, n_items, mask, want_ssim, w, ws, s);
    if (rc) return rc;
  }
  if ((stages & 2) && want_ssim) {
    rc = launch_ssim((const float*)(ws + w.off_est), (const float*)(ws + w.off_tgt), frame_off, rows, n_items,
                     pl->n_bins, w.sg, (double*)(ws + w.off_ssim), s);
    if (rc) return rc;
  }
  if (stages & 4) {
    rc = launch_finalize((const double*)(ws + w.off_part), w.n_chunks, want_ssim ? (const double*)(ws + w.off_ssim) : nullptr,
                         w.sg.n_row_tiles * w.sg.n_strips, rows, pl->n_bins, mask, n_items, out, s);
  }
  return rc;
}

extern "C" int ssr_pair_metrics_stages(const ssr_plan* pl, const float* est, const int64_t* est_off, const float* tgt,
                                       const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off,
                                       int n_items, int max_len, int64_t total_rows, unsigned mask, double* out,
                                       void* workspace, size_t workspace_bytes, void* stream, int stages) {
  return pair_metrics_impl(pl, est, nullptr, est_off, tgt, nullptr, tgt_off, len, frame_off, n_items, max_len, total_rows,
                           mask, out, workspace, workspace_bytes, stream, stages);
}

extern "C" int ssr_pair_metrics(const ssr_plan* pl, const float* est, const int64_t* est_off, const float* tgt,
                                const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off, int n_items,
                                int max_len, int64_t total_rows, unsigned mask, double* out, void* workspace,
                                size_t workspace_bytes, void* stream) {
  return pair_metrics_impl(pl, est, nullptr, est_off, tgt, nullptr, tgt_off, len, frame_off, n_items, max_len, total_rows,
                           mask, out, workspace, workspace_bytes, stream, 7);
}

extern "C" int ssr_pair_metrics_est64(const ssr_plan* pl, const double* est, const int64_t* est_off, const float* tgt,
                                      const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off, int n_items,
                                      int max_len, int64_t total_rows, unsigned mask, double* out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  return pair_metrics_impl(pl, nullptr, est, est_off, tgt, nullptr, tgt_off, len, frame_off, n_items, max_len, total_rows,
                           mask, out, workspace, workspace_bytes, stream, 7);
}

extern "C" int ssr_pair_metrics_f64(const ssr_plan* pl, const double* est, const int64_t* est_off, const double* tgt,
                                    const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off, int n_items,
                                    int max_len, int64_t total_rows, unsigned mask, double* out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return pair_metrics_impl(pl, nullptr, est, est_off, nullptr, tgt, tgt_off, len, frame_off, n_items, max_len, total_rows,
                           mask, out, workspace, workspace_bytes, stream, 7);
}

// ----------------------------------------------------------------------------------------------------
struct SpecWs { size_t off_part, off_ssim, total; int rows_per_chunk, n_chunks; SsimGeom sg; };
static SpecWs spec_ws(int n_items, int max_rows, int n_bins) {
  SpecWs w;
  w.rows_per_chunk = units_per_chunk_for(max_rows, n_items);
  w.n_chunks = ceil_div(max_rows, w.rows_per_chunk);
  w.sg = ssim_geom(max_rows, n_bins, n_items);
  size_t o = 0;
  w.off_part = o; o += align256((size_t)n_items * w.n_chunks * SSR_NPART * sizeof(double));
  w.off_ssim = o; o += align256((size_t)n_items * w.sg.n_row_tiles * w.sg.n_strips * sizeof(double));
  w.total = o;
  return w;
}

extern "C" size_t ssr_spectrogram_metrics_workspace_bytes(int n_items, int max_rows, int n_bins) {
  if (n_items <= 0) return 0;
  return spec_ws(n_items, max_rows, n_bins).total;
}

extern "C" int ssr_spectrogram_metrics(const float* est_sp, const float* tgt_sp, const int64_t* frame_off,
                                       const int32_t* n_rows, int n_items, int max_rows, int n_bins, unsigned mask,
                                       double* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!est_sp || !tgt_sp || !frame_off || !n_rows || !out) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  if ((mask & ~SSR_METRIC_ALL) || mask == 0) return fail(SSR_ERR_INVALID_ARG, "bad metric mask");
  if (max_rows < 1 || n_bins < 1) return fail(SSR_ERR_INVALID_ARG, "empty spectrogram");
  if ((int64_t)max_rows * n_bins >= ((int64_t)1 << 30))
    return fail(SSR_ERR_UNSUPPORTED, "spectrogram of 2^30 elements or more (4 GiB buffer views)");
  const bool want_ssim = mask & SSR_METRIC_SSIM;
  if (want_ssim && (max_rows < 7 || n_bins < 7)) return fail(SSR_ERR_INVALID_ARG, "win_size exceeds image extent");
  const SpecWs w = spec_ws(n_items, max_rows, n_bins);
  if (!workspace || workspace_bytes < w.total) return fail(SSR_ERR_WORKSPACE, "workspace too small");
  char* ws = (char*)workspace;
  hipStream_t s = (hipStream_t)stream;
  const bool want_red = mask & (SSR_METRIC_LSD | SSR_METRIC_SISPEC | SSR_METRIC_LOG_SISPEC);
  if (want_red) {
    SsrSpecRedParams p{est_sp, tgt_sp, frame_off, n_rows, n_bins, (int)mask, w.rows_per_chunk, w.n_chunks,
                       (double*)(ws + w.off_part)};
    hipLaunchKernelGGL(k_specred, dim3(n_items * w.n_chunks), dim3(256), SsrSpecRedLds::bytes(), s, p);
    HIP_TRY(hipGetLastError());
  }
  if (want_ssim) {
    int rc = launch_ssim(est_sp, tgt_sp, frame_off, n_rows, n_items, n_bins, w.sg, (double*)(ws + w.off_ssim), s);
    if (rc) return rc;
  }
  return launch_finalize(want_red ? (const double*)(ws + w.off_part) : nullptr, w.n_chunks,
                         want_ssim ? (const double*)(ws + w.off_ssim) : nullptr, w.sg.n_row_tiles * w.sg.n_strips, n_rows,
                         n_bins, mask, n_items, out, s);
}

// ----------------------------------------------------------------------------------------------------
extern "C" size_t ssr_ola_workspace_bytes(const ssr_plan* pl, int64_t total_rows) {
  if (!pl || total_rows <= 0) return 0;
  return align256((size_t)total_rows * pl->n_fft * sizeof(float));
}

static int run_inverse(const ssr_plan* pl, const float* in, const int64_t* in_off, const int32_t* len,
                       const int32_t* cut, const float* re, const float* im, const int64_t* frame_off,
                       const int64_t* out_off, int n_items, int max_len, int64_t total_rows, float* out,
                       void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (max_len <= pl->n_fft / 2) return fail(SSR_ERR_INVALID_ARG, "reflect padding needs len > n_fft/2");
  if (max_len >= (1 << 29)) return fail(SSR_ERR_UNSUPPORTED, "signals of 2^29 samples or more (4 GiB buffer views)");
  if (!workspace || workspace_bytes < ssr_ola_workspace_bytes(pl, total_rows)) return fail(SSR_ERR_WORKSPACE, "workspace too small");
  const int max_pairs = (int)((ssr_num_frames(pl, max_len) + 1) / 2);
  const int ppc = units_per_chunk_for(max_pairs, n_items);
  const int n_chunks = ceil_div(max_pairs, ppc);
  int rc;
  if (pl->precision == SSR_F64) {
    SsrLowpassParams<double> p{};
    p.in = in; p.in_off = in_off; p.len = len; p.cut = cut; p.frame_off = frame_off;
    p.n_fft = pl->n_fft; p.hop = pl->hop; p.pairs_per_chunk = ppc; p.n_chunks = n_chunks;
    p.spec_re = re; p.spec_im = im; p.frames = (float*)workspace;
    rc = launch_lowpass_t<double>(pl, p, n_items * n_chunks, s);
  } else {
    SsrLowpassParams<float> p{};
    p.in = in; p.in_off = in_off; p.len = len; p.cut = cut; p.frame_off = frame_off;
    p.n_fft = pl->n_fft; p.hop = pl->hop; p.pairs_per_chunk = ppc; p.n_chunks = n_chunks;
    p.spec_re = re; p.spec_im = im; p.frames = (float*)workspace;
    rc = launch_lowpass_t<float>(pl, p, n_items * n_chunks, s);
  }
  if (rc) return rc;
  SsrOlaParams q{(const float*)workspace, frame_off, len, out_off, pl->n_fft, pl->hop, pl->window64, out};
  const int bpi = ceil_div(max_len, 256);
  hipLaunchKernelGGL(k_ola, dim3((unsigned)((int64_t)n_items * bpi)), dim3(256), 0, s, q, bpi);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_fft_lowpass(const ssr_plan* pl, const float* in, const int64_t* off, const int32_t* len,
                               const int32_t* cut, const int64_t* frame_off, int n_items, int max_len,
                               int64_t total_rows, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!pl || !in || !off || !len || !cut || !frame_off || !out) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  return run_inverse(pl, in, off, len, cut, nullptr, nullptr, frame_off, off, n_items, max_len, total_rows, out, workspace,
                     workspace_bytes, (hipStream_t)stream);
}

extern "C" int ssr_istft(const ssr_plan* pl, const float* re, const float* im, const int64_t* frame_off,
                         const int32_t* len, const int64_t* out_off, int n_items, int max_len, int64_t total_rows,
                         float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!pl || !re || !im || !frame_off || !len || !out_off || !out) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  return run_inverse(pl, nullptr, nullptr, len, nullptr, re, im, frame_off, out_off, n_items, max_len, total_rows, out,
                     workspace, workspace_bytes, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------------------
static int64_t gcd64(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }

extern "C" int ssr_resample_plan(int64_t n_in, int up, int down, int* up_r, int* down_r, int64_t* n_out, int* half_len,
                                 int* n_pre_pad, int* n_pre_remove) {
  if (up < 1 || down < 1 || n_in < 0) return fail(SSR_ERR_INVALID_ARG, "up and down must be >= 1");
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

template <typename S>
static int resample_poly_t(const S* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                           const int32_t* out_len, int n_items, int max_out_len, int up, int down, const S* taps,
                           int n_taps, int n_pre_remove, S* out, void* stream) {
  if (!in || !in_off || !in_len || !out_off || !out_len || !taps || !out) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (up < 1 || down < 1 || n_taps < 1) return fail(SSR_ERR_INVALID_ARG, "bad resampling plan");
  if (n_items <= 0 || max_out_len <= 0) return SSR_OK;
  SsrResampleParamsT<S> p{in, in_off, in_len, out_off, out_len, up, down, taps, n_taps, n_pre_remove,
                          ssr_resample_pick_groups(up, down), 1, out};
  if (ssr_resample_lds_bytes(p) > 96 * 1024) p.taps_in_lds = 0;      // huge tap tables stay in HBM / L2
  const size_t lds = ssr_resample_lds_bytes(p);
  if (lds > 160 * 1024) return fail(SSR_ERR_UNSUPPORTED, "input window does not fit LDS");
  if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_resample<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int bpi = ceil_div(max_out_len, ssr_resample_opb(p));
  hipLaunchKernelGGL((k_resample<S>), dim3((unsigned)((int64_t)n_items * bpi)), dim3(SSR_RESAMPLE_NT), lds, (hipStream_t)stream, p, bpi);
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

// ----------------------------------------------------------------------------------------------------
static int xcorr_blocks(int max_len) { return max_len > 0 ? ceil_div(2 * (int64_t)max_len - 1, SSR_XC_LAGS) : 1; }

extern "C" size_t ssr_xcorr_workspace_bytes(int n_items, int max_len) {
  if (n_items <= 0) return 0;
  return 2 * align256((size_t)n_items * xcorr_blocks(max_len) * sizeof(double));
}

extern "C" int ssr_xcorr_argmax(const float* a, const int64_t* a_off, const float* b, const int64_t* b_off,
                                const int32_t* len, int n_items, int max_len, int64_t* argmax_out, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!a || !a_off || !b || !b_off || !len || !argmax_out) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  if (max_len <= 0) return fail(SSR_ERR_INVALID_ARG, "empty signals");
  const int nb = xcorr_blocks(max_len);
  const size_t half = align256((size_t)n_items * nb * sizeof(double));
  if (!workspace || workspace_bytes < 2 * half) return fail(SSR_ERR_WORKSPACE, "workspace too small");
  if ((int64_t)n_items * nb > 0x7fffffff) return fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  SsrXcorrParams p{a, a_off, b, b_off, len, nb, (double*)workspace, (int64_t*)((char*)workspace + half)};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_xcorr, dim3((unsigned)(n_items * nb)), dim3(SSR_XC_NT), SsrXcorrLds::bytes(), s, p);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_xcorr_pick, dim3(ceil_div(n_items, 64)), dim3(64), 0, s, p.best_val, p.best_idx, nb, n_items, argmax_out);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// ----------------------------------------------------------------------------------------------------
extern "C" size_t ssr_sosfiltfilt_workspace_bytes(int64_t total_len, int n_items, int edge) {
  if (total_len <= 0 || n_items <= 0 || edge < 0) return 0;
  return align256(((size_t)total_len + (size_t)2 * edge * n_items) * sizeof(double));
}

template <typename X>
static int sosfiltfilt_t(const X* x, const int64_t* off, const int32_t* len, int n_items, int64_t total_len,
                         const double* sos, const double* zi, int n_sections, int edge, double* y, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (!x || !off || !len || !sos || !zi || !y) return fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_sections < 1 || n_sections > 16) return fail(SSR_ERR_UNSUPPORTED, "n_sections must be in [1, 16]");
  if (edge < 0) return fail(SSR_ERR_INVALID_ARG, "negative edge");
  if (n_items <= 0) return SSR_OK;
  if (!workspace || workspace_bytes < ssr_sosfiltfilt_workspace_bytes(total_len, n_items, edge))
    return fail(SSR_ERR_WORKSPACE, "workspace too small");
  SsrIirParamsT<X> p{x, off, len, sos, zi, n_sections, edge, n_items, (double*)workspace, y};
  hipStream_t s = (hipStream_t)stream;
  if (n_sections <= 8) {
    const int per_wave = 8 * SSR_IIR_U;                           // utterances per one-wave workgroup
    const size_t lds = (size_t)per_wave * 4 * SSR_IIR_CH * sizeof(double);
    static thread_local int attr8 = -1;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (lds > 48 * 1024 && attr8 != dev) {
      HIP_TRY(hipFuncSetAttribute((const void*)k_sosfiltfilt<8, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr8 = dev;
    }
    hipLaunchKernelGGL((k_sosfiltfilt<8, X>), dim3(ceil_div(n_items, per_wave)), dim3(64), lds, s, p);
  } else {
    const int per_wave = 4 * SSR_IIR_U;
    const size_t lds = (size_t)per_wave * 4 * SSR_IIR_CH * sizeof(double);
    hipLaunchKernelGGL((k_sosfiltfilt<16, X>), dim3(ceil_div(n_items, per_wave)), dim3(64), lds, s, p);
  }
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_sosfiltfilt(const float* x, const int64_t* off, const int32_t* len, int n_items, int64_t total_len,
                               const double* sos, const double* zi, int n_sections, int edge, double* y,
                               void* workspace, size_t workspace_bytes, void* stream) {
  return sosfiltfilt_t<float>(x, off, len, n_items, total_len, sos, zi, n_sections, edge, y, workspace, workspace_bytes, stream);
}

extern "C" int ssr_sosfiltfilt_f64(const double* x, const int64_t* off, const int32_t* len, int n_items, int64_t total_len,
                                   const double* sos, const double* zi, int n_sections, int edge, double* y,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  return sosfiltfilt_t<double>(x, off, len, n_items, total_len, sos, zi, n_sections, edge, y, workspace, workspace_bytes, stream);
}
