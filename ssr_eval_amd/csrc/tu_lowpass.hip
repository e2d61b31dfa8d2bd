// libssrhip.so translation unit: STFT-domain low-pass / inverse STFT (K6) kernels and entry points.
#include "ssr_host.h"
#include "ssr_lowpass_group.h"

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

__global__ __launch_bounds__(256) void k_ola(SsrOlaParams p, int blocks_per_item) {
  const int item = blockIdx.x / blocks_per_item;
  const int s = (blockIdx.x % blocks_per_item) * 256 + threadIdx.x;
  ssr_ola_sample(p, item, s);
}

__global__ __launch_bounds__(256) void k_ola_paired(SsrOlaParams p, int blocks_per_item) {
  const int item = blockIdx.x / blocks_per_item;
  const int s0 = ((blockIdx.x % blocks_per_item) * 256 + threadIdx.x) * 4;
  ssr_ola_paired_quad(p, item, s0);
}

template <typename T, int LOGN> static int launch_lowpass_inst(SsrLowpassParams<T>& p, int grid, hipStream_t s) {
  const size_t lds = SsrStftLds<T, LOGN>::bytes();
  static thread_local SsrLdsSlot slot;
  if (int rc = ssr_allow_lds((const void*)k_lowpass_frames<T, LOGN>, lds, &slot)) return rc;
  hipLaunchKernelGGL((k_lowpass_frames<T, LOGN>), dim3(grid), dim3((1 << LOGN) / 8), lds, s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// Wave-autonomous engine for 2048-point plans (ssr_lowpass_wave.h): one wave per workgroup, 17 KB of LDS, 2 waves / SIMD.
// (block mapping under interleaving: a group's S one-wave workgroups on one XCD, back to back - as k_stft_wave, tu_stft.inc)
template <typename T, bool ANALYSIS, bool PAIRED>
__global__ __launch_bounds__(64, 2) void k_lowpass_wave(SsrLowpassParams<T> p, int n_groups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  int logical = (int)blockIdx.x;
  if (p.interleave > 1) {
    const int S = p.interleave, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int G = (slot / S) * 8 + xcd;
    if (G >= n_groups) return;
    logical = G * S + slot % S;
  }
  const int item = logical / p.n_chunks, chunk = logical % p.n_chunks;
  ssr_lowpass_wave_body<T, true, ANALYSIS, PAIRED>(p, blk, chunk, item, smem);
}

// Fused engine (ssr_lowpass_group.h): four waves = four frame pairs per round, overlap-add inside the kernel, no workspace;
// 80.5 KB of LDS per workgroup: two per CU.
template <bool ANALYSIS>
__global__ __launch_bounds__(SSR_LG_NT, 2) void k_lowpass_group(SsrLowpassGroupParams gp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  ssr_lowpass_group_body<ANALYSIS>(gp, blk, blockIdx.x % gp.n_chunks, blockIdx.x / gp.n_chunks, smem);
}

bool ssr_lowpass_uses_wave_engine(const ssr_plan* pl) {
#ifdef SSR_DEV_KNOBS
  static const int off = getenv("SSR_NO_WAVE") ? atoi(getenv("SSR_NO_WAVE")) : 0;
  if (off) return false;
#endif
  return !pl->eng.bluestein && pl->eng.logn == 11;
}

// The wave engine hands k_ola one segment per frame PAIR (ssr_lowpass_wave.h) when the segments of every item fit the
// item's frame rows: ceil(T / 2) (n_fft + hop) <= T n_fft for every T a signal longer than n_fft / 2 can have.
bool ssr_lowpass_pairs_frames(const ssr_plan* pl) {
#ifdef SSR_DEV_KNOBS
  static const int off = getenv("SSR_NO_PAIRED") ? atoi(getenv("SSR_NO_PAIRED")) : 0;
  if (off) return false;
#endif
  if (!ssr_lowpass_uses_wave_engine(pl) || pl->hop > pl->n_fft / 2 || pl->wss_tab == nullptr) return false;
  int64_t t = 1 + (pl->n_fft / 2 + 1) / pl->hop;          // fewest frames of a signal that passes the reflect-pad check ...
  if (t % 2 == 0) ++t;                                     // ... the tightest case is the smallest odd count from there
  return (t + 1) / 2 * ssr_seg_stride(pl->n_fft, pl->hop) <= t * pl->n_fft;
}

static bool lowpass_group_eligible(const ssr_plan* pl) {
  return ssr_lowpass_uses_wave_engine(pl) && pl->precision == SSR_F64 && pl->wss_rcp_tab != nullptr && ssr_lowpass_group_ok(pl->hop);
}
bool ssr_lowpass_fuses_ola(const ssr_plan* pl) { return pl->lowpass_engine == SSR_LOWPASS_FUSED && lowpass_group_eligible(pl); }

extern "C" int ssr_plan_set_lowpass_engine(ssr_plan* pl, int engine) {
  if (!pl) return ssr_fail(SSR_ERR_INVALID_ARG, "null plan");
  if (engine != SSR_LOWPASS_SEGMENTS && engine != SSR_LOWPASS_FUSED && engine != SSR_LOWPASS_CONV)
    return ssr_fail(SSR_ERR_INVALID_ARG, "unknown low-pass engine");
  if (pl->ex && engine != SSR_LOWPASS_CONV) return ssr_fail(SSR_ERR_UNSUPPORTED, "a plan from ssr_plan_create_ex has the conv engine only");
  if (engine == SSR_LOWPASS_CONV) {
    if (int rc_dev = ssr_check_plan_device(pl, true)) return rc_dev;
    if (int rc = ssr_tl_supported(pl)) return rc;      // (tables: at the first launch, or the caller's)
  }
  if (engine == SSR_LOWPASS_FUSED && !lowpass_group_eligible(pl))
    return ssr_fail(SSR_ERR_UNSUPPORTED, "the fused low-pass engine needs a float64 2048-point plan with 228 <= hop <= 914");
  pl->lowpass_engine = engine;
  return SSR_OK;
}

static int launch_lowpass_group(const ssr_plan* pl, const float* in, const int64_t* in_off, const int32_t* len, const int32_t* cut,
                                const float* re, const float* im, const int64_t* frame_off, const int64_t* out_off, int n_items,
                                int max_len, float* out, hipStream_t s) {
  const DevTables<double>& d = ssr_tables_of<double>(pl);
  SsrLowpassGroupParams gp{};
  gp.lp.in = in; gp.lp.in_off = in_off; gp.lp.len = len; gp.lp.cut = cut; gp.lp.frame_off = frame_off;
  gp.lp.n_fft = pl->n_fft; gp.lp.hop = pl->hop; gp.lp.window = d.window; gp.lp.tw = d.tw; gp.lp.spec_re = re; gp.lp.spec_im = im;
  gp.out_off = out_off; gp.out = out; gp.window64 = pl->window64; gp.wss_tab = pl->wss_tab; gp.wss_rcp_tab = pl->wss_rcp_tab;
  const int max_rounds = ssr_ceil_div((ssr_num_frames(pl, max_len) + 1) / 2, SSR_LG_WAVES);
  // a chunk = a run of rounds of one item; a chunk that starts inside a signal re-runs one warm-up round, so chunks are kept long
  // (>= 8 rounds) and a signal of a few seconds is ONE chunk; long signals are split so that the launch has >= ~1024 workgroups
  int rpc = ssr_units_per_chunk_for(max_rounds, n_items, 1024);
  if (rpc < 8) rpc = 8;
  if (rpc > max_rounds) rpc = max_rounds;
#ifdef SSR_DEV_KNOBS
  static const int rpc_env = getenv("SSR_LG_RPC") ? atoi(getenv("SSR_LG_RPC")) : 0;
  if (rpc_env > 0) rpc = rpc_env;
#endif
  gp.rounds_per_chunk = rpc;
  gp.n_chunks = ssr_ceil_div(max_rounds, rpc);
  if ((int64_t)n_items * gp.n_chunks > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  const size_t lds = ssr_lowpass_group_lds_bytes(pl->hop);
  const bool analysis = re == nullptr;
  static thread_local SsrLdsSlot slot[2];
  if (int rc = ssr_allow_lds(analysis ? (const void*)k_lowpass_group<true> : (const void*)k_lowpass_group<false>, lds, &slot[analysis])) return rc;
  if (analysis) hipLaunchKernelGGL(k_lowpass_group<true>, dim3((unsigned)(n_items * gp.n_chunks)), dim3(SSR_LG_NT), lds, s, gp);
  else hipLaunchKernelGGL(k_lowpass_group<false>, dim3((unsigned)(n_items * gp.n_chunks)), dim3(SSR_LG_NT), lds, s, gp);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

template <typename T> int ssr_launch_lowpass(const ssr_plan* pl, SsrLowpassParams<T>& p, int grid, hipStream_t s) {
  const DevTables<T>& d = ssr_tables_of<T>(pl);
  p.window = d.window; p.tw = d.tw;
  if (pl->eng.bluestein) return ssr_fail(SSR_ERR_UNSUPPORTED, "inverse STFT needs a power-of-two n_fft in [256, 4096]");
  if (ssr_lowpass_uses_wave_engine(pl)) {
    typedef SsrWaveLds<T, true> WaveLds;
    const bool analysis = p.spec_re == nullptr, paired = ssr_lowpass_pairs_frames(pl);
    int n_groups = grid;
    if (p.interleave > 1) {
      n_groups = grid / p.interleave;
      grid = ssr_ceil_div(n_groups, 8) * 8 * p.interleave;
    }
    if (analysis && paired) hipLaunchKernelGGL((k_lowpass_wave<T, true, true>), dim3(grid), dim3(64), WaveLds::bytes(), s, p, n_groups);
    else if (analysis) hipLaunchKernelGGL((k_lowpass_wave<T, true, false>), dim3(grid), dim3(64), WaveLds::bytes(), s, p, n_groups);
    else if (paired) hipLaunchKernelGGL((k_lowpass_wave<T, false, true>), dim3(grid), dim3(64), WaveLds::bytes(), s, p, n_groups);
    else hipLaunchKernelGGL((k_lowpass_wave<T, false, false>), dim3(grid), dim3(64), WaveLds::bytes(), s, p, n_groups);
    HIP_TRY(hipGetLastError());
    return SSR_OK;
  }
  switch (pl->eng.logn) {
    case 8: return launch_lowpass_inst<T, 8>(p, grid, s);
    case 9: return launch_lowpass_inst<T, 9>(p, grid, s);
    case 10: return launch_lowpass_inst<T, 10>(p, grid, s);
    case 11: return launch_lowpass_inst<T, 11>(p, grid, s);
    case 12: return launch_lowpass_inst<T, 12>(p, grid, s);
  }
  return ssr_fail(SSR_ERR_UNSUPPORTED, "no kernel for this FFT length");
}
template int ssr_launch_lowpass<float>(const ssr_plan*, SsrLowpassParams<float>&, int, hipStream_t);
template int ssr_launch_lowpass<double>(const ssr_plan*, SsrLowpassParams<double>&, int, hipStream_t);

// ----------------------------------------------------------------------------------------------------
extern "C" size_t ssr_ola_workspace_bytes(const ssr_plan* pl, int64_t total_rows) {
  if (!pl || total_rows <= 0) return 0;
  if (pl->lowpass_engine == SSR_LOWPASS_CONV) return ssr_tl_workspace_bytes(pl, total_rows);
  return ssr_align256((size_t)total_rows * pl->n_fft * sizeof(float));
}

static int run_inverse(const ssr_plan* pl, const float* in, const int64_t* in_off, const int32_t* len,
                       const int32_t* cut, const float* re, const float* im, const int64_t* frame_off,
                       const int64_t* out_off, int n_items, int max_len, int64_t total_rows, float* out,
                       void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (int rc_len = ssr_check_max_len(pl, max_len)) return rc_len;
  if (max_len >= (1 << 29)) return ssr_fail(SSR_ERR_UNSUPPORTED, "signals of 2^29 samples or more (4 GiB buffer views)");
  if (!workspace || workspace_bytes < ssr_ola_workspace_bytes(pl, total_rows)) return ssr_fail(SSR_ERR_WORKSPACE, "workspace too small");
  if (pl->lowpass_engine == SSR_LOWPASS_CONV)
    return ssr_tl_run_inverse(pl, in, in_off, len, cut, re, im, frame_off, out_off, n_items, max_len, total_rows, out, workspace,
                              workspace_bytes, s);
  if (ssr_lowpass_fuses_ola(pl))      // (the workspace stays part of the contract: other plans / precisions overlap-add through it)
    return launch_lowpass_group(pl, in, in_off, len, cut, re, im, frame_off, out_off, n_items, max_len, out, s);
  const int max_pairs = (int)((ssr_num_frames(pl, max_len) + 1) / 2);
  const int ppc = ssr_units_per_chunk_for(max_pairs, n_items, ssr_lowpass_uses_wave_engine(pl) ? 4 * ssr_target_wgs() : 0);
  int n_chunks = ssr_ceil_div(max_pairs, ppc);
#ifdef SSR_DEV_KNOBS
  static const int il_env = getenv("SSR_LP_INTERLEAVE") ? atoi(getenv("SSR_LP_INTERLEAVE")) : 8;
#else
  const int il_env = 8;
#endif
  const int interleave = ssr_lowpass_uses_wave_engine(pl) && il_env > 1 ? il_env : 1;     // chunks come in whole groups
  n_chunks = ssr_ceil_div(n_chunks, interleave) * interleave;
  int rc;
  if (pl->precision == SSR_F64) {
    SsrLowpassParams<double> p{};
    p.in = in; p.in_off = in_off; p.len = len; p.cut = cut; p.frame_off = frame_off;
    p.n_fft = pl->n_fft; p.hop = pl->hop; p.pairs_per_chunk = ppc; p.n_chunks = n_chunks; p.interleave = interleave;
    p.spec_re = re; p.spec_im = im; p.frames = (float*)workspace;
    rc = ssr_launch_lowpass<double>(pl, p, n_items * n_chunks, s);
  } else {
    SsrLowpassParams<float> p{};
    p.in = in; p.in_off = in_off; p.len = len; p.cut = cut; p.frame_off = frame_off;
    p.n_fft = pl->n_fft; p.hop = pl->hop; p.pairs_per_chunk = ppc; p.n_chunks = n_chunks; p.interleave = interleave;
    p.spec_re = re; p.spec_im = im; p.frames = (float*)workspace;
    rc = ssr_launch_lowpass<float>(pl, p, n_items * n_chunks, s);
  }
  if (rc) return rc;
  SsrOlaParams q{(const float*)workspace, frame_off, len, out_off, pl->n_fft, pl->hop, pl->window64, out, pl->wss_tab,
                 1.0f / (float)pl->hop, 1.0f / (float)(2 * pl->hop)};
  const int bpi = ssr_ceil_div(max_len, 256), bpi4 = ssr_ceil_div(max_len, 1024);
  if (ssr_lowpass_pairs_frames(pl)) hipLaunchKernelGGL(k_ola_paired, dim3((unsigned)((int64_t)n_items * bpi4)), dim3(256), 0, s, q, bpi4);
  else hipLaunchKernelGGL(k_ola, dim3((unsigned)((int64_t)n_items * bpi)), dim3(256), 0, s, q, bpi);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_fft_lowpass(const ssr_plan* pl, const float* in, const int64_t* off, const int32_t* len,
                               const int32_t* cut, const int64_t* frame_off, int n_items, int max_len,
                               int64_t total_rows, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!pl || !in || !off || !len || !cut || !frame_off || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  if (int rc_dev = ssr_check_plan_device(pl, true)) return rc_dev;
  return run_inverse(pl, in, off, len, cut, nullptr, nullptr, frame_off, off, n_items, max_len, total_rows, out, workspace,
                     workspace_bytes, (hipStream_t)stream);
}

__global__ void k_fill_i32(int32_t* dst, int32_t v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = v;
}

extern "C" int ssr_fft_lowpass_multi(const ssr_plan* pl, const float* in, const int64_t* off, const int32_t* len,
                                     const int32_t* cuts_host, int n_keys, const int64_t* frame_off, int n_items, int max_len,
                                     int64_t total_rows, float* out, int64_t key_stride, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  if (!pl || !in || !off || !len || !cuts_host || !frame_off || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0 || n_keys <= 0) return SSR_OK;
  if (int rc_dev = ssr_check_plan_device(pl, true)) return rc_dev;
  if (int rc_len = ssr_check_max_len(pl, max_len)) return rc_len;
  if (max_len >= (1 << 29)) return ssr_fail(SSR_ERR_UNSUPPORTED, "signals of 2^29 samples or more (4 GiB buffer views)");
  hipStream_t s = (hipStream_t)stream;
  if (pl->lowpass_engine == SSR_LOWPASS_CONV)
    return ssr_tl_run_multi(pl, in, off, len, cuts_host, n_keys, frame_off, n_items, max_len, total_rows, out, key_stride, workspace,
                            workspace_bytes, s);
  // the FFT engines have nothing to share between keys (forward and inverse transform of a frame are one kernel): K plain calls
  int32_t* cut = nullptr;
  HIP_TRY(hipMallocAsync((void**)&cut, sizeof(int32_t) * (size_t)n_items, s));
  int rc = SSR_OK;
  for (int k = 0; k < n_keys && !rc; ++k) {
    hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)ssr_ceil_div(n_items, 256)), dim3(256), 0, s, cut, cuts_host[k], n_items);
    rc = run_inverse(pl, in, off, len, cut, nullptr, nullptr, frame_off, off, n_items, max_len, total_rows,
                     out + (int64_t)k * key_stride, workspace, workspace_bytes, s);
  }
  (void)hipFreeAsync(cut, s);
  return rc;
}

extern "C" int ssr_istft(const ssr_plan* pl, const float* re, const float* im, const int64_t* frame_off,
                         const int32_t* len, const int64_t* out_off, int n_items, int max_len, int64_t total_rows,
                         float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!pl || !re || !im || !frame_off || !len || !out_off || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  if (int rc_dev = ssr_check_plan_device(pl, true)) return rc_dev;
  return run_inverse(pl, nullptr, nullptr, len, nullptr, re, im, frame_off, out_off, n_items, max_len, total_rows, out,
                     workspace, workspace_bytes, (hipStream_t)stream);
}
