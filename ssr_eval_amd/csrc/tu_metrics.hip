// libssrhip.so translation unit: SSIM / spectrogram reductions / finalisation kernels (K3-K5) and the metric entry
// points (ssr_pair_metrics*, ssr_spectrogram_metrics).
#include "ssr_host.h"
#include "ssr_metrics.h"

#ifndef SSR_SSIM_WAVES_PER_EU
#define SSR_SSIM_WAVES_PER_EU 1
#endif
#ifndef SSR_SSIM8_WPE
#define SSR_SSIM8_WPE 2
#endif
template <int CPT, bool CONTIG>
__global__ __launch_bounds__(SSR_SSIM_NT, CPT == 8 ? SSR_SSIM8_WPE : SSR_SSIM_WAVES_PER_EU) void k_ssim(SsrSsimParams p) {      // (eight columns: 256 VGPRs)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  const int tiles = p.n_row_tiles * p.n_strips;
  ssr_ssim_body<CPT, CONTIG>(p, blk, blockIdx.x % tiles, blockIdx.x / tiles, smem);
}

// row pitch of the pair pipeline's magnitude images: rows padded to 16 bytes (k_ssim's aligned loads, ssr_metrics.h CONTIG)
static int mag_pitch(int n_bins) { return (n_bins + 3) & ~3; }

__global__ __launch_bounds__(256) void k_specred(SsrSpecRedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SsrBlk blk{(int)threadIdx.x};
  ssr_specred_body(p, blk, blockIdx.x % p.n_chunks, blockIdx.x / p.n_chunks, smem);
}

__global__ __launch_bounds__(64) void k_finalize(SsrFinalizeParams p) {
  const int item = blockIdx.x * 64 + threadIdx.x;
  if (item < p.n_items) ssr_finalize_item(p, item);
}

struct SsimGeom { int rows_per_tile, n_row_tiles, n_strips, cpt; };
// aligned_rows: the pair pipeline's images (rows padded to 16 bytes on 256-byte-aligned bases) - eligible for the CONTIG kernel
static SsimGeom ssim_geom(int max_rows, int n_bins, int n_items, bool aligned_rows) {
  SsimGeom g;
  const int out_rows = max_rows - 6 > 1 ? max_rows - 6 : 1;
  int64_t r = ((int64_t)out_rows * n_items + ssr_target_wgs() - 1) / ssr_target_wgs();
  if (r < 8) r = 8;
  if (r > 128) r = 128;
  if (r > out_rows) r = out_rows;
  g.rows_per_tile = (int)r;
  g.n_row_tiles = ssr_ceil_div(out_rows, g.rows_per_tile);
  g.cpt = ssr_ssim_pick_cpt(n_bins);
  if (aligned_rows && g.cpt != 4) {
    // The CONTIG kernel (four consecutive columns per lane, both seven-row rings, no branches) costs ~0.68 of the strided one per
    // column slot (measured on 401 x 1115 images: five CONTIG strips 0.83 ms against three six-column strips 1.03 ms per 1024
    // pairs), so it is taken whenever its strips x 5 slots x 0.68 undercut the strided choice's strips x (cpt + 1).
    const int outs = n_bins - (SSR_SSIM_WIN - 1);
    if (outs > 0) {
      const int strips4 = ssr_ceil_div(outs, ssr_ssim_strip_out(4)), strips_c = ssr_ceil_div(outs, ssr_ssim_strip_out(g.cpt));
      if (strips4 * 5 * 68 < strips_c * (g.cpt + 1) * 100) g.cpt = 4;
    }
  }
#ifdef SSR_DEV_KNOBS
  static const int cpt_env = getenv("SSR_SSIM_CPT") ? atoi(getenv("SSR_SSIM_CPT")) : 0;
  if (cpt_env >= 1 && cpt_env <= SSR_SSIM_MAXCPT) g.cpt = cpt_env;
  if (cpt_env == 8 && aligned_rows) g.cpt = 8;              // the eight-column CONTIG variant (experiment: VERDICT r3 item 5)
#endif
  g.n_strips = n_bins > 6 ? ssr_ceil_div(n_bins - 6, ssr_ssim_strip_out(g.cpt)) : 1;
  return g;
}


struct PairWs {
  size_t off_est, off_tgt, off_part, off_ssim, total;
  int units_per_chunk, n_chunks;
  SsimGeom sg;
};
// want_mag: the two magnitude images are only materialised when SSIM is asked for (8 bytes per bin of the batch - 38 GB for
// 12,500 utterances of 4 s - against a few hundred bytes per item for the partial records)
static PairWs pair_ws(const ssr_plan* pl, int n_items, int max_len, int64_t total_rows, bool in64, bool want_mag) {
  PairWs w;
  const int max_T = (int)ssr_num_frames(pl, max_len);
  w.units_per_chunk = ssr_pair_units_per_chunk(pl, max_T, n_items, in64);     // depends on the engine that will run
  w.n_chunks = ssr_ceil_div(max_T, w.units_per_chunk);
  const int S = ssr_pair_interleave(pl, in64);                                 // whole interleaving groups (empty chunks write zeros)
  w.n_chunks = ssr_ceil_div(w.n_chunks, S) * S;
  w.sg = ssim_geom(max_T, pl->n_bins, n_items, true);
  size_t o = 0;
  w.off_est = o; o += want_mag ? ssr_align256((size_t)total_rows * mag_pitch(pl->n_bins) * sizeof(float)) : 0;
  w.off_tgt = o; o += want_mag ? ssr_align256((size_t)total_rows * mag_pitch(pl->n_bins) * sizeof(float)) : 0;
  w.off_part = o; o += ssr_align256((size_t)n_items * w.n_chunks * SSR_NPART * sizeof(double));
  w.off_ssim = o; o += ssr_align256((size_t)n_items * w.sg.n_row_tiles * w.sg.n_strips * sizeof(double));
  w.total = o;
  return w;
}

// ----------------------------------------------------------------------------------------------------
extern "C" size_t ssr_pair_metrics_workspace_bytes_for(const ssr_plan* pl, int n_items, int max_len, int64_t total_rows,
                                                        unsigned metric_mask) {
  if (!pl || n_items <= 0) return 0;
  const bool mag = metric_mask & SSR_METRIC_SSIM;
  // the float32 and float64-signal entry points may chunk differently (different engines): cover both
  const size_t a = pair_ws(pl, n_items, max_len, total_rows, false, mag).total, b = pair_ws(pl, n_items, max_len, total_rows, true, mag).total;
  return (a > b ? a : b) + ssr_align256((size_t)n_items * sizeof(int32_t));
}

extern "C" size_t ssr_pair_metrics_workspace_bytes(const ssr_plan* pl, int n_items, int max_len, int64_t total_rows) {
  return ssr_pair_metrics_workspace_bytes_for(pl, n_items, max_len, total_rows, SSR_METRIC_ALL);
}

template <int CPT, bool CONTIG = false> static int launch_ssim_inst(const SsrSsimParams& p, int grid, hipStream_t s) {
#ifdef SSR_DEV_KNOBS
  static const size_t extra = getenv("SSR_SSIM_LDS_EXTRA") ? (size_t)atoi(getenv("SSR_SSIM_LDS_EXTRA")) : 0;   // caps workgroups / CU
#else
  const size_t extra = 0;
#endif
  const size_t lds = SsrSsimLds<CPT, CONTIG>::bytes() + extra;
  static thread_local SsrLdsSlot slot;
  if (int rc = ssr_allow_lds((const void*)k_ssim<CPT, CONTIG>, lds, &slot)) return rc;
  hipLaunchKernelGGL((k_ssim<CPT, CONTIG>), dim3(grid), dim3(SSR_SSIM_NT), lds, s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// pitch: floats between image rows (0: F, the caller's own [T, F] tensors)
static int launch_ssim(const float* x, const float* y, const int64_t* frame_off, const int32_t* n_rows, int n_items,
                       int F, int pitch, const SsimGeom& g, double* part, hipStream_t s) {
  SsrSsimParams p{x, y, frame_off, n_rows, F, g.rows_per_tile, g.n_row_tiles, g.n_strips, part, pitch};
  const int grid = n_items * g.n_row_tiles * g.n_strips;
#ifdef SSR_DEV_KNOBS
  static const int no_contig = getenv("SSR_SSIM_NO_CONTIG") ? atoi(getenv("SSR_SSIM_NO_CONTIG")) : 0;
#else
  const int no_contig = 0;
#endif
  // four consecutive columns per thread through aligned 16-byte loads: rows and both bases 16-byte aligned
  if (g.cpt == 4 && !no_contig && pitch > 0 && pitch % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
    return launch_ssim_inst<4, true>(p, grid, s);
#ifdef SSR_DEV_KNOBS          /* the eight-column experiment (slower, 80 B of scratch per lane): profiling builds only */
  if (g.cpt == 8) {
    if (pitch > 0 && pitch % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) return launch_ssim_inst<8, true>(p, grid, s);
    return ssr_fail(SSR_ERR_UNSUPPORTED, "the eight-column SSIM kernel needs 16-byte aligned rows");
  }
#endif
  switch (g.cpt) {
    case 1: return launch_ssim_inst<1>(p, grid, s);
    case 2: return launch_ssim_inst<2>(p, grid, s);
    case 3: return launch_ssim_inst<3>(p, grid, s);
    case 4: return launch_ssim_inst<4>(p, grid, s);
    case 5: return launch_ssim_inst<5>(p, grid, s);
    case 6: return launch_ssim_inst<6>(p, grid, s);
  }
  return ssr_fail(SSR_ERR_UNSUPPORTED, "bad SSIM geometry");
}

static int launch_finalize(const double* part, int n_chunks, const double* ssim_part, int n_tiles, const int32_t* n_rows,
                           int F, unsigned mask, int n_items, double* out, hipStream_t s) {
  SsrFinalizeParams p{part, n_chunks, ssim_part, n_tiles, n_rows, F, (int)mask, n_items, out};
  hipLaunchKernelGGL(k_finalize, dim3(ssr_ceil_div(n_items, 64)), dim3(64), 0, s, p);
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
  p.units_per_chunk = w.units_per_chunk; p.n_chunks = w.n_chunks; p.interleave = ssr_pair_interleave(pl, est64 != nullptr);
  p.out_a = (float*)(ws + w.off_est); p.out_b = (float*)(ws + w.off_tgt); p.out_pitch = mag_pitch(pl->n_bins);
  p.part = (double*)(ws + w.off_part);
  return ssr_launch_stft<T>(pl, p, n_items * w.n_chunks, s);
}

// stages: 1 = STFT + LSD/SISpec epilogue, 2 = SSIM, 4 = finalise (bench.py times stages separately)
static int pair_metrics_impl(const ssr_plan* pl, const float* est, const double* est64, const int64_t* est_off,
                             const float* tgt, const double* tgt64, const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off,
                             int n_items, int max_len, int64_t total_rows, unsigned mask, double* out,
                             void* workspace, size_t workspace_bytes, void* stream, int stages) {
  if (!pl || (!est && !est64) || (!tgt && !tgt64) || !est_off || !tgt_off || !len || !frame_off || !out)
    return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  if (int rc_dev = ssr_check_plan_device(pl)) return rc_dev;
  if (max_len < 1) return ssr_fail(SSR_ERR_INVALID_ARG, "empty signals");
  if (max_len >= (1 << 29)) return ssr_fail(SSR_ERR_UNSUPPORTED, "signals of 2^29 samples or more (4 GiB buffer views)");
  if ((mask & ~SSR_METRIC_ALL) || mask == 0) return ssr_fail(SSR_ERR_INVALID_ARG, "bad metric mask");
  const int max_T = (int)ssr_num_frames(pl, max_len);
  const bool want_ssim = mask & SSR_METRIC_SSIM;
  if (want_ssim && (int64_t)max_T * pl->n_bins >= ((int64_t)1 << 30))
    return ssr_fail(SSR_ERR_UNSUPPORTED, "spectrogram of 2^30 elements or more (4 GiB buffer views)");
  if (want_ssim && (max_T < 7 || pl->n_bins < 7)) return ssr_fail(SSR_ERR_INVALID_ARG, "win_size exceeds image extent");
  const PairWs w = pair_ws(pl, n_items, max_len, total_rows, est64 != nullptr, want_ssim);
  // rows array lives at the tail of the ssim partial area's alignment slack: allocate it explicitly
  const size_t rows_bytes = ssr_align256((size_t)n_items * sizeof(int32_t));
  if (!workspace || workspace_bytes < w.total + rows_bytes) return ssr_fail(SSR_ERR_WORKSPACE, "workspace too small");
  char* ws = (char*)workspace;
  int32_t* rows = (int32_t*)(ws + w.total);
  hipStream_t s = (hipStream_t)stream;
  int rc = SSR_OK;
  if (stages & 1) {
    hipLaunchKernelGGL(k_rows_from_len, dim3(ssr_ceil_div(n_items, 256)), dim3(256), 0, s, len, n_items, pl->n_fft, pl->hop, rows);
    HIP_TRY(hipGetLastError());
    rc = pl->precision == SSR_F64
             ? pair_stage_stft<double>(pl, est, est64, est_off, tgt, tgt64, tgt_off, len, frame_off, n_items, mask, want_ssim, w, ws, s)
             : pair_stage_stft<float>(pl, est, est64, est_off, tgt, tgt64, tgt_off, len, frame_off, n_items, mask, want_ssim, w, ws, s);
    if (rc) return rc;
  }
  if ((stages & 2) && want_ssim) {
    rc = launch_ssim((const float*)(ws + w.off_est), (const float*)(ws + w.off_tgt), frame_off, rows, n_items,
                     pl->n_bins, mag_pitch(pl->n_bins), w.sg, (double*)(ws + w.off_ssim), s);
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
// ssr_pair_metrics_multi: ONE target, K estimates per item (SSR_Eval_Helper.evaluate_single, ssr_eval/eval.py:136-154: every
// degradation key of a file is scored against the same target).  The target is transformed ONCE - with estimate 0, by the pair
// kernel exactly as ssr_pair_metrics runs it (key 0 is bit-identical) - and its magnitude image written once; the other estimates
// go through the same kernel TWO PER COMPLEX TRANSFORM (no target, no metric epilogue: magnitude rows only), their LSD / SISpec
// terms come from k_specred_wave against the stored target image, and k_ssim reads that one image for every key:
// K + 1 real transforms and K + 1 images instead of 2 K and 2 K.
#include "ssr_specred_wave.h"
template <int KG> __global__ __launch_bounds__(64) void k_specred_wave(SsrSpecWaveParams p) {
  ssr_specred_wave_body<KG>(p, blockIdx.x % p.n_chunks, blockIdx.x / p.n_chunks);
}

struct MultiWs {
  PairWs w;                       // chunking of the transform passes + SSIM geometry for n_items * n_keys virtual items
  size_t plane, off_est, off_tgt, off_part_a, off_part_s, off_ssim, off_rows, total;
  int spec_rows_per_chunk, spec_chunks, spec_kg, n_tiles;
  bool fast;
};
// est64 (ssr_pair_metrics_multi_est64): the estimates are float64 signals - two per complex transform where the plan has a wave kernel for
// that (n_fft = 3 q on the rotating four-wave engine: AudioMetrics(48000), ssr_stft_r3_rot.h SSR_IN_EST64X2)
static bool multi_fast_path(const ssr_plan* pl, bool est64) {
  if (est64) return pl->precision == SSR_F64 && ssr_stft_rn_wave_radix(pl) == 3 && pl->weng.m == 1536;
  return ssr_stft_uses_wave_engine(pl, false) || ssr_stft_rn_wave_radix(pl) != 0;
}
static MultiWs multi_ws(const ssr_plan* pl, int n_items, int n_keys, int max_len, int64_t total_rows, unsigned mask, bool est64 = false) {
  MultiWs m;
  const bool want_ssim = mask & SSR_METRIC_SSIM;
  m.fast = multi_fast_path(pl, est64) && n_keys > 1;
  const bool mag = want_ssim || m.fast;
  m.w = pair_ws(pl, n_items, max_len, total_rows, est64, mag);
  const int max_T = (int)ssr_num_frames(pl, max_len);
  m.w.sg = ssim_geom(max_T, pl->n_bins, m.fast ? n_items * n_keys : n_items, true);    // (the plain passes keep ssr_pair_metrics' tiles)
  m.n_tiles = m.w.sg.n_row_tiles * m.w.sg.n_strips;
  m.plane = mag ? ssr_align256((size_t)total_rows * mag_pitch(pl->n_bins) * sizeof(float)) : 0;
  const int n_spec = ((n_keys - 1) / 2) * 2;                                  // keys whose reductions come from the images (in pairs)
  // keys of an item per wave (they share the target's rows).  Measured on cfg-3 (6 such keys, 1024 items): 1 key per wave 4.37 ms
  // (20 GB of images at 4.6 TB/s: HBM-bound), 2 per wave 3.2-3.4 ms (90 VGPRs, five waves per SIMD), 3 per wave the same,
  // 6 per wave 5.11 ms (173 VGPRs: latency-bound at two waves per SIMD)
  // after the round-4 packing of the float32 sequences (VALU-bound before, close to HBM-bound now): 2 per wave 2.92 ms, 3 per wave
  // 2.77 ms - three where the keys divide by three (cfg-3: 6)
  m.spec_kg = (n_spec > 0 && n_spec % 3 == 0) ? 3 : 2;
#ifdef SSR_DEV_KNOBS
  if (getenv("SSR_SPEC_KG")) m.spec_kg = atoi(getenv("SSR_SPEC_KG"));
#endif
  const int64_t groups = (int64_t)n_items * (n_spec > 0 ? n_spec / m.spec_kg : 1);
  int64_t spc = ((int64_t)16384 + groups - 1) / groups;                       // ~16 k one-wave workgroups
  if (spc > max_T / 8) spc = max_T / 8;
  if (spc < 1) spc = 1;
  m.spec_chunks = (int)spc;
  m.spec_rows_per_chunk = ssr_ceil_div(max_T, m.spec_chunks);
  m.spec_chunks = ssr_ceil_div(max_T, m.spec_rows_per_chunk);
  size_t o = 0;
  m.off_est = o; o += (size_t)n_keys * m.plane;
  m.off_tgt = o; o += m.plane;
  m.off_part_a = o; o += 2 * ssr_align256((size_t)n_items * m.w.n_chunks * SSR_NPART * sizeof(double));      // key 0 and an odd last key
  m.off_part_s = o; o += ssr_align256((size_t)n_items * n_keys * m.spec_chunks * SSR_NPART * sizeof(double));
  m.off_ssim = o; o += ssr_align256((size_t)n_items * n_keys * m.n_tiles * sizeof(double));
  m.off_rows = o; o += ssr_align256((size_t)n_items * sizeof(int32_t));
  m.total = o;
  return m;
}

extern "C" size_t ssr_pair_metrics_multi_workspace_bytes(const ssr_plan* pl, int n_items, int n_keys, int max_len, int64_t total_rows,
                                                         unsigned metric_mask) {
  if (!pl || n_items <= 0 || n_keys <= 0) return 0;
  return multi_ws(pl, n_items, n_keys, max_len, total_rows, metric_mask).total;
}

// (a, a64) / (b, b64): each signal as float32 OR float64 samples (the other pointer null)
template <typename T>
static int multi_stage_stft(const ssr_plan* pl, const float* a, const double* a64, const int64_t* a_off, const float* b, const double* b64,
                            const int64_t* b_off, const int32_t* len,
                            const int64_t* frame_off, int n_items, unsigned mask, bool mag, float* out_a, float* out_b, double* part,
                            const PairWs& w, hipStream_t s) {
  SsrStftParams<T> p{};
  p.a = a; p.a64 = a64; p.b = b; p.b64 = b64; p.a_off = a_off; p.b_off = b_off; p.len = len; p.frame_off = frame_off;
  p.mode = SSR_MODE_PAIR; p.out_kind = mag ? SSR_OUT_MAG : SSR_OUT_NONE; p.metric_mask = (int)mask;
  p.n_fft = pl->n_fft; p.hop = pl->hop; p.n_bins = pl->n_bins;
  p.units_per_chunk = w.units_per_chunk; p.n_chunks = w.n_chunks; p.interleave = ssr_pair_interleave(pl, a64 != nullptr);
  p.out_a = out_a; p.out_b = out_b; p.out_pitch = mag_pitch(pl->n_bins); p.part = part;
  return ssr_launch_stft<T>(pl, p, n_items * w.n_chunks, s);
}

// est / est64: the K estimates as float32 or as float64 signals (the other pointer null)
static int pair_metrics_multi_impl(const ssr_plan* pl, const float* est, const double* est64, const int64_t* est_off, const float* tgt,
                                   const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off, int n_items, int n_keys, int max_len,
                                   int64_t total_rows, unsigned mask, double* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!pl || (!est && !est64) || !est_off || !tgt || !tgt_off || !len || !frame_off || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0 || n_keys <= 0) return SSR_OK;
  if (int rc_dev = ssr_check_plan_device(pl)) return rc_dev;
  if (max_len < 1) return ssr_fail(SSR_ERR_INVALID_ARG, "empty signals");
  if (max_len >= (1 << 29)) return ssr_fail(SSR_ERR_UNSUPPORTED, "signals of 2^29 samples or more (4 GiB buffer views)");
  if ((mask & ~SSR_METRIC_ALL) || mask == 0) return ssr_fail(SSR_ERR_INVALID_ARG, "bad metric mask");
  if ((int64_t)n_items * n_keys > 0x3fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  const int max_T = (int)ssr_num_frames(pl, max_len);
  const bool want_ssim = mask & SSR_METRIC_SSIM;
  if ((int64_t)max_T * pl->n_bins >= ((int64_t)1 << 30)) return ssr_fail(SSR_ERR_UNSUPPORTED, "spectrogram of 2^30 elements or more (4 GiB buffer views)");
  if (want_ssim && (max_T < 7 || pl->n_bins < 7)) return ssr_fail(SSR_ERR_INVALID_ARG, "win_size exceeds image extent");
  const bool e64 = est64 != nullptr;
  const MultiWs m = multi_ws(pl, n_items, n_keys, max_len, total_rows, mask, e64);
  if (!workspace || workspace_bytes < m.total) return ssr_fail(SSR_ERR_WORKSPACE, "workspace too small");
  char* ws = (char*)workspace;
  hipStream_t s = (hipStream_t)stream;
  int32_t* rows = (int32_t*)(ws + m.off_rows);
  hipLaunchKernelGGL(k_rows_from_len, dim3(ssr_ceil_div(n_items, 256)), dim3(256), 0, s, len, n_items, pl->n_fft, pl->hop, rows);
  HIP_TRY(hipGetLastError());
  const bool mag = m.plane != 0;
  const size_t part_a_bytes = ssr_align256((size_t)n_items * m.w.n_chunks * SSR_NPART * sizeof(double));
  const int pitch = mag_pitch(pl->n_bins);
  const unsigned red_mask = mask & (SSR_METRIC_LSD | SSR_METRIC_LOG_SISPEC | SSR_METRIC_SISPEC);
  auto plane_of = [&](int k) { return mag ? (float*)(ws + m.off_est + (size_t)k * m.plane) : nullptr; };
  float* tgt_plane = mag ? (float*)(ws + m.off_tgt) : nullptr;
  double* ssim_part = (double*)(ws + m.off_ssim);
  // est_tgt(k, ...): estimate k with the target (metric terms in the epilogue); est_est(k, ...): estimates k and k + 1 in one complex
  // transform, images only
  auto stft = [&](const float* a, const double* a64, const int64_t* a_off, const float* b, const double* b64, const int64_t* b_off, unsigned msk,
                  float* oa, float* ob, double* part) {
    return pl->precision == SSR_F64
               ? multi_stage_stft<double>(pl, a, a64, a_off, b, b64, b_off, len, frame_off, n_items, msk, mag, oa, ob, part, m.w, s)
               : multi_stage_stft<float>(pl, a, a64, a_off, b, b64, b_off, len, frame_off, n_items, msk, mag, oa, ob, part, m.w, s);
  };
  auto est_tgt = [&](int k, unsigned msk, float* oa, float* ob, double* part) {
    return stft(est, est64, est_off + (size_t)k * n_items, tgt, nullptr, tgt_off, msk, oa, ob, part);
  };
  auto est_est = [&](int k, float* oa, float* ob) {
    return stft(est, est64, est_off + (size_t)k * n_items, est, est64, est_off + (size_t)(k + 1) * n_items, 0u, oa, ob, nullptr);
  };
  auto finalize = [&](const double* part, int n_chunks, const double* sp, int n_virtual, int key0) {
    SsrFinalizeParams p{part, n_chunks, want_ssim ? sp : nullptr, m.n_tiles, rows, pl->n_bins, (int)mask, n_virtual, out, n_items, n_keys, key0};
    hipLaunchKernelGGL(k_finalize, dim3(ssr_ceil_div(n_virtual, 64)), dim3(64), 0, s, p);
    return hipGetLastError();
  };
  auto ssim = [&](int key0, int n_k) {      // keys key0 .. key0 + n_k - 1 against the one target image
    SsrSsimParams p{plane_of(key0), tgt_plane, frame_off, rows, pl->n_bins, m.w.sg.rows_per_tile, m.w.sg.n_row_tiles, m.w.sg.n_strips,
                    ssim_part + (size_t)key0 * n_items * m.n_tiles, pitch, n_items, (int64_t)(m.plane / sizeof(float))};
    const int grid = n_items * n_k * m.n_tiles;
    if (m.w.sg.cpt == 4) return launch_ssim_inst<4, true>(p, grid, s);
#ifdef SSR_DEV_KNOBS
    if (m.w.sg.cpt == 8) return launch_ssim_inst<8, true>(p, grid, s);
#endif
    switch (m.w.sg.cpt) {
      case 1: return launch_ssim_inst<1>(p, grid, s);
      case 2: return launch_ssim_inst<2>(p, grid, s);
      case 3: return launch_ssim_inst<3>(p, grid, s);
      case 5: return launch_ssim_inst<5>(p, grid, s);
      case 6: return launch_ssim_inst<6>(p, grid, s);
    }
    return ssr_fail(SSR_ERR_UNSUPPORTED, "bad SSIM geometry");
  };
  int rc;
  if (!m.fast) {
    // every key through the pair kernel against the target, as K calls of ssr_pair_metrics would (bit-identical to them): block
    // engines (float64-signal plans have their own entry points), or a single key
    for (int k = 0; k < n_keys; ++k) {
      double* part = (double*)(ws + m.off_part_a);
      if ((rc = est_tgt(k, mask, plane_of(k), tgt_plane, part))) return rc;
      if (want_ssim && (rc = ssim(k, 1))) return rc;
      HIP_TRY(finalize(part, m.w.n_chunks, ssim_part + (size_t)k * n_items * m.n_tiles, n_items, k));
    }
    return SSR_OK;
  }
  // key 0 with the target: metrics in the epilogue, both images written
  double* part0 = (double*)(ws + m.off_part_a);
  if ((rc = est_tgt(0, mask, plane_of(0), tgt_plane, part0))) return rc;
  // keys 1 .. in pairs: two estimates per complex transform, images only
  int k = 1;
  for (; k + 1 < n_keys; k += 2)
    if ((rc = est_est(k, plane_of(k), plane_of(k + 1)))) return rc;
  const int n_spec = k - 1;               // keys 1 .. k - 1 get their reductions from the images
  double* part_last = (double*)(ws + m.off_part_a + part_a_bytes);
  const bool odd_last = k < n_keys;
  if (odd_last)                           // one estimate left: with the target again, whose rows are NOT rewritten (out_b = null)
    if ((rc = est_tgt(k, mask, plane_of(k), nullptr, part_last))) return rc;
  if (n_spec > 0 && red_mask) {
    SsrSpecWaveParams q{plane_of(1), tgt_plane, frame_off, rows, pl->n_bins, pitch, (int)red_mask, m.spec_rows_per_chunk, m.spec_chunks, n_items,
                        (int64_t)(m.plane / sizeof(float)), (double*)(ws + m.off_part_s)};
    // KG keys of an item per wave share the target's rows (n_spec is even: the keys came in pairs)
    const int kg = m.spec_kg;
    const dim3 grid((unsigned)((int64_t)(n_spec / kg) * n_items * m.spec_chunks));
    if (kg == 3) hipLaunchKernelGGL(k_specred_wave<3>, grid, dim3(64), 0, s, q);
    else if (kg == 2) hipLaunchKernelGGL(k_specred_wave<2>, grid, dim3(64), 0, s, q);
    else hipLaunchKernelGGL(k_specred_wave<1>, grid, dim3(64), 0, s, q);
    HIP_TRY(hipGetLastError());
  }
  if (want_ssim && (rc = ssim(0, n_keys))) return rc;
  HIP_TRY(finalize(part0, m.w.n_chunks, ssim_part, n_items, 0));
  if (n_spec > 0)
    HIP_TRY(finalize(red_mask ? (const double*)(ws + m.off_part_s) : nullptr, m.spec_chunks, ssim_part + (size_t)n_items * m.n_tiles, n_spec * n_items, 1));
  if (odd_last) HIP_TRY(finalize(part_last, m.w.n_chunks, ssim_part + (size_t)k * n_items * m.n_tiles, n_items, k));
  return SSR_OK;
}

extern "C" int ssr_pair_metrics_multi(const ssr_plan* pl, const float* est, const int64_t* est_off, const float* tgt, const int64_t* tgt_off,
                                      const int32_t* len, const int64_t* frame_off, int n_items, int n_keys, int max_len, int64_t total_rows,
                                      unsigned mask, double* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!est) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  return pair_metrics_multi_impl(pl, est, nullptr, est_off, tgt, tgt_off, len, frame_off, n_items, n_keys, max_len, total_rows, mask, out,
                                 workspace, workspace_bytes, stream);
}

// K float64 estimates per float32 target (ssr_hip.h): key 0 through the float64-estimate pair kernel with the target, the others two per
// complex transform into float32 magnitude rows (|.| of the unrounded float64 spectrum, rounded once), their terms from k_specred_wave
extern "C" size_t ssr_pair_metrics_multi_est64_workspace_bytes(const ssr_plan* pl, int n_items, int n_keys, int max_len, int64_t total_rows,
                                                               unsigned metric_mask) {
  if (!pl || n_items <= 0 || n_keys <= 0) return 0;
  return multi_ws(pl, n_items, n_keys, max_len, total_rows, metric_mask, true).total;
}
extern "C" int ssr_pair_metrics_multi_est64(const ssr_plan* pl, const double* est, const int64_t* est_off, const float* tgt, const int64_t* tgt_off,
                                            const int32_t* len, const int64_t* frame_off, int n_items, int n_keys, int max_len,
                                            int64_t total_rows, unsigned mask, double* out, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  if (!est) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  return pair_metrics_multi_impl(pl, nullptr, est, est_off, tgt, tgt_off, len, frame_off, n_items, n_keys, max_len, total_rows, mask, out,
                                 workspace, workspace_bytes, stream);
}

// ----------------------------------------------------------------------------------------------------
struct SpecWs { size_t off_part, off_ssim, total; int rows_per_chunk, n_chunks; SsimGeom sg; };
static SpecWs spec_ws(int n_items, int max_rows, int n_bins) {
  SpecWs w;
  w.rows_per_chunk = ssr_units_per_chunk_for(max_rows, n_items);
  w.n_chunks = ssr_ceil_div(max_rows, w.rows_per_chunk);
  w.sg = ssim_geom(max_rows, n_bins, n_items, false);
  size_t o = 0;
  w.off_part = o; o += ssr_align256((size_t)n_items * w.n_chunks * SSR_NPART * sizeof(double));
  w.off_ssim = o; o += ssr_align256((size_t)n_items * w.sg.n_row_tiles * w.sg.n_strips * sizeof(double));
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
  if (!est_sp || !tgt_sp || !frame_off || !n_rows || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  if ((mask & ~SSR_METRIC_ALL) || mask == 0) return ssr_fail(SSR_ERR_INVALID_ARG, "bad metric mask");
  if (max_rows < 1 || n_bins < 1) return ssr_fail(SSR_ERR_INVALID_ARG, "empty spectrogram");
  if ((int64_t)max_rows * n_bins >= ((int64_t)1 << 30))
    return ssr_fail(SSR_ERR_UNSUPPORTED, "spectrogram of 2^30 elements or more (4 GiB buffer views)");
  const bool want_ssim = mask & SSR_METRIC_SSIM;
  if (want_ssim && (max_rows < 7 || n_bins < 7)) return ssr_fail(SSR_ERR_INVALID_ARG, "win_size exceeds image extent");
  const SpecWs w = spec_ws(n_items, max_rows, n_bins);
  if (!workspace || workspace_bytes < w.total) return ssr_fail(SSR_ERR_WORKSPACE, "workspace too small");
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
    int rc = launch_ssim(est_sp, tgt_sp, frame_off, n_rows, n_items, n_bins, 0, w.sg, (double*)(ws + w.off_ssim), s);
    if (rc) return rc;
  }
  return launch_finalize(want_red ? (const double*)(ws + w.off_part) : nullptr, w.n_chunks,
                         want_ssim ? (const double*)(ws + w.off_ssim) : nullptr, w.sg.n_row_tiles * w.sg.n_strips, n_rows,
                         n_bins, mask, n_items, out, s);
}
