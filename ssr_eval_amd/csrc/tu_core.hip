// libssrhip.so translation unit: error state, plans, the STFT entry point and dispatcher, small elementwise kernels.
// See include/ssr_hip.h for the contract; kernel bodies live in the ssr_*.h headers (shared with the host emulation
// used by the CPU tests).
#include "ssr_host.h"
#include "ssr_stft.h"

#define SSR_VERSION 200

static thread_local std::string g_err;
int ssr_fail(int code, const std::string& msg) { g_err = msg; return code; }

extern "C" const char* ssr_last_error(void) { return g_err.c_str(); }
extern "C" int ssr_version(void) { return SSR_VERSION; }

int ssr_check_plan_device(const ssr_plan* pl, bool ex_ok) {
  if (pl->ex && !ex_ok)
    return ssr_fail(SSR_ERR_UNSUPPORTED, "a plan from ssr_plan_create_ex serves ssr_stft(SSR_STFT_COMPLEX), ssr_istft and ssr_fft_lowpass only");
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev != pl->device)
    return ssr_fail(SSR_ERR_INVALID_ARG, "plan was created on HIP device " + std::to_string(pl->device) +
                                             " but the current device is " + std::to_string(dev));
  return SSR_OK;
}

// workgroups per launch aimed for when chunking items
int ssr_target_wgs() {
#ifdef SSR_DEV_KNOBS
  static const int v = getenv("SSR_TARGET_WGS") ? atoi(getenv("SSR_TARGET_WGS")) : 4096;
  return v;
#else
  return 4096;
#endif
}

bool ssr_stft_uses_wave_engine(const ssr_plan* pl, bool in64) {
#ifdef SSR_DEV_KNOBS
  static const int off = getenv("SSR_NO_WAVE") ? atoi(getenv("SSR_NO_WAVE")) : 0;
  if (off) return false;
#endif
  return !in64 && !pl->eng.bluestein && pl->eng.radix == 1 && pl->eng.logn == 11;
}

int ssr_stft_rn_wave_radix(const ssr_plan* pl) {
#ifdef SSR_DEV_KNOBS
  static const int off = getenv("SSR_NO_WAVE") ? atoi(getenv("SSR_NO_WAVE")) : 0;
  if (off) return 0;
#endif
  return pl->weng.ok ? pl->weng.radix : 0;
}

int ssr_pair_units_per_chunk(const ssr_plan* pl, int max_units, int n_items, bool in64) {
  // aim for the same number of WAVES per launch whatever the workgroup size: 4096 four-wave workgroups
  int per_wg = 4;
  if (ssr_stft_uses_wave_engine(pl, in64)) per_wg = 1;
  else if (!in64 && ssr_stft_rn_wave_radix(pl)) per_wg = ssr_stft_rn_wave_radix(pl) == 3 ? 4 : ssr_stft_rn_wave_radix(pl);
  return ssr_units_per_chunk_for(max_units, n_items, per_wg < 4 ? (4 / per_wg) * ssr_target_wgs() : 0);
}

int ssr_pair_interleave(const ssr_plan* pl, bool in64) {
  if (!ssr_stft_uses_wave_engine(pl, in64)) return 1;
#ifdef SSR_DEV_KNOBS
  static const int v = getenv("SSR_WAVE_INTERLEAVE") ? atoi(getenv("SSR_WAVE_INTERLEAVE")) : 8;
  return v > 1 ? v : 1;
#else
  return 8;
#endif
}

int ssr_units_per_chunk_for(int max_units, int n_items, int target_wgs) {
  const int target = target_wgs > 0 ? target_wgs : ssr_target_wgs();
  int64_t u = ((int64_t)max_units * n_items + target - 1) / target;
  if (u < 4) u = 4;
  if (u > 128) u = 128;             // ragged batches: short workgroups keep the tail of a launch balanced
  if (u > max_units) u = max_units;
  if (u < 1) u = 1;
  return (int)u;
}

__global__ __launch_bounds__(256) void k_magphase(const float* re, const float* im, int64_t n, float eps, float* mag,
                                                  float* cosv, float* sinv) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float r = re[i], q = im[i];
    float p = ssr_fadd_rn(ssr_fmul_rn(r, r), ssr_fmul_rn(q, q));   // torch: clamp(real**2 + imag**2, eps, inf) ** 0.5 (dsp.py:78): three roundings
    p = p < eps ? eps : p;
    const float m = sqrtf(p);
    mag[i] = m;
    cosv[i] = r / m;
    sinv[i] = q / m;
  }
}

// ----------------------------------------------------------------------------------------------------
// plan
template <typename V> static int upload(ssr_plan* pl, const std::vector<V>& h, V** d) {
  *d = nullptr;
  if (h.empty()) return SSR_OK;
  HIP_TRY(hipMalloc((void**)d, h.size() * sizeof(V)));
  pl->allocs.push_back((void*)*d);
  HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(V), hipMemcpyHostToDevice));
  return SSR_OK;
}

template <typename T> static int build_dev_tables(ssr_plan* pl, DevTables<T>& d, DevTables<T>& dw) {
  SsrTables<T> t;
  if (!ssr_build_tables<T>(pl->n_fft, t)) return ssr_fail(SSR_ERR_UNSUPPORTED, "unsupported n_fft");
  int rc;
  const bool own_wave_tables = pl->weng.ok && (pl->weng.radix != pl->eng.radix || pl->weng.m != 0);
  if (own_wave_tables) {                                       // the wave engine splits (or sizes its transforms) differently: its own tables
    SsrTables<T> w;
    if (!ssr_build_tables_for<T>(pl->n_fft, pl->weng, w)) return ssr_fail(SSR_ERR_UNSUPPORTED, "unsupported n_fft");
    if ((rc = upload(pl, w.tw, &dw.tw))) return rc;
    if ((rc = upload(pl, w.wchirp, &dw.wchirp))) return rc;
    if ((rc = upload(pl, w.bfilt, &dw.bfilt))) return rc;
    if ((rc = upload(pl, w.chirp, &dw.chirp))) return rc;
  }
  if ((rc = upload(pl, t.window, &d.window))) return rc;
  if ((rc = upload(pl, t.window_h, &d.window_h))) return rc;
  if ((rc = upload(pl, t.tw, &d.tw))) return rc;
  if ((rc = upload(pl, t.wchirp, &d.wchirp))) return rc;
  if ((rc = upload(pl, t.bfilt, &d.bfilt))) return rc;
  if ((rc = upload(pl, t.chirp, &d.chirp))) return rc;
  if (pl->weng.ok && !own_wave_tables) dw = d;
  return SSR_OK;
}

template <typename T> int ssr_launch_stft(const ssr_plan* pl, SsrStftParams<T>& p, int grid, hipStream_t s) {
  const DevTables<T>& d = ssr_tables_of<T>(pl);
  p.window = d.window_h; p.tw = d.tw; p.wchirp = d.wchirp; p.bfilt = d.bfilt; p.chirp = d.chirp;
  const bool in64 = p.a64 != nullptr;
  // float64 ESTIMATE against a float32 target at n_fft = 3 q, q <= 768 (AudioMetrics(48000) behind an IIR degradation): the rotating
  // four-wave engine has an IN64 variant (round 5); every other float64 combination runs the block engines
  // (b64 with a ZERO metric mask: two float64 ESTIMATES per complex transform, images only - ssr_pair_metrics_multi_est64)
  const bool est64_rot = in64 && (p.b64 == nullptr || p.metric_mask == 0) && sizeof(T) == 8 && ssr_stft_rn_wave_radix(pl) == 3 &&
                         pl->weng.m == 1536;   // (SSR_W24_N, ssr_fft24.h)
  if ((!in64 || est64_rot) && p.mode == SSR_MODE_PAIR && ssr_stft_rn_wave_radix(pl)) {
    const DevTables<T>& w = ssr_wave_tables_of<T>(pl);
    p.tw = w.tw; p.wchirp = w.wchirp; p.bfilt = w.bfilt; p.chirp = w.chirp;
    return ssr_launch_stft_rn_wave<T>(pl, p, grid, s);
  }
  if (pl->eng.radix == 3) return in64 ? ssr_launch_stft_r3_64<T>(pl, p, grid, s) : ssr_launch_stft_r3<T>(pl, p, grid, s);
  if (p.mode != SSR_MODE_PAIR) return ssr_launch_stft_single<T>(pl, p, grid, s);
  return in64 ? ssr_launch_stft_pair64<T>(pl, p, grid, s) : ssr_launch_stft_pair<T>(pl, p, grid, s);
}
template int ssr_launch_stft<float>(const ssr_plan*, SsrStftParams<float>&, int, hipStream_t);
template int ssr_launch_stft<double>(const ssr_plan*, SsrStftParams<double>&, int, hipStream_t);

extern "C" int ssr_plan_create(int n_fft, int hop, int precision, ssr_plan** out) {
  if (!out) return ssr_fail(SSR_ERR_INVALID_ARG, "plan output pointer is null");
  *out = nullptr;
  if (n_fft < 2 || hop < 1) return ssr_fail(SSR_ERR_INVALID_ARG, "n_fft must be >= 2 and hop >= 1");
  if (precision != SSR_F32 && precision != SSR_F64) return ssr_fail(SSR_ERR_INVALID_ARG, "precision must be SSR_F32 or SSR_F64");
  SsrEngine eng = ssr_pick_engine(n_fft);
  if (!eng.ok) return ssr_fail(SSR_ERR_UNSUPPORTED, "n_fft too large: Bluestein length would exceed 8192 (n_fft <= 4096)");
  ssr_plan* pl = new ssr_plan();
  pl->n_fft = n_fft; pl->hop = hop; pl->n_bins = n_fft / 2 + 1; pl->precision = precision; pl->eng = eng;
  pl->weng = ssr_pick_wave_engine(n_fft);
  int rc = SSR_OK;
  if (hipGetDevice(&pl->device) != hipSuccess) rc = ssr_fail(SSR_ERR_HIP, "hipGetDevice failed (no HIP device?)");
  if (!rc) rc = (precision == SSR_F64) ? build_dev_tables<double>(pl, pl->f64, pl->f64w) : build_dev_tables<float>(pl, pl->f32, pl->f32w);
  if (!rc) {
    SsrTables<double> t;
    ssr_build_tables<double>(n_fft, t);
    rc = upload(pl, t.window, &pl->window64);
    if (!rc && hop <= n_fft) {
      // position pos with every overlapping frame present sees window[m + k hop]^2, m = pos mod hop, for every k with
      // m + k hop < n_fft - added from the largest k down (= ascending frame index, the order of ssr_ola_wss), clamped like it
      std::vector<double> tab((size_t)hop);
      for (int m = 0; m < hop; ++m) {
        double wss = 0.0;
        for (int mm = m + ((n_fft - 1 - m) / hop) * hop; mm >= m; mm -= hop) wss += t.window[mm] * t.window[mm];
        tab[m] = wss < 1e-11 ? 1e-11 : wss;
      }
      rc = upload(pl, tab, &pl->wss_tab);
      for (double& v : tab) v = 1.0 / v;                      // the fused overlap-add multiplies (ssr_lowpass_group.h)
      if (!rc) rc = upload(pl, tab, &pl->wss_rcp_tab);
    }
  }
  if (rc) { ssr_plan_destroy(pl); return rc; }
  *out = pl;
  return SSR_OK;
}

// FDomainHelper(window_size, hop_size, center, pad_mode, window) with anything but the defaults (dsp.py:7-15): torchlibrosa builds
// its Conv1d weights from ANY window and pads or not as asked, so the conv engine is the natural (and only) home of such a plan.
extern "C" int ssr_plan_create_ex(int n_fft, int hop, const double* window, int center, int pad_mode, ssr_plan** out) {
  if (!out) return ssr_fail(SSR_ERR_INVALID_ARG, "plan output pointer is null");
  *out = nullptr;
  if (n_fft < 2 || hop < 1) return ssr_fail(SSR_ERR_INVALID_ARG, "n_fft must be >= 2 and hop >= 1");
  if (pad_mode != SSR_PAD_REFLECT && pad_mode != SSR_PAD_CONSTANT) return ssr_fail(SSR_ERR_INVALID_ARG, "pad_mode must be SSR_PAD_REFLECT or SSR_PAD_CONSTANT");
  ssr_plan* pl = new ssr_plan();
  pl->n_fft = n_fft; pl->hop = hop; pl->n_bins = n_fft / 2 + 1; pl->precision = SSR_F32;
  pl->ex = true; pl->ex_center = center ? 1 : 0; pl->ex_pad_reflect = pad_mode == SSR_PAD_REFLECT ? 1 : 0;
  if (window) pl->ex_window.assign(window, window + n_fft);
  int rc = SSR_OK;
  if (hipGetDevice(&pl->device) != hipSuccess) rc = ssr_fail(SSR_ERR_HIP, "hipGetDevice failed (no HIP device?)");
  if (!rc) rc = ssr_tl_supported(pl);          // (tables: at the first launch, or the caller's - ssr_plan_set_tl_weights)
  if (rc) { ssr_plan_destroy(pl); return rc; }
  pl->lowpass_engine = SSR_LOWPASS_CONV;
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
  if (!pl) return ssr_fail(SSR_ERR_INVALID_ARG, "plan is null");
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
  if (n + 2 * (int64_t)ssr_plan_pad(pl) < pl->n_fft) return 0;             // (center = False only: not one whole frame)
  return 1 + (n + 2 * (int64_t)ssr_plan_pad(pl) - pl->n_fft) / pl->hop;
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
  p.units_per_chunk = ssr_units_per_chunk_for(max_units, n_items);
  p.n_chunks = ssr_ceil_div(max_units, p.units_per_chunk);
  p.out_a = out_a; p.out_b = out_b; p.part = nullptr;
  return ssr_launch_stft<T>(pl, p, n_items * p.n_chunks, s);
}

extern "C" int ssr_stft(const ssr_plan* pl, const float* wav, const int64_t* wav_off, const int32_t* wav_len,
                        const int64_t* frame_off, int n_items, int max_len, int out_kind, float* out_a, float* out_b,
                        void* stream) {
  if (!pl || !wav || !wav_off || !wav_len || !frame_off || !out_a) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (out_kind != SSR_STFT_MAG && out_kind != SSR_STFT_COMPLEX) return ssr_fail(SSR_ERR_INVALID_ARG, "bad out_kind");
  if (out_kind == SSR_STFT_COMPLEX && !out_b) return ssr_fail(SSR_ERR_INVALID_ARG, "complex output needs out_b");
  if (n_items <= 0) return SSR_OK;
  if (int rc_dev = ssr_check_plan_device(pl, out_kind == SSR_STFT_COMPLEX)) return rc_dev;
  if (max_len < 1) return ssr_fail(SSR_ERR_INVALID_ARG, "empty signals");
  if (max_len >= (1 << 29)) return ssr_fail(SSR_ERR_UNSUPPORTED, "signals of 2^29 samples or more (4 GiB buffer views)");
  hipStream_t s = (hipStream_t)stream;
  if (pl->lowpass_engine == SSR_LOWPASS_CONV && out_kind == SSR_STFT_COMPLEX)      // torchlibrosa's STFT.forward as it computes
    return ssr_tl_stft(pl, wav, wav_off, wav_len, frame_off, n_items, max_len, out_a, out_b, s);
  return pl->precision == SSR_F64
             ? stft_single_t<double>(pl, wav, wav_off, wav_len, frame_off, n_items, max_len, out_kind, out_a, out_b, s)
             : stft_single_t<float>(pl, wav, wav_off, wav_len, frame_off, n_items, max_len, out_kind, out_a, out_b, s);
}

extern "C" int ssr_magphase(const float* re, const float* im, int64_t n, float eps, float* mag, float* cosv,
                            float* sinv, void* stream) {
  if (!re || !im || !mag || !cosv || !sinv) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
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

// N2 ingest: 16-bit PCM frames (interleaved channels, as a WAV / FLAC decoder hands them over) -> float32 mono, the values
// librosa.load produces: sample / 32768 (soundfile's float32 read of PCM_16; a power of two, exact) and, for more than one
// channel, the float32 channel mean (numpy.mean over <= 8 such values: their sum is exact in float32 whatever the order, the
// division by the channel count rounds once).
__global__ __launch_bounds__(256) void k_pcm16_to_float(const int16_t* in, const int64_t* in_off, const int32_t* n_frames,
                                                        const int32_t* n_channels, int blocks_per_item, float* out,
                                                        const int64_t* out_off) {
  const int item = blockIdx.x / blocks_per_item;
  const int nf = n_frames[item], nch = n_channels[item];
  const int16_t* src = in + in_off[item];
  float* dst = out + out_off[item];
  for (int f = (blockIdx.x % blocks_per_item) * 256 + threadIdx.x; f < nf; f += blocks_per_item * 256) {
    if (nch == 1) {
      dst[f] = (float)src[f] * (1.0f / 32768.0f);
    } else {
      float acc = 0.0f;
      for (int c = 0; c < nch; ++c) acc += (float)src[(int64_t)f * nch + c] * (1.0f / 32768.0f);
      dst[f] = acc / (float)nch;
    }
  }
}

// Multi-channel SISpec (metrics.py:114-121 on [B, C, T, F] tensors with C > 1), stage 1: per image the sums on the
// DIFFERENCE d = e - t (exact in float64): {Sdd, Stt, Sdt}; log_domain applies to_log (utils.py:43-44, float32) first.
__global__ __launch_bounds__(256) void k_mc_diff_sums(const float* a, const float* b, int64_t per_item, int log_domain, double* sums) {
  __shared__ double sh[3][4];
  const float* pa = a + (int64_t)blockIdx.x * per_item;
  const float* pb = b + (int64_t)blockIdx.x * per_item;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int64_t i = threadIdx.x; i < per_item; i += 256) {
    float e = pa[i], t = pb[i];
    if (log_domain) { e = log10f(e + 1e-12f); t = log10f(t + 1e-12f); }
    const double td = (double)t, d = (double)e - td;
    s0 += d * d; s1 += td * td; s2 += d * td;
  }
  s0 = ssr_wave_sum<64>(s0); s1 = ssr_wave_sum<64>(s1); s2 = ssr_wave_sum<64>(s2);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; sh[2][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) sums[(int64_t)blockIdx.x * 3 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// Stage 2 (one thread; B and C are tiny): pow_norm is per (b, c), pow_p_norm over every dimension but the batch
// (utils.py:68-92) - the per-channel projections alpha_c = <e_c, t_c> / (sum_c ||t_c||^2 + EPS) share ONE all-channel
// target energy.  With d = e - t:  e - alpha t = d + (1 - alpha) t, so the noise energy of a channel is
// Sdd + 2 (1 - alpha) Sdt + (1 - alpha)^2 Stt - no difference of nearly equal sums however close est is to target.
// out[b] = 10 log10(||scaled||^2 / (||noise||^2 + EPS) + EPS), out[n_batch] = sum_b out[b] / n_batch  (metrics.py:120-121).
__global__ void k_mc_finalize(const double* sums, int n_batch, int n_ch, double* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double EPS = 1e-12;
  double total = 0.0;
  for (int b = 0; b < n_batch; ++b) {
    const double* s = sums + (int64_t)b * n_ch * 3;
    double stt_all = 0.0;
    for (int c = 0; c < n_ch; ++c) stt_all += s[3 * c + 1];
    double tt = 0.0, nn = 0.0;
    for (int c = 0; c < n_ch; ++c) {
      const double sdd = s[3 * c], stt = s[3 * c + 1], sdt = s[3 * c + 2];
      const double alpha = (sdt + stt) / (stt_all + EPS);
      const double om = ((stt_all - stt) + EPS - sdt) / (stt_all + EPS);        // 1 - alpha
      tt += alpha * alpha * stt;
      nn += sdd + 2.0 * om * sdt + om * om * stt;
    }
    if (nn < 0.0) nn = 0.0;
    const double v = 10.0 * log10(tt / (nn + EPS) + EPS);
    out[b] = v;
    total += v;
  }
  out[n_batch] = total / (double)n_batch;
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
  if (!x || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n <= 0) return SSR_OK;
  hipLaunchKernelGGL(k_elementwise, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (int)SSR_EW_TO_LOG, x, n, out);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_from_log(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n <= 0) return SSR_OK;
  hipLaunchKernelGGL(k_elementwise, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (int)SSR_EW_FROM_LOG, x, n, out);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_energy_sums(const float* a, const float* b, int n_items, int64_t per_item, double* sums, void* stream) {
  if (!a || !b || !sums) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0) return SSR_OK;
  if (per_item < 0) return ssr_fail(SSR_ERR_INVALID_ARG, "negative item size");
  hipLaunchKernelGGL(k_energy_sums, dim3((unsigned)n_items), dim3(256), 0, (hipStream_t)stream, a, b, per_item, sums);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_pcm16_to_float(const int16_t* pcm, const int64_t* pcm_off, const int32_t* n_frames, const int32_t* n_channels,
                                  int n_items, int max_frames, float* out, const int64_t* out_off, void* stream) {
  if (!pcm || !pcm_off || !n_frames || !n_channels || !out || !out_off) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_items <= 0 || max_frames <= 0) return SSR_OK;
  int bpi = ssr_ceil_div(max_frames, 256 * 8);                     // a thread converts ~8 frames
  if (bpi > 1024) bpi = 1024;
  if ((int64_t)n_items * bpi > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  hipLaunchKernelGGL(k_pcm16_to_float, dim3((unsigned)(n_items * bpi)), dim3(256), 0, (hipStream_t)stream, pcm, pcm_off, n_frames,
                     n_channels, bpi, out, out_off);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_sispec_multichannel(const float* est, const float* tgt, int n_batch, int n_channels, int64_t per_image,
                                       int log_domain, double* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!est || !tgt || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n_batch <= 0 || n_channels <= 0 || per_image < 0) return ssr_fail(SSR_ERR_INVALID_ARG, "bad tensor shape");
  const size_t need = (size_t)n_batch * n_channels * 3 * sizeof(double);
  if (!workspace || workspace_bytes < need) return ssr_fail(SSR_ERR_WORKSPACE, "workspace too small (n_batch * n_channels * 24 bytes)");
  double* sums = (double*)workspace;
  hipLaunchKernelGGL(k_mc_diff_sums, dim3((unsigned)(n_batch * n_channels)), dim3(256), 0, (hipStream_t)stream, est, tgt, per_image,
                     log_domain, sums);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_mc_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)sums, n_batch, n_channels, out);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

extern "C" int ssr_scale_items(const float* x, const float* mul, const float* div, int n_items, int64_t per_item,
                               float* out, void* stream) {
  if (!x || !mul || !div || !out) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  const int64_t n = (int64_t)n_items * per_item;
  if (n <= 0) return SSR_OK;
  hipLaunchKernelGGL(k_scale_items, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, mul, div, per_item, n, out);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}
