// libssrhip.so translation unit: K6 in the reference's arithmetic class - torchlibrosa's dense float32 DFT convolutions on the
// fp32 matrix cores (ssr_tl_gemm.h), the SSR_LOWPASS_CONV engine of ssr_fft_lowpass / ssr_istft / ssr_stft(COMPLEX).
#include <cmath>
#include <mutex>

#include "ssr_host.h"
#include "ssr_tl_gemm.h"

template <int MODE>
__global__ __launch_bounds__(SSR_TL_NT, 2) void k_tl_gemm(SsrTlParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ssr_tl_gemm_body<MODE>(p, smem);
}
__global__ __launch_bounds__(256) void k_tl_pad(SsrTlPadParams p, int blocks_per_item) {
  ssr_tl_pad_body(p, blockIdx.x / blocks_per_item, (int64_t)(blockIdx.x % blocks_per_item) * 256 + threadIdx.x);
}
__global__ __launch_bounds__(256) void k_tl_pack(SsrTlPackParams p) {
  ssr_tl_pack_body(p, (int64_t)blockIdx.x * 256 + threadIdx.x);
}
__global__ __launch_bounds__(256) void k_tl_fold(SsrTlFoldParams p, int blocks_per_item) {
  ssr_tl_fold_body(p, blockIdx.x / blocks_per_item, (blockIdx.x % blocks_per_item) * 256 + threadIdx.x);
}

// ---- tables: torchlibrosa's Conv1d weights (DFTBase.dft_matrix / idft_matrix, STFT.__init__, ISTFT.init_real_imag_conv) ----------
// W[j][k] = exp(-2 pi i j k / n); forward weights Re / Im (W[:, :F] * hann[:, None]) stored float32; inverse weights
// Re / Im (conj(W) / n * hann[None, :]) stored float32.  The phase is reduced exactly (j k mod n) and evaluated in long double,
// the product with the float64 window is a float64 product as numpy forms it, then one rounding to float32.
// window: float64 [n] (librosa.filters.get_window(name, n_fft, fftbins=True) as the caller evaluated it) or nullptr = periodic Hann.
static void tl_host_tables(int n, int ldw, const double* window, std::vector<float>& wre_t, std::vector<float>& wim_t,
                           std::vector<float>& ire_t, std::vector<float>& iim_t, std::vector<float>& w2) {
  const int F = n / 2 + 1;
  std::vector<double> cs((size_t)n), sn((size_t)n), win((size_t)n);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int m = 0; m < n; ++m) {
    cs[m] = (double)cosl(two_pi * m / n);
    sn[m] = (double)sinl(two_pi * m / n);
    win[m] = window ? window[m] : (double)(0.5L - 0.5L * cosl(two_pi * m / n));
  }
  // exact values where numpy's are exact too
  cs[0] = 1.0; sn[0] = 0.0;
  if (n % 2 == 0) { cs[n / 2] = -1.0; sn[n / 2] = 0.0; }
  if (n % 4 == 0) { cs[n / 4] = 0.0; sn[n / 4] = 1.0; cs[3 * n / 4] = 0.0; sn[3 * n / 4] = -1.0; }
  wre_t.assign((size_t)n * ldw, 0.0f);
  wim_t.assign((size_t)n * ldw, 0.0f);
  for (int j = 0; j < n; ++j)
    for (int k = 0; k < F; ++k) {
      const int m = (int)(((int64_t)j * k) % n);
      wre_t[(size_t)j * ldw + k] = (float)(cs[m] * win[j]);
      wim_t[(size_t)j * ldw + k] = (float)(-sn[m] * win[j]);
    }
  ire_t.assign((size_t)n * n, 0.0f);
  iim_t.assign((size_t)n * n, 0.0f);
  for (int k = 0; k < n; ++k)
    for (int j = 0; j < n; ++j) {
      const int m = (int)(((int64_t)j * k) % n);
      ire_t[(size_t)k * n + j] = (float)((cs[m] / n) * win[j]);
      iim_t[(size_t)k * n + j] = (float)((sn[m] / n) * win[j]);
    }
  w2.resize((size_t)n);
  for (int m = 0; m < n; ++m) w2[m] = (float)(win[m] * win[m]);
}

static int tl_ldw(int n_fft) { return ((n_fft / 2 + 1 + SSR_TL_BN - 1) / SSR_TL_BN) * SSR_TL_BN; }

extern "C" int ssr_tl_weights_ex(int n_fft, const double* window, float* fwd_re_t, float* fwd_im_t, float* inv_re_t, float* inv_im_t,
                                 float* w2) {
  if (n_fft < 32 || n_fft > 4096 || n_fft % 32) return ssr_fail(SSR_ERR_UNSUPPORTED, "the conv engine needs n_fft = 32 m <= 4096");
  std::vector<float> a, b, c, d, e;
  const int ldw = tl_ldw(n_fft), F = n_fft / 2 + 1;
  tl_host_tables(n_fft, ldw, window, a, b, c, d, e);
  for (int j = 0; j < n_fft; ++j)
    for (int k = 0; k < F; ++k) {
      if (fwd_re_t) fwd_re_t[(size_t)j * F + k] = a[(size_t)j * ldw + k];
      if (fwd_im_t) fwd_im_t[(size_t)j * F + k] = b[(size_t)j * ldw + k];
    }
  if (inv_re_t) memcpy(inv_re_t, c.data(), c.size() * sizeof(float));
  if (inv_im_t) memcpy(inv_im_t, d.data(), d.size() * sizeof(float));
  if (w2) memcpy(w2, e.data(), e.size() * sizeof(float));
  return SSR_OK;
}

extern "C" int ssr_tl_weights(int n_fft, float* fwd_re_t, float* fwd_im_t, float* inv_re_t, float* inv_im_t, float* w2) {
  return ssr_tl_weights_ex(n_fft, nullptr, fwd_re_t, fwd_im_t, inv_re_t, inv_im_t, w2);
}

static std::mutex g_tl_mutex;
int ssr_tl_build(ssr_plan* pl) {
  std::lock_guard<std::mutex> lock(g_tl_mutex);
  if (pl->tl_w2) return SSR_OK;
  if (pl->n_fft < 32 || pl->n_fft > 4096 || pl->n_fft % 32)
    return ssr_fail(SSR_ERR_UNSUPPORTED, "the conv engine needs n_fft = 32 m <= 4096");
  std::vector<float> a, b, c, d, e;
  const int ldw = tl_ldw(pl->n_fft);
  tl_host_tables(pl->n_fft, ldw, pl->ex_window.empty() ? nullptr : pl->ex_window.data(), a, b, c, d, e);
  float* dev[5] = {};
  const std::vector<float>* src[5] = {&a, &b, &c, &d, &e};
  for (int i = 0; i < 5; ++i) {
    HIP_TRY(hipMalloc((void**)&dev[i], src[i]->size() * sizeof(float)));
    pl->allocs.push_back(dev[i]);
    HIP_TRY(hipMemcpy(dev[i], src[i]->data(), src[i]->size() * sizeof(float), hipMemcpyHostToDevice));
  }
  pl->tl_wre_t = dev[0]; pl->tl_wim_t = dev[1]; pl->tl_ire_t = dev[2]; pl->tl_iim_t = dev[3]; pl->tl_ldw = ldw;
  pl->tl_w2 = dev[4];
  return SSR_OK;
}

// ---- workspace layout: [frames: rows n_fft][spec_re: rows n_fft][spec_im: rows n_fft][xpad: rows hop + items n_fft] -----------
size_t ssr_tl_workspace_bytes(const ssr_plan* pl, int64_t total_rows) {
  // n_items <= total_rows (every item has at least one frame)
  return ssr_align256((size_t)total_rows * ((size_t)4 * pl->n_fft + pl->hop) * sizeof(float));
}

static size_t tl_lds_bytes(bool inv) {
  return (size_t)2 * ((inv ? 2 : 1) * SSR_TL_BM * SSR_TL_LDA + 2 * SSR_TL_BK * SSR_TL_BN) * sizeof(float);
}

template <int MODE> static int tl_launch(const SsrTlParams& p, int n_tiles, hipStream_t s) {
  const int64_t grid = (int64_t)n_tiles * p.n_items * p.m_tiles;
  if (grid > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  if (grid <= 0) return SSR_OK;
  hipLaunchKernelGGL((k_tl_gemm<MODE>), dim3((unsigned)grid), dim3(SSR_TL_NT), tl_lds_bytes(MODE == SSR_TL_INV), s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

static void tl_fill(const ssr_plan* pl, SsrTlParams& p, const int32_t* len, const int32_t* cut, const int64_t* frame_off,
                    int n_items, int max_len, int64_t total_rows, char* ws) {
  p.len = len; p.cut = cut; p.frame_off = frame_off;
  p.n_fft = pl->n_fft; p.hop = pl->hop; p.n_bins = pl->n_bins; p.n_items = n_items;
  p.pad = ssr_plan_pad(pl); p.pad_reflect = pl->ex_pad_reflect;
  p.m_tiles = ssr_ceil_div(ssr_num_frames(pl, max_len), SSR_TL_BM);
  p.wre_t = pl->tl_wre_t; p.wim_t = pl->tl_wim_t; p.ldw = pl->tl_ldw; p.ire_t = pl->tl_ire_t; p.iim_t = pl->tl_iim_t;
  const size_t mat = (size_t)total_rows * pl->n_fft * sizeof(float);
  p.frames = (float*)ws;
  p.spec_re = (float*)(ws + mat);
  p.spec_im = (float*)(ws + 2 * mat);
  p.xpad = (const float*)(ws + 3 * mat);
}

static int tl_pad(const ssr_plan* pl, const SsrTlParams& p, const float* in, const int64_t* in_off, int max_len, hipStream_t s) {
  SsrTlPadParams q{in, in_off, p.len, p.frame_off, pl->n_fft, pl->hop, (float*)p.xpad, p.pad_stride, p.pad, p.pad_reflect};
  const int bpi = ssr_ceil_div((int64_t)max_len + pl->n_fft, 256);
  hipLaunchKernelGGL(k_tl_pad, dim3((unsigned)((int64_t)p.n_items * bpi)), dim3(256), 0, s, q, bpi);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// ssr_fft_lowpass (in != nullptr) / ssr_istft (re, im given) on the conv engine
int ssr_tl_run_inverse(const ssr_plan* pl, const float* in, const int64_t* in_off, const int32_t* len, const int32_t* cut,
                       const float* re, const float* im, const int64_t* frame_off, const int64_t* out_off, int n_items,
                       int max_len, int64_t total_rows, float* out, void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (!pl->tl_w2) return ssr_fail(SSR_ERR_INVALID_ARG, "conv engine tables missing (ssr_plan_set_lowpass_engine)");
  if (!workspace || workspace_bytes < ssr_tl_workspace_bytes(pl, total_rows)) return ssr_fail(SSR_ERR_WORKSPACE, "workspace too small");
  if (n_items > total_rows)            // (the padded copies are laid out for >= 1 frame per item: only a center = 0 batch can break that)
    return ssr_fail(SSR_ERR_INVALID_ARG, "center = 0: every item of the batch needs len >= n_fft");
  SsrTlParams p{};
  tl_fill(pl, p, len, cut, frame_off, n_items, max_len, total_rows, (char*)workspace);
  int rc;
  if (in) {
    if ((rc = tl_pad(pl, p, in, in_off, max_len, s))) return rc;
    if ((rc = tl_launch<SSR_TL_FWD_LOWPASS>(p, ssr_ceil_div(pl->n_bins, SSR_TL_BN), s))) return rc;
  } else {
    SsrTlPackParams q{re, im, total_rows, pl->n_fft, pl->n_bins, p.spec_re, p.spec_im};
    const int64_t n = total_rows * pl->n_fft;
    hipLaunchKernelGGL(k_tl_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, q);
    HIP_TRY(hipGetLastError());
    p.cut = nullptr;
  }
  static thread_local SsrLdsSlot slot;
  if ((rc = ssr_allow_lds((const void*)k_tl_gemm<SSR_TL_INV>, tl_lds_bytes(true), &slot))) return rc;
  if ((rc = tl_launch<SSR_TL_INV>(p, pl->n_fft / SSR_TL_BN, s))) return rc;
  SsrTlFoldParams f{p.frames, frame_off, len, out_off, pl->n_fft, pl->hop, pl->tl_w2, out, p.pad, p.pad_reflect};
  const int bpi = ssr_ceil_div(max_len, 256);
  hipLaunchKernelGGL(k_tl_fold, dim3((unsigned)((int64_t)n_items * bpi)), dim3(256), 0, s, f, bpi);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}

// ssr_stft(COMPLEX) on the conv engine: STFT.forward (FDomainHelper.complex_spectrogram / spectrogram_phase, dsp.py:61-81).
// The padded copy of the batch lives in a stream-ordered allocation (the entry point has no workspace argument).
int ssr_tl_stft(const ssr_plan* pl, const float* wav, const int64_t* wav_off, const int32_t* wav_len, const int64_t* frame_off,
                int n_items, int max_len, float* out_re, float* out_im, hipStream_t s) {
  if (!pl->tl_w2) return ssr_fail(SSR_ERR_INVALID_ARG, "conv engine tables missing (ssr_plan_set_lowpass_engine)");
  if (int rc_len = ssr_check_max_len(pl, max_len)) return rc_len;
  const int64_t pad_stride = (((int64_t)max_len + pl->n_fft + 3) / 4) * 4;
  const size_t bytes = (size_t)n_items * pad_stride * sizeof(float);
  void* pad = nullptr;
  HIP_TRY(hipMallocAsync(&pad, bytes, s));
  SsrTlParams p{};
  p.len = wav_len; p.cut = nullptr; p.frame_off = frame_off;
  p.n_fft = pl->n_fft; p.hop = pl->hop; p.n_bins = pl->n_bins; p.n_items = n_items;
  p.pad = ssr_plan_pad(pl); p.pad_reflect = pl->ex_pad_reflect;
  p.m_tiles = ssr_ceil_div(ssr_num_frames(pl, max_len), SSR_TL_BM);
  p.wre_t = pl->tl_wre_t; p.wim_t = pl->tl_wim_t; p.ldw = pl->tl_ldw;
  p.xpad = (const float*)pad; p.pad_stride = pad_stride; p.out_re = out_re; p.out_im = out_im;
  int rc = tl_pad(pl, p, wav, wav_off, max_len, s);
  if (!rc) rc = tl_launch<SSR_TL_FWD_STFT>(p, ssr_ceil_div(pl->n_bins, SSR_TL_BN), s);
  (void)hipFreeAsync(pad, s);
  return rc;
}
