// libssrhip.so translation unit: K6 in the reference's arithmetic - torchlibrosa's dense float32 DFT convolutions on the
// fp32 matrix cores (ssr_tl_gemm.h), the SSR_LOWPASS_CONV engine of ssr_fft_lowpass / ssr_fft_lowpass_multi / ssr_istft /
// ssr_stft(COMPLEX).
#include <algorithm>
#include <cmath>
#include <mutex>

#include "ssr_host.h"
#include "ssr_tl_gemm.h"

template <int MODE>
__global__ __launch_bounds__(SSR_TL_FWD_NT, 2) void k_tl_fwd(SsrTlParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  ssr_tl_fwd_body<MODE>(p, smem_f);
}
template <int BM>
__global__ __launch_bounds__(BM * 4, 2) void k_tl_inv(SsrTlParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem_i[];
  ssr_tl_inv_body<BM>(p, smem_i);
}
__global__ __launch_bounds__(256) void k_tl_pad(SsrTlPadParams p, int blocks_per_item) {
  ssr_tl_pad_body(p, blockIdx.x / blocks_per_item, (int64_t)(blockIdx.x % blocks_per_item) * 256 + threadIdx.x);
}
__global__ __launch_bounds__(256) void k_tl_pack(SsrTlPackParams p) {
  __shared__ float tile[2 * 32 * 33];
  ssr_tl_pack_body(p, tile);
}
__global__ __launch_bounds__(256) void k_tl_fold(SsrTlFoldParams p, int blocks_per_item) {
  ssr_tl_fold_body(p, blockIdx.x / blocks_per_item, (blockIdx.x % blocks_per_item) * 1024 + threadIdx.x);     // 1024 samples per block
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
// Upload one set of tables (host layouts: wre_t / wim_t [n_fft][ldw], ire_t / iim_t [n_fft][n_fft], w2 [n_fft]) to the plan.
// The device copy of the Im inverse table has its rows ch > n_fft / 2 NEGATED: the mirrored channels' imaginary parts are -I and
// the inverse kernel reads +I from the un-mirrored spectrum ((-I) w == I (-w) bit for bit).  The inverse tables carry 16 rows +
// SSR_TL_BN floats of zeros as slack (a block's last chunk reads 16 channel rows whatever the block holds - the rows past it
// multiply the zero row of the spectrum -, and the last column tile of an n_fft that is not a multiple of 128 reads past a row's end).
static int tl_upload(ssr_plan* pl, std::vector<float>& a, std::vector<float>& b, std::vector<float>& c, std::vector<float>& d,
                     std::vector<float>& e, int ldw) {
  const int n = pl->n_fft;
  for (int ch = n / 2 + 1; ch < n; ++ch)
    for (int j = 0; j < n; ++j) d[(size_t)ch * n + j] = -d[(size_t)ch * n + j];
  c.resize(c.size() + (size_t)SSR_TL_BK * n + SSR_TL_BN, 0.0f);
  d.resize(d.size() + (size_t)SSR_TL_BK * n + SSR_TL_BN, 0.0f);
  // a superseded set (ssr_plan_set_tl_weights twice, or after a launch built the library's own): released once nothing in flight
  // can still read it
  if (pl->tl_w2) {
    HIP_TRY(hipDeviceSynchronize());
    float* old_tabs[5] = {pl->tl_wre_t, pl->tl_wim_t, pl->tl_ire_t, pl->tl_iim_t, pl->tl_w2};
    for (float* o : old_tabs) {
      auto it = std::find(pl->allocs.begin(), pl->allocs.end(), (void*)o);
      if (it != pl->allocs.end()) { (void)hipFree(o); pl->allocs.erase(it); }
    }
    pl->tl_wre_t = pl->tl_wim_t = pl->tl_ire_t = pl->tl_iim_t = pl->tl_w2 = nullptr;
  }
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

int ssr_tl_supported(const ssr_plan* pl) {
  if (pl->n_fft < 32 || pl->n_fft > 4096 || pl->n_fft % 32)
    return ssr_fail(SSR_ERR_UNSUPPORTED, "the conv engine needs n_fft = 32 m <= 4096");
  return SSR_OK;
}
int ssr_tl_build(ssr_plan* pl) {
  std::lock_guard<std::mutex> lock(g_tl_mutex);
  if (pl->tl_w2) return SSR_OK;
  if (int rc = ssr_tl_supported(pl)) return rc;
  std::vector<float> a, b, c, d, e;
  const int ldw = tl_ldw(pl->n_fft);
  tl_host_tables(pl->n_fft, ldw, pl->ex_window.empty() ? nullptr : pl->ex_window.data(), a, b, c, d, e);
  return tl_upload(pl, a, b, c, d, e, ldw);
}

// torchlibrosa's own conv weights, handed over by the caller in the modules' layout (STFT.conv_real.weight[:, 0, :] = fwd_re
// [n_bins][n_fft]; ISTFT.conv_real.weight[:, :, 0] = inv_re [n_fft (output sample)][n_fft (channel)]): the Python mirror builds
// them with the numpy expressions of DFTBase.dft_matrix / idft_matrix so that the products are the reference's to the last bit.
extern "C" int ssr_plan_set_tl_weights(ssr_plan* pl, const float* fwd_re, const float* fwd_im, const float* inv_re,
                                       const float* inv_im, const float* w2) {
  if (!pl || !fwd_re || !fwd_im || !inv_re || !inv_im || !w2) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (pl->n_fft < 32 || pl->n_fft > 4096 || pl->n_fft % 32)
    return ssr_fail(SSR_ERR_UNSUPPORTED, "the conv engine needs n_fft = 32 m <= 4096");
  if (int rc_dev = ssr_check_plan_device(pl, true)) return rc_dev;
  std::lock_guard<std::mutex> lock(g_tl_mutex);
  const int n = pl->n_fft, F = n / 2 + 1, ldw = tl_ldw(n);
  std::vector<float> a((size_t)n * ldw, 0.0f), b((size_t)n * ldw, 0.0f), c((size_t)n * n), d((size_t)n * n), e(w2, w2 + n);
  for (int k = 0; k < F; ++k)
    for (int j = 0; j < n; ++j) {
      a[(size_t)j * ldw + k] = fwd_re[(size_t)k * n + j];
      b[(size_t)j * ldw + k] = fwd_im[(size_t)k * n + j];
    }
  for (int j = 0; j < n; ++j)
    for (int ch = 0; ch < n; ++ch) {
      c[(size_t)ch * n + j] = inv_re[(size_t)j * n + ch];
      d[(size_t)ch * n + j] = inv_im[(size_t)j * n + ch];
    }
  return tl_upload(pl, a, b, c, d, e, ldw);
}

// ---- workspace layout ------------------------------------------------------------------------------------------------------------
//   [frames: rows n_fft][spec_re: n_bins ldt][spec_im: n_bins ldt][zero row: ldt][rowbase: (rows + 128) int64][xpad: rows hop + items n_fft]
// ldt = rows rounded up to 4, + 128 (a row tile may start up to 3 rows before an item and always covers 128 rows)
static int64_t tl_ldt(int64_t total_rows) { return ((total_rows + 3) & ~(int64_t)3) + 128; }
size_t ssr_tl_workspace_bytes(const ssr_plan* pl, int64_t total_rows) {
  // n_items <= total_rows (every item has at least one frame)
  const size_t rows = (size_t)total_rows, n = (size_t)pl->n_fft;
  return ssr_align256(sizeof(float) * (rows * n + (2 * (size_t)pl->n_bins + 1) * (size_t)tl_ldt(total_rows) + rows * ((size_t)pl->hop + n)) +
                      sizeof(int64_t) * (rows + 128));
}

struct TlWs { float *frames, *spec_re, *spec_im, *zero_row; int64_t* rowbase; float* xpad; int64_t ldt; };
static TlWs tl_carve(const ssr_plan* pl, int64_t total_rows, char* ws) {
  TlWs w;
  w.ldt = tl_ldt(total_rows);
  const size_t mat = (size_t)total_rows * pl->n_fft * sizeof(float), sp = (size_t)pl->n_bins * w.ldt * sizeof(float);
  w.frames = (float*)ws;
  w.spec_re = (float*)(ws + mat);
  w.spec_im = (float*)(ws + mat + sp);
  w.zero_row = (float*)(ws + mat + 2 * sp);
  w.rowbase = (int64_t*)(ws + mat + 2 * sp + (size_t)w.ldt * sizeof(float));
  w.xpad = (float*)((char*)w.rowbase + sizeof(int64_t) * ((size_t)total_rows + 128));
  return w;
}

static size_t tl_fwd_lds() { return (size_t)3 * (SSR_TL_FWD_BM * SSR_TL_LDA + 2 * SSR_TL_BK * SSR_TL_BN) * sizeof(float); }
static size_t tl_inv_lds(int bm) { return (size_t)3 * (2 * SSR_TL_BK * bm + 2 * SSR_TL_BK * SSR_TL_BN) * sizeof(float); }

// rows per workgroup tile of the inverse product: 128 (eight waves, one workgroup per CU) once the launch fills the chip with
// them, else 64 (four waves, two per CU)
static int tl_pick_inv(int64_t rows) {
  int bm = rows >= 4096 ? 128 : 64;
#ifdef SSR_DEV_KNOBS
  if (const char* e = getenv("SSR_TL_BM")) bm = atoi(e) == 64 ? 64 : 128;
#endif
  return bm;
}

template <int MODE> static int tl_launch_fwd(SsrTlParams p, int nt, hipStream_t s) {
  static thread_local SsrLdsSlot slot;
  if (int rc = ssr_allow_lds((const void*)k_tl_fwd<MODE>, tl_fwd_lds(), &slot)) return rc;
  const int64_t mt = p.uniform_cut >= 0 ? p.m_tiles : (int64_t)p.m_tiles * p.n_items;
  const int64_t grid = ssr_tl_grid(mt, nt);
  if (grid > 0x7fffffff || mt > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  if (grid <= 0) return SSR_OK;
  hipLaunchKernelGGL((k_tl_fwd<MODE>), dim3((unsigned)grid), dim3(SSR_TL_FWD_NT), tl_fwd_lds(), s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}
template <int BM> static int tl_launch_inv_inst(const SsrTlParams& p, int64_t grid, hipStream_t s) {
  static thread_local SsrLdsSlot slot;
  if (int rc = ssr_allow_lds((const void*)k_tl_inv<BM>, tl_inv_lds(BM), &slot)) return rc;
  hipLaunchKernelGGL((k_tl_inv<BM>), dim3((unsigned)grid), dim3(BM * 4), tl_inv_lds(BM), s, p);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}
static int tl_launch_inv(SsrTlParams p, hipStream_t s) {
  const int64_t mt = p.uniform_cut >= 0 ? p.m_tiles : (int64_t)p.m_tiles * p.n_items;
  const int64_t grid = ssr_tl_grid(mt, ssr_ceil_div(p.n_fft, SSR_TL_BN));
  if (grid > 0x7fffffff || mt > 0x7fffffff) return ssr_fail(SSR_ERR_UNSUPPORTED, "batch too large for one launch");
  if (grid <= 0) return SSR_OK;
  return p.bm == 64 ? tl_launch_inv_inst<64>(p, grid, s) : tl_launch_inv_inst<128>(p, grid, s);
}

static void tl_fill(const ssr_plan* pl, SsrTlParams& p, const int32_t* len, const int64_t* frame_off, int n_items,
                    int64_t total_rows, const TlWs& w) {
  p.len = len; p.frame_off = frame_off;
  p.n_fft = pl->n_fft; p.hop = pl->hop; p.n_bins = pl->n_bins; p.n_items = n_items; p.total_rows = total_rows;
  p.pad = ssr_plan_pad(pl); p.pad_reflect = pl->ex_pad_reflect;
  p.wre_t = pl->tl_wre_t; p.wim_t = pl->tl_wim_t; p.ldw = pl->tl_ldw; p.ire_t = pl->tl_ire_t; p.iim_t = pl->tl_iim_t;
  p.frames = w.frames; p.spec_re = w.spec_re; p.spec_im = w.spec_im; p.zero_row = w.zero_row; p.ldt = w.ldt;
  p.xpad = w.xpad; p.rowbase = w.rowbase;
}
// row tiles of a launch: uniform cut -> over the batch's rows; per-item cuts -> per item, from the row below the item's first that
// is a multiple of 4
static void tl_tiles(const ssr_plan* pl, SsrTlParams& p, int bm, int uniform_cut, int max_len) {
  p.bm = bm; p.uniform_cut = uniform_cut;
  p.m_tiles = uniform_cut >= 0 ? ssr_ceil_div(p.total_rows, bm) : ssr_ceil_div(ssr_num_frames(pl, max_len) + 3, bm);
}

static int tl_pad(const ssr_plan* pl, const SsrTlParams& p, const float* in, const int64_t* in_off, int max_len, int64_t* rowbase,
                  hipStream_t s) {
  SsrTlPadParams q{in, in_off, p.len, p.frame_off, pl->n_fft, pl->hop, (float*)p.xpad, p.pad_stride, p.pad, p.pad_reflect, rowbase};
  const int bpi = ssr_ceil_div((int64_t)max_len + pl->n_fft, 256);
  hipLaunchKernelGGL(k_tl_pad, dim3((unsigned)((int64_t)p.n_items * bpi)), dim3(256), 0, s, q, bpi);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}
static int tl_fold(const ssr_plan* pl, const SsrTlParams& p, const int64_t* out_off, int max_len, float* out, hipStream_t s) {
  SsrTlFoldParams f{p.frames, p.frame_off, p.len, out_off, pl->n_fft, pl->hop, pl->tl_w2, out, p.pad, p.pad_reflect};
  const int bpi = ssr_ceil_div(max_len, 1024);
  hipLaunchKernelGGL(k_tl_fold, dim3((unsigned)((int64_t)p.n_items * bpi)), dim3(256), 0, s, f, bpi);
  HIP_TRY(hipGetLastError());
  return SSR_OK;
}
// The library's own tables are built at the FIRST launch that finds none (ssr_plan_set_lowpass_engine / ssr_plan_create_ex only
// check that the engine can serve the plan): a caller that hands over its own tables first (ssr_plan_set_tl_weights - the Python
// mirror always does) never pays for the set it would replace (~50 MB of HBM and the long-double host build at 2048).
static int tl_tables(const ssr_plan* pl) {
  if (pl->tl_w2) return SSR_OK;
  if (pl->lowpass_engine != SSR_LOWPASS_CONV) return ssr_fail(SSR_ERR_INVALID_ARG, "conv engine tables missing (ssr_plan_set_lowpass_engine)");
  return ssr_tl_build(const_cast<ssr_plan*>(pl));
}
static int tl_check(const ssr_plan* pl, int n_items, int64_t total_rows, const void* workspace, size_t workspace_bytes) {
  if (int rc_t = tl_tables(pl)) return rc_t;
  if (!workspace || workspace_bytes < ssr_tl_workspace_bytes(pl, total_rows)) return ssr_fail(SSR_ERR_WORKSPACE, "workspace too small");
  if (n_items > total_rows)            // (the padded copies are laid out for >= 1 frame per item: only a center = 0 batch can break that)
    return ssr_fail(SSR_ERR_INVALID_ARG, "center = 0: every item of the batch needs len >= n_fft");
  return SSR_OK;
}

// ssr_fft_lowpass (in != nullptr: per-item cuts from the device array) / ssr_istft (re, im given: every bin, one "cut" = n_bins)
int ssr_tl_run_inverse(const ssr_plan* pl, const float* in, const int64_t* in_off, const int32_t* len, const int32_t* cut,
                       const float* re, const float* im, const int64_t* frame_off, const int64_t* out_off, int n_items,
                       int max_len, int64_t total_rows, float* out, void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (int rc = tl_check(pl, n_items, total_rows, workspace, workspace_bytes)) return rc;
  const TlWs w = tl_carve(pl, total_rows, (char*)workspace);
  SsrTlParams p{};
  tl_fill(pl, p, len, frame_off, n_items, total_rows, w);
  HIP_TRY(hipMemsetAsync(w.zero_row, 0, (size_t)w.ldt * sizeof(float), s));
  int rc;
  const int bm = tl_pick_inv(total_rows);
  if (in) {
    p.cut = cut; p.rowbase = nullptr;
    tl_tiles(pl, p, SSR_TL_FWD_BM, -1, max_len);
    if ((rc = tl_pad(pl, p, in, in_off, max_len, nullptr, s))) return rc;
    if ((rc = tl_launch_fwd<SSR_TL_FWD_LOWPASS>(p, ssr_ceil_div(pl->n_bins, SSR_TL_BN), s))) return rc;
    tl_tiles(pl, p, bm, -1, max_len);
  } else {
    SsrTlPackParams q{re, im, total_rows, pl->n_bins, w.spec_re, w.spec_im, w.ldt};
    hipLaunchKernelGGL(k_tl_pack, dim3((unsigned)ssr_ceil_div(total_rows, 32), (unsigned)ssr_ceil_div(pl->n_bins, 32)), dim3(256), 0, s, q);
    HIP_TRY(hipGetLastError());
    tl_tiles(pl, p, bm, pl->n_bins, max_len);
  }
  if ((rc = tl_launch_inv(p, s))) return rc;
  return tl_fold(pl, p, out_off, max_len, out, s);
}

// ssr_fft_lowpass_multi: ONE padded copy, ONE forward product at the largest cut (spectrogram_phase and mag * cos / mag * sin of a
// bin do not depend on the cut), then per key the inverse product over that key's channels and the fold.
int ssr_tl_run_multi(const ssr_plan* pl, const float* in, const int64_t* in_off, const int32_t* len, const int32_t* cuts_host,
                     int n_keys, const int64_t* frame_off, int n_items, int max_len, int64_t total_rows, float* out,
                     int64_t key_stride, void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (int rc = tl_check(pl, n_items, total_rows, workspace, workspace_bytes)) return rc;
  const TlWs w = tl_carve(pl, total_rows, (char*)workspace);
  SsrTlParams p{};
  tl_fill(pl, p, len, frame_off, n_items, total_rows, w);
  int cmax = 0;
  for (int k = 0; k < n_keys; ++k) {
    const int c = cuts_host[k] < 0 ? 0 : (cuts_host[k] > pl->n_bins ? pl->n_bins : cuts_host[k]);
    cmax = c > cmax ? c : cmax;
  }
  // rows that belong to no item (an item the transform cannot frame) read the start of the padded buffer and are never used
  HIP_TRY(hipMemsetAsync(w.zero_row, 0, (size_t)w.ldt * sizeof(float) + sizeof(int64_t) * ((size_t)total_rows + 128), s));
  int rc;
  const int bm = tl_pick_inv(total_rows);
  if ((rc = tl_pad(pl, p, in, in_off, max_len, w.rowbase, s))) return rc;
  if (cmax > 0) {
    tl_tiles(pl, p, SSR_TL_FWD_BM, cmax, max_len);
    if ((rc = tl_launch_fwd<SSR_TL_FWD_LOWPASS>(p, ssr_ceil_div(cmax, SSR_TL_BN), s))) return rc;
  }
  for (int k = 0; k < n_keys; ++k) {
    const int c = cuts_host[k] < 0 ? 0 : (cuts_host[k] > pl->n_bins ? pl->n_bins : cuts_host[k]);
    tl_tiles(pl, p, bm, c, max_len);
    if ((rc = tl_launch_inv(p, s))) return rc;
    if ((rc = tl_fold(pl, p, in_off, max_len, out + (int64_t)k * key_stride, s))) return rc;
  }
  return SSR_OK;
}

// ssr_stft(COMPLEX) on the conv engine: STFT.forward (FDomainHelper.complex_spectrogram / spectrogram_phase, dsp.py:61-81).
// The padded copy of the batch lives in a stream-ordered allocation (the entry point has no workspace argument).
int ssr_tl_stft(const ssr_plan* pl, const float* wav, const int64_t* wav_off, const int32_t* wav_len, const int64_t* frame_off,
                int n_items, int max_len, float* out_re, float* out_im, hipStream_t s) {
  if (int rc_t = tl_tables(pl)) return rc_t;
  if (int rc_len = ssr_check_max_len(pl, max_len)) return rc_len;
  const int64_t pad_stride = (((int64_t)max_len + pl->n_fft + 3) / 4) * 4;
  const size_t bytes = (size_t)n_items * pad_stride * sizeof(float);
  void* pad = nullptr;
  HIP_TRY(hipMallocAsync(&pad, bytes, s));
  SsrTlParams p{};
  p.len = wav_len; p.cut = nullptr; p.frame_off = frame_off;
  p.n_fft = pl->n_fft; p.hop = pl->hop; p.n_bins = pl->n_bins; p.n_items = n_items;
  p.pad = ssr_plan_pad(pl); p.pad_reflect = pl->ex_pad_reflect;
  p.wre_t = pl->tl_wre_t; p.wim_t = pl->tl_wim_t; p.ldw = pl->tl_ldw;
  p.xpad = (const float*)pad; p.pad_stride = pad_stride; p.out_re = out_re; p.out_im = out_im;
  tl_tiles(pl, p, SSR_TL_FWD_BM, -1, max_len);
  int rc = tl_pad(pl, p, wav, wav_off, max_len, nullptr, s);
  if (!rc) rc = tl_launch_fwd<SSR_TL_FWD_STFT>(p, ssr_ceil_div(pl->n_bins, SSR_TL_BN), s);
  (void)hipFreeAsync(pad, s);
  return rc;
}
