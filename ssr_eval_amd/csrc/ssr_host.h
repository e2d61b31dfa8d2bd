// Host-side glue shared by the translation units of libssrhip.so (not part of the C ABI; see include/ssr_hip.h).
// The library is compiled as several .hip files (one per kernel family / precision) so that the few hundred kernel
// instantiations build in parallel; this header carries what they share: the error state, the plan object, launch
// geometry and the launcher entry points each unit defines.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ssr_hip.h"
#include "ssr_block.h"
#include "ssr_tables.h"

int ssr_fail(int code, const std::string& msg);   // records the thread-local message behind ssr_last_error()
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return ssr_fail(SSR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

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
  SsrEngine weng;              // the wave engine's variant for float32 pairs (ssr_pick_wave_engine; !ok: none)
  DevTables<float> f32w;       // its tables where they differ from the block engine's (radix 2), else copies of the pointers
  DevTables<double> f64w;
  double* window64 = nullptr;  // always present (OLA normalisation)
  double* wss_tab = nullptr;   // [hop] overlap-added squared window where every overlapping frame exists (hop <= n_fft)
  double* wss_rcp_tab = nullptr;   // [hop] its reciprocal
  int lowpass_engine = 0;      // SSR_LOWPASS_SEGMENTS / SSR_LOWPASS_FUSED / SSR_LOWPASS_CONV (ssr_plan_set_lowpass_engine)
  // SSR_LOWPASS_CONV: torchlibrosa's float32 Conv1d weights, transposed (tu_tlconv.hip; built when the engine is selected)
  float *tl_wre_t = nullptr, *tl_wim_t = nullptr, *tl_ire_t = nullptr, *tl_iim_t = nullptr, *tl_w2 = nullptr;
  int tl_ldw = 0;
  // ssr_plan_create_ex: a caller-supplied analysis / synthesis window, center = False, constant padding - conv engine only
  std::vector<double> ex_window;     // empty: periodic Hann
  int ex_center = 1, ex_pad_reflect = 1;
  bool ex = false;                   // true: only ssr_stft(COMPLEX) / ssr_istft / ssr_fft_lowpass on the conv engine exist for this plan
  std::vector<void*> allocs;
};

template <typename T> inline const DevTables<T>& ssr_tables_of(const ssr_plan* pl);
template <> inline const DevTables<float>& ssr_tables_of<float>(const ssr_plan* pl) { return pl->f32; }
template <> inline const DevTables<double>& ssr_tables_of<double>(const ssr_plan* pl) { return pl->f64; }
template <typename T> inline const DevTables<T>& ssr_wave_tables_of(const ssr_plan* pl);
template <> inline const DevTables<float>& ssr_wave_tables_of<float>(const ssr_plan* pl) { return pl->f32w; }
template <> inline const DevTables<double>& ssr_wave_tables_of<double>(const ssr_plan* pl) { return pl->f64w; }

// Every entry point that takes a plan runs on the plan's device: its tables live there.  (The Python mirror selects
// the device before calling in; a C caller that forgot gets an error instead of an illegal address.)
int ssr_check_plan_device(const ssr_plan* pl, bool ex_ok = false);      // ex_ok: the entry point serves ssr_plan_create_ex plans

// Opt a kernel into more than 48 KiB of dynamic LDS, once per (kernel, device, largest size so far) instead of once per
// launch.  `slot` is a per-kernel static the caller owns (one per template instantiation): the largest size each device has
// been opted into - a kernel whose LDS size depends on the call (resampler rate pair, radix-3 q) grows it per device.
struct SsrLdsSlot { enum { MAX_DEV = 32 }; size_t opted[MAX_DEV] = {}; };
inline int ssr_allow_lds(const void* fn, size_t lds, SsrLdsSlot* slot) {
  if (lds <= 48 * 1024) return SSR_OK;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const bool tracked = dev >= 0 && dev < SsrLdsSlot::MAX_DEV;
  if (tracked && slot->opted[dev] >= lds) return SSR_OK;
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (tracked) slot->opted[dev] = lds;
  return SSR_OK;
}

// ---- launch geometry (deterministic functions of the batch shape; they also define the workspace layout) ----
inline int ssr_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline size_t ssr_align256(size_t x) { return (x + 255) & ~(size_t)255; }
int ssr_target_wgs();                                  // workgroups per launch aimed for when chunking items
int ssr_units_per_chunk_for(int max_units, int n_items, int target_wgs = 0);
// Pair transform of this plan on float32 (in64 = false) signals runs the wave-autonomous engine (ssr_stft_wave.h: one
// wave per workgroup) -> the chunking aims for 4x as many (one-wave) workgroups.
bool ssr_stft_uses_wave_engine(const ssr_plan* pl, bool in64);
int ssr_stft_rn_wave_radix(const ssr_plan* pl);            // 1 / 2 / 3: n_fft = R q over M = 2048, float32 pairs (ssr_stft_rn_wave.h); 0: none
int ssr_pair_units_per_chunk(const ssr_plan* pl, int max_units, int n_items, bool in64);
// chunks per interleaving group of the pair transform (1: none).  The wave engine's chunks interleave in groups of 8 so that
// the frames whose samples overlap are transformed at the same time on one XCD (ssr_stft_wave.h); n_chunks is a multiple of it.
int ssr_pair_interleave(const ssr_plan* pl, bool in64);

// ---- launchers defined by the kernel translation units --------------------------------------------------------
template <typename T> struct SsrStftParams;
template <typename T> struct SsrLowpassParams;
// tu_stft_*.hip: direct / Bluestein engines.  part 0: float32 pairs, part 1: float64-signal pairs, part 2: single
template <typename T> int ssr_launch_stft_pair(const ssr_plan*, SsrStftParams<T>&, int grid, hipStream_t);
template <typename T> int ssr_launch_stft_pair64(const ssr_plan*, SsrStftParams<T>&, int grid, hipStream_t);
template <typename T> int ssr_launch_stft_single(const ssr_plan*, SsrStftParams<T>&, int grid, hipStream_t);
// radix-3 engine (n_fft = 3q)
template <typename T> int ssr_launch_stft_r3(const ssr_plan*, SsrStftParams<T>&, int grid, hipStream_t);
// R autonomous waves per workgroup (n_fft = R q, float32 pairs); defined next to the radix-3 engine
template <typename T> int ssr_launch_stft_rn_wave(const ssr_plan*, SsrStftParams<T>&, int grid, hipStream_t);
template <typename T> int ssr_launch_stft_r3_64(const ssr_plan*, SsrStftParams<T>&, int grid, hipStream_t);
// dispatcher (tu_core.hip): fills the plan tables into p and picks the unit
template <typename T> int ssr_launch_stft(const ssr_plan*, SsrStftParams<T>&, int grid, hipStream_t);
// tu_lowpass.hip
template <typename T> int ssr_launch_lowpass(const ssr_plan*, SsrLowpassParams<T>&, int grid, hipStream_t);
// tu_tlconv.hip: the reference-arithmetic engine (dense float32 DFT products on the matrix cores)
int ssr_tl_build(ssr_plan* pl);
int ssr_tl_supported(const ssr_plan* pl);       // the conv engine can serve this n_fft (tables are built at the first launch)
inline int ssr_plan_pad(const ssr_plan* pl) { return pl->ex_center ? pl->n_fft / 2 : 0; }
// the longest item must be transformable (lengths live on the device: a shorter item in a batch is skipped and its output zeroed)
inline int ssr_check_max_len(const ssr_plan* pl, int max_len) {
  const int pad = ssr_plan_pad(pl);
  if (pad && pl->ex_pad_reflect && max_len <= pad) return ssr_fail(SSR_ERR_INVALID_ARG, "reflect padding needs len > n_fft/2");
  if ((int64_t)max_len + 2 * pad < pl->n_fft || max_len < 1) return ssr_fail(SSR_ERR_INVALID_ARG, "signals shorter than one frame");
  return SSR_OK;
}
size_t ssr_tl_workspace_bytes(const ssr_plan* pl, int64_t total_rows);
int ssr_tl_run_inverse(const ssr_plan* pl, const float* in, const int64_t* in_off, const int32_t* len, const int32_t* cut,
                       const float* re, const float* im, const int64_t* frame_off, const int64_t* out_off, int n_items,
                       int max_len, int64_t total_rows, float* out, void* workspace, size_t workspace_bytes, hipStream_t s);
int ssr_tl_run_multi(const ssr_plan* pl, const float* in, const int64_t* in_off, const int32_t* len, const int32_t* cuts_host,
                     int n_keys, const int64_t* frame_off, int n_items, int max_len, int64_t total_rows, float* out,
                     int64_t key_stride, void* workspace, size_t workspace_bytes, hipStream_t s);
int ssr_tl_stft(const ssr_plan* pl, const float* wav, const int64_t* wav_off, const int32_t* wav_len, const int64_t* frame_off,
                int n_items, int max_len, float* out_re, float* out_im, hipStream_t s);
