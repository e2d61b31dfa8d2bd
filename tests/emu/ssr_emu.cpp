// Host emulation harness (TEST INFRASTRUCTURE): runs the kernel bodies of ssr_eval_amd/csrc/*.h
// phase by phase on the CPU (g++ -DSSR_HOST_EMU) so their arithmetic can be parity-checked against
// the oracle without a GPU.  LDS is a heap buffer pre-filled with NaN bytes so a read of anything
// an earlier phase did not write poisons the result.
#define SSR_HOST_EMU 1
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../ssr_eval_amd/csrc/ssr_metrics.h"
#include "../../ssr_eval_amd/csrc/ssr_lowpass.h"
#include "../../ssr_eval_amd/csrc/ssr_iir.h"
#include "../../ssr_eval_amd/csrc/ssr_xcorr.h"
#include "../../ssr_eval_amd/csrc/ssr_resample.h"
#include "../../ssr_eval_amd/csrc/ssr_resample_mfma.h"
#include "../../ssr_eval_amd/csrc/ssr_sinc.h"
#include "../../ssr_eval_amd/csrc/ssr_stft_r3.h"
#include "../../ssr_eval_amd/csrc/ssr_stft_wave.h"
#include "../../ssr_eval_amd/csrc/ssr_lowpass_wave.h"
#include "../../ssr_eval_amd/csrc/ssr_lowpass_group.h"
#include "../../ssr_eval_amd/csrc/ssr_stft_rn_wave.h"
#include "../../ssr_eval_amd/csrc/ssr_stft_r3_rot.h"
#include "../../ssr_eval_amd/csrc/ssr_tables.h"

static std::vector<char> poisoned(size_t bytes) { return std::vector<char>(bytes + 64, (char)0xFF); }

static int g_force_ppt = 0;   // tests can force 4, 8 or 16 points per thread for the 2048-point direct engine
extern "C" void emu_force_ppt(int ppt) { g_force_ppt = ppt; }

template <typename T, int LOGN, bool BLU, int PPT>
static void run_stft_ppt(SsrStftParams<T> p, int n_items) {
  SsrBlk blk{SsrFftPlan<LOGN, PPT>::NT};
  for (int item = 0; item < n_items; ++item)
    for (int c = 0; c < p.n_chunks; ++c) {
      auto lds = poisoned(SsrStftLds<T, LOGN, PPT>::bytes());
      const bool sums = p.metric_mask & (SSR_M_SISPEC | SSR_M_LOG_SISPEC);
      if (p.mode != SSR_MODE_PAIR) ssr_stft_body<T, LOGN, BLU, SSR_MODE_SINGLE, false, PPT, 0>(p, blk, c, item, lds.data());
      else if (p.a64 && p.b64 && sums) ssr_stft_body<T, LOGN, BLU, SSR_MODE_PAIR, true, PPT, SSR_IN_BOTH64>(p, blk, c, item, lds.data());
      else if (p.a64 && p.b64) ssr_stft_body<T, LOGN, BLU, SSR_MODE_PAIR, false, PPT, SSR_IN_BOTH64>(p, blk, c, item, lds.data());
      else if (p.a64 && sums) ssr_stft_body<T, LOGN, BLU, SSR_MODE_PAIR, true, PPT, SSR_IN_EST64>(p, blk, c, item, lds.data());
      else if (p.a64) ssr_stft_body<T, LOGN, BLU, SSR_MODE_PAIR, false, PPT, SSR_IN_EST64>(p, blk, c, item, lds.data());
      else if (sums) ssr_stft_body<T, LOGN, BLU, SSR_MODE_PAIR, true, PPT, 0>(p, blk, c, item, lds.data());
      else ssr_stft_body<T, LOGN, BLU, SSR_MODE_PAIR, false, PPT, 0>(p, blk, c, item, lds.data());
    }
}

template <typename T, int LOGN, bool BLU>
static void run_stft(SsrStftParams<T> p, int n_items) {
  if (g_force_ppt == 16 && LOGN >= 9) return run_stft_ppt<T, LOGN, BLU, 16>(p, n_items);
  if (g_force_ppt == 8) return run_stft_ppt<T, LOGN, BLU, 8>(p, n_items);
  if (g_force_ppt == 4) return run_stft_ppt<T, LOGN, BLU, 4>(p, n_items);
  return run_stft_ppt<T, LOGN, BLU, ssr_stft_ppt(LOGN, BLU)>(p, n_items);
}

template <typename T, int LOGN>
static void run_stft_r3(SsrStftParams<T> p, int n_items) {
  SsrBlk blk{SsrFftPlan<LOGN, 8>::NT};
  for (int item = 0; item < n_items; ++item)
    for (int c = 0; c < p.n_chunks; ++c) {
      auto lds = poisoned(SsrStftR3Lds<T, LOGN>::bytes(p.n_fft / 3));
      const bool sums = p.metric_mask & (SSR_M_SISPEC | SSR_M_LOG_SISPEC);
      if (p.mode != SSR_MODE_PAIR) ssr_stft_r3_body<T, LOGN, SSR_MODE_SINGLE, false, 0>(p, blk, c, item, lds.data());
      else if (p.a64 && p.b64 && sums) ssr_stft_r3_body<T, LOGN, SSR_MODE_PAIR, true, SSR_IN_BOTH64>(p, blk, c, item, lds.data());
      else if (p.a64 && p.b64) ssr_stft_r3_body<T, LOGN, SSR_MODE_PAIR, false, SSR_IN_BOTH64>(p, blk, c, item, lds.data());
      else if (p.a64 && sums) ssr_stft_r3_body<T, LOGN, SSR_MODE_PAIR, true, SSR_IN_EST64>(p, blk, c, item, lds.data());
      else if (p.a64) ssr_stft_r3_body<T, LOGN, SSR_MODE_PAIR, false, SSR_IN_EST64>(p, blk, c, item, lds.data());
      else if (sums) ssr_stft_r3_body<T, LOGN, SSR_MODE_PAIR, true, 0>(p, blk, c, item, lds.data());
      else ssr_stft_r3_body<T, LOGN, SSR_MODE_PAIR, false, 0>(p, blk, c, item, lds.data());
    }
}

template <typename T>
static int emu_stft_t(int n_fft, int hop, int mode, int out_kind, int mask, const float* a, const double* a64, const float* b,
                      const double* b64,
                      const int64_t* a_off, const int64_t* b_off, const int32_t* len, const int64_t* frame_off,
                      int n_items, int units_per_chunk, int n_chunks, float* out_a, float* out_b, double* part) {
  SsrTables<T> t;
  if (!ssr_build_tables<T>(n_fft, t)) return -3;
  SsrStftParams<T> p{};
  p.a = a; p.a64 = a64; p.b = b; p.b64 = b64; p.a_off = a_off; p.b_off = b_off; p.len = len; p.frame_off = frame_off;
  p.mode = mode; p.out_kind = out_kind; p.metric_mask = mask;
  p.n_fft = n_fft; p.hop = hop; p.n_bins = n_fft / 2 + 1;
  p.units_per_chunk = units_per_chunk; p.n_chunks = n_chunks;
  p.window = t.window_h.data(); p.tw = t.tw.data();
  p.wchirp = t.wchirp.data(); p.bfilt = t.bfilt.data(); p.chirp = t.chirp.data();
  p.out_a = out_a; p.out_b = out_b; p.part = part;
  if (t.eng.radix == 3) {
    switch (t.eng.logn) {
      case 8: run_stft_r3<T, 8>(p, n_items); return 0;
      case 9: run_stft_r3<T, 9>(p, n_items); return 0;
      case 10: run_stft_r3<T, 10>(p, n_items); return 0;
      case 11: run_stft_r3<T, 11>(p, n_items); return 0;
    }
    return -3;
  }
#define CASE(L)                                                         \
  case L:                                                               \
    if (t.eng.bluestein) run_stft<T, L, true>(p, n_items);              \
    else run_stft<T, L, false>(p, n_items);                             \
    return 0;
  switch (t.eng.logn) { CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) }
#undef CASE
  return -3;
}

extern "C" int emu_stft(int precision, int n_fft, int hop, int mode, int out_kind, int mask, const float* a,
                        const float* b, const int64_t* a_off, const int64_t* b_off, const int32_t* len,
                        const int64_t* frame_off, int n_items, int units_per_chunk, int n_chunks, float* out_a,
                        float* out_b, double* part) {
  if (precision == 1)
    return emu_stft_t<double>(n_fft, hop, mode, out_kind, mask, a, nullptr, b, nullptr, a_off, b_off, len, frame_off, n_items,
                              units_per_chunk, n_chunks, out_a, out_b, part);
  return emu_stft_t<float>(n_fft, hop, mode, out_kind, mask, a, nullptr, b, nullptr, a_off, b_off, len, frame_off, n_items,
                           units_per_chunk, n_chunks, out_a, out_b, part);
}

// wave-autonomous 2048-point pair engine (ssr_stft_wave.h): one 64-lane "workgroup" per (item, chunk)
template <typename T>
static int emu_stft_wave_t(int hop, int out_kind, int mask, int split, int interleave, const float* a, const float* b, const int64_t* a_off,
                           const int64_t* b_off, const int32_t* len, const int64_t* frame_off, int n_items,
                           int units_per_chunk, int n_chunks, float* out_a, float* out_b, double* part) {
  SsrTables<T> t;
  if (!ssr_build_tables<T>(2048, t)) return -3;
  SsrStftParams<T> p{};
  p.a = a; p.b = b; p.a_off = a_off; p.b_off = b_off; p.len = len; p.frame_off = frame_off;
  p.mode = SSR_MODE_PAIR; p.out_kind = out_kind; p.metric_mask = mask;
  p.n_fft = 2048; p.hop = hop; p.n_bins = 1025;
  p.units_per_chunk = units_per_chunk; p.n_chunks = n_chunks; p.interleave = interleave;
  if (interleave > 1 && n_chunks % interleave) return -5;
  p.window = t.window_h.data(); p.tw = t.tw.data();
  p.out_a = out_a; p.out_b = out_b; p.part = part;
  SsrBlk blk{64};
  const bool sums = mask & (SSR_M_SISPEC | SSR_M_LOG_SISPEC);
  for (int item = 0; item < n_items; ++item)
    for (int c = 0; c < n_chunks; ++c) {
      if (split) {
        auto lds = poisoned(ssr_stft_wave_lds_bytes<T, true, true>());
        if (sums) ssr_stft_wave_body<T, true, true>(p, blk, c, item, lds.data());
        else ssr_stft_wave_body<T, false, true>(p, blk, c, item, lds.data());
      } else {
        auto lds = poisoned(ssr_stft_wave_lds_bytes<T, false, true>());
        if (sums) ssr_stft_wave_body<T, true, false>(p, blk, c, item, lds.data());
        else ssr_stft_wave_body<T, false, false>(p, blk, c, item, lds.data());
      }
    }
  return 0;
}
extern "C" int emu_stft_wave(int precision, int hop, int out_kind, int mask, int split, int interleave, const float* a, const float* b,
                             const int64_t* a_off, const int64_t* b_off, const int32_t* len, const int64_t* frame_off,
                             int n_items, int units_per_chunk, int n_chunks, float* out_a, float* out_b, double* part) {
  if (precision == 1)
    return emu_stft_wave_t<double>(hop, out_kind, mask, split, interleave, a, b, a_off, b_off, len, frame_off, n_items, units_per_chunk,
                                   n_chunks, out_a, out_b, part);
  return emu_stft_wave_t<float>(hop, out_kind, mask, split, interleave, a, b, a_off, b_off, len, frame_off, n_items, units_per_chunk,
                                n_chunks, out_a, out_b, part);
}

// radix-R x Bluestein pair engine on R autonomous waves (ssr_stft_rn_wave.h): n_fft = R q, M = 2048, R = 1 / 2 / 3 as
// ssr_pick_wave_engine decides
template <typename T, int NW, int NQ, int P>
static void emu_rn_wave_run(const SsrStftParams<T>& p, int n_items, int n_chunks, bool sums) {
  SsrBlk blk{64 * NW};
  for (int item = 0; item < n_items; ++item)
    for (int c = 0; c < n_chunks; ++c) {
      auto lds = poisoned(ssr_stft_rn_wave_lds_bytes<T, NW, true, P>());
      if (sums) ssr_stft_rn_wave_body<T, true, NW, NQ, P>(p, blk, c, item, lds.data());
      else ssr_stft_rn_wave_body<T, false, NW, NQ, P>(p, blk, c, item, lds.data());
    }
}
// n_fft = 3 q, q <= 768: four waves rotating through the sub-sequence transforms (ssr_stft_r3_rot.h) - the product's kernel for 2229
template <typename T>
static void emu_r3_rot_run(const SsrStftParams<T>& p, int n_items, int n_chunks, bool sums) {
  SsrBlk blk{64 * SSR_R3ROT_WAVES};
  for (int item = 0; item < n_items; ++item)
    for (int c = 0; c < n_chunks; ++c) {
      auto lds = poisoned(SsrR3RotLds<T, 24>::bytes(p.n_fft / 3));
      if constexpr (sizeof(T) == 8) {
        if (p.a64 && p.b64) { ssr_stft_r3_rot_body<T, false, 3, 24, SSR_IN_EST64X2>(p, blk, c, item, lds.data()); continue; }   // two float64 estimates (round 6)
      }
      if (p.a64 && sums) ssr_stft_r3_rot_body<T, true, 3, 24, SSR_IN_EST64>(p, blk, c, item, lds.data());     // float64 estimate (round 5)
      else if (p.a64) ssr_stft_r3_rot_body<T, false, 3, 24, SSR_IN_EST64>(p, blk, c, item, lds.data());
      else if (sums) ssr_stft_r3_rot_body<T, true, 3, 24>(p, blk, c, item, lds.data());
      else ssr_stft_r3_rot_body<T, false, 3, 24>(p, blk, c, item, lds.data());
    }
}
// m1536: 1 = what the product picks (M = 1536 for q <= 768; radix 3: the rotating four-wave kernel), 0 = force the 2048-point
// transforms, 2 = M = 1536 on the three-wave workgroups (radix 3 only differs)
template <typename T>
static int emu_stft_r3_wave_t(int n_fft, int hop, int out_kind, int mask, int m1536, const float* a, const double* a64, const float* b, const int64_t* a_off,
                              const int64_t* b_off, const int32_t* len, const int64_t* frame_off, int n_items,
                              int units_per_chunk, int n_chunks, float* out_a, float* out_b, double* part, const double* b64 = nullptr) {
  SsrEngine we = ssr_pick_wave_engine(n_fft);
  if (!we.ok) return -4;
  if (!m1536) we.m = 0;
  SsrTables<T> t;
  if (!ssr_build_tables_for<T>(n_fft, we, t)) return -3;
  SsrStftParams<T> p{};
  p.a = a; p.a64 = a64; p.b = b; p.b64 = b64; p.a_off = a_off; p.b_off = b_off; p.len = len; p.frame_off = frame_off;
  p.mode = SSR_MODE_PAIR; p.out_kind = out_kind; p.metric_mask = mask;
  p.n_fft = n_fft; p.hop = hop; p.n_bins = n_fft / 2 + 1;
  p.units_per_chunk = units_per_chunk; p.n_chunks = n_chunks;
  p.window = t.window_h.data(); p.tw = t.tw.data();
  p.wchirp = t.wchirp.data(); p.bfilt = t.bfilt.data(); p.chirp = t.chirp.data();
  p.out_a = out_a; p.out_b = out_b; p.part = part;
  const bool sums = mask & (SSR_M_SISPEC | SSR_M_LOG_SISPEC);
  const bool wide = we.q > 768;
  if (a64 && !(we.m == SSR_W24_N && we.radix == 3 && m1536 == 1)) return -5;      // the float64 estimate exists on the rotating engine only
  if (we.m == SSR_W24_N) {
    if (we.radix == 1) emu_rn_wave_run<T, 1, 3, 24>(p, n_items, n_chunks, sums);
    else if (we.radix == 2) emu_rn_wave_run<T, 2, 3, 24>(p, n_items, n_chunks, sums);
    else if (m1536 == 2) emu_rn_wave_run<T, 3, 3, 24>(p, n_items, n_chunks, sums);
    else emu_r3_rot_run<T>(p, n_items, n_chunks, sums);
    return 24;
  }
  if (we.radix == 1) { if (wide) emu_rn_wave_run<T, 1, 4, 32>(p, n_items, n_chunks, sums); else emu_rn_wave_run<T, 1, 3, 32>(p, n_items, n_chunks, sums); }
  else if (we.radix == 2) { if (wide) emu_rn_wave_run<T, 2, 4, 32>(p, n_items, n_chunks, sums); else emu_rn_wave_run<T, 2, 3, 32>(p, n_items, n_chunks, sums); }
  else emu_rn_wave_run<T, 3, 3, 32>(p, n_items, n_chunks, sums);
  return 32;
}
extern "C" int emu_stft_r3_wave(int precision, int n_fft, int hop, int out_kind, int mask, int m1536, const float* a, const float* b,
                                const int64_t* a_off, const int64_t* b_off, const int32_t* len, const int64_t* frame_off,
                                int n_items, int units_per_chunk, int n_chunks, float* out_a, float* out_b, double* part) {
  if (precision == 1)
    return emu_stft_r3_wave_t<double>(n_fft, hop, out_kind, mask, m1536, a, nullptr, b, a_off, b_off, len, frame_off, n_items, units_per_chunk,
                                      n_chunks, out_a, out_b, part);
  return emu_stft_r3_wave_t<float>(n_fft, hop, out_kind, mask, m1536, a, nullptr, b, a_off, b_off, len, frame_off, n_items, units_per_chunk,
                                   n_chunks, out_a, out_b, part);
}
// the rotating four-wave engine with a float64 estimate (float64 transforms)
extern "C" int emu_stft_r3_rot_est64(int n_fft, int hop, int out_kind, int mask, const double* a64, const float* b,
                                     const int64_t* a_off, const int64_t* b_off, const int32_t* len, const int64_t* frame_off,
                                     int n_items, int units_per_chunk, int n_chunks, float* out_a, float* out_b, double* part) {
  return emu_stft_r3_wave_t<double>(n_fft, hop, out_kind, mask, 1, nullptr, a64, b, a_off, b_off, len, frame_off, n_items, units_per_chunk,
                                    n_chunks, out_a, out_b, part);
}

// ... and with TWO float64 estimates in one complex transform, magnitude rows only (ssr_pair_metrics_multi_est64, round 6)
extern "C" int emu_stft_r3_rot_est64x2(int n_fft, int hop, const double* a64, const double* b64, const int64_t* a_off, const int64_t* b_off,
                                       const int32_t* len, const int64_t* frame_off, int n_items, int units_per_chunk, int n_chunks,
                                       float* out_a, float* out_b) {
  return emu_stft_r3_wave_t<double>(n_fft, hop, SSR_OUT_MAG, 0, 1, nullptr, a64, nullptr, a_off, b_off, len, frame_off, n_items, units_per_chunk,
                                    n_chunks, out_a, out_b, nullptr, b64);
}

// pair mode with a float64 estimate and a float32 (b) or float64 (b64) target (IN64 kernel variants)
extern "C" int emu_stft_in64(int precision, int n_fft, int hop, int out_kind, int mask, const double* a64,
                             const float* b, const double* b64, const int64_t* a_off, const int64_t* b_off,
                             const int32_t* len, const int64_t* frame_off, int n_items, int units_per_chunk, int n_chunks,
                             float* out_a, float* out_b, double* part) {
  if (precision == 1)
    return emu_stft_t<double>(n_fft, hop, SSR_MODE_PAIR, out_kind, mask, nullptr, a64, b, b64, a_off, b_off, len,
                              frame_off, n_items, units_per_chunk, n_chunks, out_a, out_b, part);
  return emu_stft_t<float>(n_fft, hop, SSR_MODE_PAIR, out_kind, mask, nullptr, a64, b, b64, a_off, b_off, len, frame_off,
                           n_items, units_per_chunk, n_chunks, out_a, out_b, part);
}

template <int CPT, bool CONTIG = false>
static void run_ssim(const SsrSsimParams& p, int n_items) {
  SsrBlk blk{SSR_SSIM_NT};
  for (int item = 0; item < n_items; ++item)
    for (int t = 0; t < p.n_row_tiles * p.n_strips; ++t) {
      auto lds = poisoned(SsrSsimLds<CPT, CONTIG>::bytes());
      ssr_ssim_body<CPT, CONTIG>(p, blk, t, item, lds.data());
    }
}

// returns the number of strips it used through *n_strips_out when part == nullptr (geometry query)
extern "C" int emu_ssim_geom(int F, int* cpt, int* n_strips) {
  *cpt = ssr_ssim_pick_cpt(F);
  *n_strips = F > 6 ? (F - 6 + ssr_ssim_strip_out(*cpt) - 1) / ssr_ssim_strip_out(*cpt) : 1;
  return 0;
}

// pitch: floats between image rows (0: F); contig: the CPT = 4 variant on 16-byte-aligned rows (pitch a multiple of 4)
extern "C" int emu_ssim(const float* x, const float* y, const int64_t* frame_off, const int32_t* n_rows, int n_items,
                        int F, int pitch, int contig, int rows_per_tile, int n_row_tiles, int n_strips, int cpt, double* part) {
  SsrSsimParams p{x, y, frame_off, n_rows, F, rows_per_tile, n_row_tiles, n_strips, part, pitch};
  if (contig) {
    if ((cpt != 4 && cpt != 8) || pitch % 4 != 0 || pitch < F) return -2;
    if (cpt == 4) run_ssim<4, true>(p, n_items);
    else run_ssim<8, true>(p, n_items);
    return 0;
  }
  switch (cpt) {
    case 1: run_ssim<1>(p, n_items); return 0;
    case 2: run_ssim<2>(p, n_items); return 0;
    case 3: run_ssim<3>(p, n_items); return 0;
    case 4: run_ssim<4>(p, n_items); return 0;
    case 5: run_ssim<5>(p, n_items); return 0;
    case 6: run_ssim<6>(p, n_items); return 0;
  }
  return -1;
}

extern "C" int emu_specred(const float* x, const float* y, const int64_t* frame_off, const int32_t* n_rows,
                           int n_items, int F, int mask, int rows_per_chunk, int n_chunks, double* part) {
  SsrSpecRedParams p{x, y, frame_off, n_rows, F, mask, rows_per_chunk, n_chunks, part};
  SsrBlk blk{SsrSpecRedLds::NT};
  for (int item = 0; item < n_items; ++item)
    for (int c = 0; c < n_chunks; ++c) {
      auto lds = poisoned(SsrSpecRedLds::bytes());
      ssr_specred_body(p, blk, c, item, lds.data());
    }
  return 0;
}

extern "C" int emu_finalize(const double* part, int n_chunks, const double* ssim_part, int n_tiles,
                            const int32_t* n_rows, int F, int mask, int n_items, double* out) {
  SsrFinalizeParams p{part, n_chunks, ssim_part, n_tiles, n_rows, F, mask, n_items, out};
  for (int i = 0; i < n_items; ++i) ssr_finalize_item(p, i);
  return 0;
}

// ---- FFT low-pass / ISTFT ---------------------------------------------------------------------------
template <typename T>
static int emu_lowpass_frames_t(int n_fft, int hop, const float* in, const int64_t* in_off, const int32_t* len,
                                const int32_t* cut, const int64_t* frame_off, int n_items, int pairs_per_chunk,
                                int n_chunks, const float* re_in, const float* im_in, float* frames) {
  SsrTables<T> t;
  if (!ssr_build_tables<T>(n_fft, t) || t.eng.bluestein) return -3;
  SsrLowpassParams<T> p{};
  p.in = in; p.in_off = in_off; p.len = len; p.cut = cut; p.frame_off = frame_off;
  p.n_fft = n_fft; p.hop = hop; p.pairs_per_chunk = pairs_per_chunk; p.n_chunks = n_chunks;
  p.window = t.window.data(); p.tw = t.tw.data(); p.spec_re = re_in; p.spec_im = im_in; p.frames = frames;
#define CASE(L)                                                               \
  case L: {                                                                   \
    SsrBlk blk{SsrFftPlan<L>::NT};                                            \
    for (int item = 0; item < n_items; ++item)                                \
      for (int c = 0; c < n_chunks; ++c) {                                    \
        auto lds = poisoned(SsrStftLds<T, L>::bytes());                       \
        ssr_lowpass_frames_body<T, L>(p, blk, c, item, lds.data());           \
      }                                                                       \
    return 0;                                                                 \
  }
  switch (t.eng.logn) { CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) }
#undef CASE
  return -3;
}

extern "C" int emu_lowpass_frames(int precision, int n_fft, int hop, const float* in, const int64_t* in_off,
                                  const int32_t* len, const int32_t* cut, const int64_t* frame_off, int n_items,
                                  int pairs_per_chunk, int n_chunks, const float* re_in, const float* im_in,
                                  float* frames) {
  if (precision == 1)
    return emu_lowpass_frames_t<double>(n_fft, hop, in, in_off, len, cut, frame_off, n_items, pairs_per_chunk,
                                        n_chunks, re_in, im_in, frames);
  return emu_lowpass_frames_t<float>(n_fft, hop, in, in_off, len, cut, frame_off, n_items, pairs_per_chunk, n_chunks,
                                     re_in, im_in, frames);
}

// wave-autonomous low-pass / ISTFT frames kernel (ssr_lowpass_wave.h; 2048-point plans); paired: one segment per frame pair
template <typename T>
static int emu_lowpass_wave_t(int hop, int split, int paired, int interleave, const float* in, const int64_t* in_off, const int32_t* len,
                              const int32_t* cut, const int64_t* frame_off, int n_items, int pairs_per_chunk, int n_chunks,
                              const float* re_in, const float* im_in, float* frames) {
  SsrTables<T> t;
  if (!ssr_build_tables<T>(2048, t)) return -3;
  SsrLowpassParams<T> p{};
  p.in = in; p.in_off = in_off; p.len = len; p.cut = cut; p.frame_off = frame_off;
  p.n_fft = 2048; p.hop = hop; p.pairs_per_chunk = pairs_per_chunk; p.n_chunks = n_chunks; p.interleave = interleave;
  if (interleave > 1 && n_chunks % interleave) return -5;
  p.window = t.window.data(); p.tw = t.tw.data();
  p.spec_re = re_in; p.spec_im = im_in; p.frames = frames;
  SsrBlk blk{64};
  for (int item = 0; item < n_items; ++item)
    for (int c = 0; c < n_chunks; ++c) {
      if (split && paired) {
        auto lds = poisoned(SsrWaveLds<T, true>::bytes());
        if (re_in) ssr_lowpass_wave_body<T, true, false, true>(p, blk, c, item, lds.data());
        else ssr_lowpass_wave_body<T, true, true, true>(p, blk, c, item, lds.data());
      } else if (split) {
        auto lds = poisoned(SsrWaveLds<T, true>::bytes());
        if (re_in) ssr_lowpass_wave_body<T, true, false, false>(p, blk, c, item, lds.data());
        else ssr_lowpass_wave_body<T, true, true, false>(p, blk, c, item, lds.data());
      } else {
        auto lds = poisoned(SsrWaveLds<T, false>::bytes());
        if (re_in) ssr_lowpass_wave_body<T, false, false, false>(p, blk, c, item, lds.data());
        else ssr_lowpass_wave_body<T, false, true, false>(p, blk, c, item, lds.data());
      }
    }
  return 0;
}
extern "C" int emu_lowpass_wave(int precision, int hop, int split, int paired, int interleave, const float* in, const int64_t* in_off,
                                const int32_t* len, const int32_t* cut, const int64_t* frame_off, int n_items,
                                int pairs_per_chunk, int n_chunks, const float* re_in, const float* im_in, float* frames) {
  if (precision == 1)
    return emu_lowpass_wave_t<double>(hop, split, paired, interleave, in, in_off, len, cut, frame_off, n_items, pairs_per_chunk, n_chunks,
                                      re_in, im_in, frames);
  return emu_lowpass_wave_t<float>(hop, split, paired, interleave, in, in_off, len, cut, frame_off, n_items, pairs_per_chunk, n_chunks, re_in,
                                   im_in, frames);
}

// fused low-pass / ISTFT with the overlap-add inside the kernel (ssr_lowpass_group.h): 512-thread "workgroups", float64 plans
extern "C" int emu_lowpass_group(int hop, const float* in, const int64_t* in_off, const int32_t* len, const int32_t* cut,
                                 const int64_t* frame_off, const int64_t* out_off, int n_items, int rounds_per_chunk, int n_chunks,
                                 const float* re_in, const float* im_in, float* out) {
  if (!ssr_lowpass_group_ok(hop)) return -4;
  SsrTables<double> t;
  if (!ssr_build_tables<double>(2048, t)) return -3;
  std::vector<double> tab((size_t)hop);
  for (int m = 0; m < hop; ++m) {
    double wss = 0.0;
    for (int mm = m + ((2048 - 1 - m) / hop) * hop; mm >= m; mm -= hop) wss += t.window[mm] * t.window[mm];
    tab[m] = wss < 1e-11 ? 1e-11 : wss;
  }
  SsrLowpassGroupParams gp{};
  gp.lp.in = in; gp.lp.in_off = in_off; gp.lp.len = len; gp.lp.cut = cut; gp.lp.frame_off = frame_off;
  gp.lp.n_fft = 2048; gp.lp.hop = hop; gp.lp.window = t.window.data(); gp.lp.tw = t.tw.data();
  gp.lp.spec_re = re_in; gp.lp.spec_im = im_in;
  std::vector<double> rtab(tab);
  for (double& v : rtab) v = 1.0 / v;
  gp.out_off = out_off; gp.out = out; gp.window64 = t.window.data(); gp.wss_tab = tab.data(); gp.wss_rcp_tab = rtab.data();
  gp.rounds_per_chunk = rounds_per_chunk; gp.n_chunks = n_chunks;
  SsrBlk blk{SSR_LG_NT};
  for (int item = 0; item < n_items; ++item)
    for (int c = 0; c < n_chunks; ++c) {
      auto lds = poisoned(ssr_lowpass_group_lds_bytes(hop));
      if (re_in) ssr_lowpass_group_body<false>(gp, blk, c, item, lds.data());
      else ssr_lowpass_group_body<true>(gp, blk, c, item, lds.data());
    }
  return 0;
}

extern "C" int emu_ola(int n_fft, int hop, int paired, const float* frames, const int64_t* frame_off, const int32_t* len,
                       const int64_t* out_off, int n_items, int max_len, float* out) {
  SsrTables<double> t;
  if (!ssr_build_tables<double>(n_fft, t)) return -3;
  std::vector<double> tab((size_t)hop);
  for (int m = 0; m < hop && hop <= n_fft; ++m) {
    double wss = 0.0;
    for (int mm = m + ((n_fft - 1 - m) / hop) * hop; mm >= m; mm -= hop) wss += t.window[mm] * t.window[mm];
    tab[m] = wss < 1e-11 ? 1e-11 : wss;
  }
  SsrOlaParams p{frames, frame_off, len, out_off, n_fft, hop, t.window.data(), out, tab.data(), 1.0f / (float)hop,
                 1.0f / (float)(2 * hop)};
  for (int item = 0; item < n_items; ++item) {
    if (paired) for (int s = 0; s < max_len; s += 4) ssr_ola_paired_quad(p, item, s);
    else for (int s = 0; s < max_len; ++s) ssr_ola_sample(p, item, s);
  }
  return 0;
}

// ---- polyphase resampler ------------------------------------------------------------------------------
template <typename S>
static int emu_resample_t(const S* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                          const int32_t* out_len, int n_items, int max_out_len, int up, int down, const S* taps, int n_taps,
                          int n_pre_remove, int groups, int taps_in_lds, S* out) {
  SsrResampleParamsT<S> p{in, in_off, in_len, out_off, out_len, up, down, taps, n_taps, n_pre_remove,
                          groups > 0 ? groups : ssr_resample_pick_groups(up, down, n_taps, sizeof(S)), taps_in_lds, out};
  if (groups < 0) {      // the window-free fallback (one output per thread)
    for (int item = 0; item < n_items; ++item)
      for (int64_t m = 0; m < max_out_len; ++m) ssr_resample_direct_output<S>(p, item, m);
    return 0;
  }
  SsrBlk blk{SSR_RESAMPLE_NT};
  const int n_blocks = (max_out_len + ssr_resample_opb(p) - 1) / ssr_resample_opb(p);
  const int total = n_items * n_blocks, n_wg = 3;                     // three "persistent workgroups" share the work
  for (int w = 0; w < n_wg; ++w) {
    auto lds = poisoned(ssr_resample_lds_bytes(p));
    ssr_resample_persistent_body<S>(p, blk, w, n_wg, total, n_blocks, lds.data());
  }
  return 0;
}
extern "C" int emu_resample(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                            const int32_t* out_len, int n_items, int max_out_len, int up, int down, const float* taps,
                            int n_taps, int n_pre_remove, int groups, int taps_in_lds, float* out) {
  return emu_resample_t<float>(in, in_off, in_len, out_off, out_len, n_items, max_out_len, up, down, taps, n_taps,
                               n_pre_remove, groups, taps_in_lds, out);
}
extern "C" int emu_resample_f64(const double* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                                const int32_t* out_len, int n_items, int max_out_len, int up, int down, const double* taps,
                                int n_taps, int n_pre_remove, int groups, int taps_in_lds, double* out) {
  return emu_resample_t<double>(in, in_off, in_len, out_off, out_len, n_items, max_out_len, up, down, taps, n_taps,
                                n_pre_remove, groups, taps_in_lds, out);
}

// matrix-core variant (ssr_resample_mfma.h); returns NB * 1000 + K (the geometry used), -2 when the plan does not fit
extern "C" int emu_resample_mfma(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                                 const int32_t* out_len, int n_items, int max_out_len, int up, int down, const float* taps,
                                 int n_taps, int n_pre_remove, int n_wg, float* out) {
  const SsrResampleMfmaGeom g = ssr_resample_mfma_geom(up, down, n_taps);
  if (!g.ok) return -2;
  SsrResampleMfmaParams p{};
  p.rp = SsrResampleParams{in, in_off, in_len, out_off, out_len, up, down, taps, n_taps, n_pre_remove, 0, 1, out};
  p.n_items_total = n_items;
  p.max_out_len = max_out_len;
  p.n_groups = (n_items + SSR_RMF_T - 1) / SSR_RMF_T;
  p.passes_per_group = (max_out_len + 32 * g.NB - 1) / (32 * g.NB);
  const int total = p.n_groups * p.passes_per_group;
  SsrBlk blk{SSR_RMF_NT};
  for (int w = 0; w < n_wg; ++w) {
    auto lds = poisoned(g.lds_bytes);
    if (g.W <= 64 * 5) ssr_resample_mfma_body<5>(p, blk, w, n_wg, total, lds.data());
    else ssr_resample_mfma_body<SSR_RMF_MAXCH>(p, blk, w, n_wg, total, lds.data());
  }
  return g.NB * 1000 + g.K;
}

// ---- windowed-sinc resampler (N2) ------------------------------------------------------------------------
extern "C" int emu_resample_sinc(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                                 const int32_t* out_len, int n_items, int max_out_len, const double* tr, const double* win,
                                 const double* delta, int n_win, int num_table, int index_step, double scale, double ratio,
                                 int phase_period, int lds_cap_floats, float* out) {
  const SsrSincGeometry g = ssr_sinc_geometry(phase_period, ratio, n_win, index_step, lds_cap_floats);
  SsrSincParams p{in, in_off, in_len, out_off, out_len, tr, win, delta, n_win, num_table, index_step, scale, out,
                  g.period, g.pw, g.m, g.max_room, g.lds_floats};
  const int bpi = (max_out_len + g.outputs_per_block - 1) / g.outputs_per_block;
  SsrBlk blk{SSR_SINC_NT};
  for (int item = 0; item < n_items; ++item)
    for (int b = 0; b < bpi; ++b) {
      auto lds = poisoned((size_t)g.lds_floats * sizeof(float));
      if (g.pad) ssr_sinc_block_body<true>(p, blk, item, b, lds.data());
      else ssr_sinc_block_body<false>(p, blk, item, b, lds.data());
    }
  return g.period * 1000000 + g.pw * 10000 + g.pad * 1000 + (g.m < 1000 ? g.m : 999);   // geometry actually used
}

// the device's phase-major table (ssr_sinc_table_body) and the per-lane-row tap loop, stated sequentially
extern "C" int emu_resample_sinc_tab(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                                     const int32_t* out_len, int n_items, const double* tr, const double* win, const double* delta,
                                     int n_win, int num_table, int index_step, double scale, float* out) {
  SsrSincParams p{in, in_off, in_len, out_off, out_len, tr, win, delta, n_win, num_table, index_step, scale, out, 1, 1, 1,
                  n_win / index_step + 1, 0};
  p.tab_r = ssr_sinc_tab_r(p.max_room);
  p.tab_rows = index_step + 1;
  std::vector<double> tab((size_t)p.tab_rows * p.tab_r * 2, -12345.0);
  p.tab = tab.data();
  for (int64_t e = 0; e < (int64_t)p.tab_rows * p.tab_r; ++e) ssr_sinc_table_body(p, e);
  for (int item = 0; item < n_items; ++item)
    for (int64_t t = 0; t < out_len[item]; ++t)
      out[out_off[item] + t] = ssr_sinc_one_tab_host(p, in + in_off[item], in_len[item], t);
  return p.tab_r;
}

// ---- zero-phase IIR (sequential statement of the wavefront kernel's arithmetic) -------------------------------
extern "C" int emu_sosfiltfilt(const float* x, const int64_t* off, const int32_t* len, int n_items, const double* sos,
                               const double* zi, int n_sections, int edge, double* fwd, double* y) {
  SsrIirParams p{x, off, len, sos, zi, n_sections, edge, n_items, fwd, y};
  for (int i = 0; i < n_items; ++i) ssr_iir_item_host(p, i);
  return 0;
}
extern "C" int emu_sosfiltfilt_f64(const double* x, const int64_t* off, const int32_t* len, int n_items, const double* sos,
                                   const double* zi, int n_sections, int edge, double* fwd, double* y) {
  SsrIirParamsT<double> p{x, off, len, sos, zi, n_sections, edge, n_items, fwd, y};
  for (int i = 0; i < n_items; ++i) ssr_iir_item_host(p, i);
  return 0;
}

// ---- cross-correlation argmax (N4) ------------------------------------------------------------------------
extern "C" int emu_xcorr_argmax(const float* a, const int64_t* a_off, const float* b, const int64_t* b_off,
                                const int32_t* len, int n_items, int max_len, int64_t* argmax_out) {
  const int nb = (int)((2 * (int64_t)max_len - 1 + SSR_XC_LAGS - 1) / SSR_XC_LAGS);
  std::vector<double> bv((size_t)n_items * nb);
  std::vector<int64_t> bi((size_t)n_items * nb);
  SsrXcorrParams p{a, a_off, b, b_off, len, nb, bv.data(), bi.data()};
  SsrBlk blk{SSR_XC_NT};
  for (int item = 0; item < n_items; ++item) {
    for (int c = 0; c < nb; ++c) {
      auto lds = poisoned(SsrXcorrLds::bytes());
      ssr_xcorr_body(p, blk, c, item, lds.data());
    }
    ssr_xcorr_pick(bv.data(), bi.data(), nb, item, argmax_out);
  }
  return 0;
}
