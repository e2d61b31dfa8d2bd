// K6, reference-arithmetic engine (SSR_LOWPASS_CONV): torchlibrosa's STFT / ISTFT evaluated the way the package evaluates them -
// as DENSE float32 DFT matrix products (ssr_eval/dsp.py:1,21-39 wraps torchlibrosa.stft.STFT / ISTFT = nn.Conv1d modules whose
// weights are the DFT x periodic-Hann matrices computed in float64 and stored float32) - on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: bitwise an ascending-k chain of float32 fused multiply-adds, MI355X_MICROARCH.md).
//
// Why it exists: a hard-low-passed signal's stop band is the transform's own round-off floor, and LSD / log-SISpec take its
// logarithm.  A float64 FFT rounded once (the default engine) puts that floor ~20 dB below a 2048-term float32 dot product, which
// moves LSD of the degraded input by 2-7 % (tests/test_oracle.py::test_lowpass_arithmetic_class_sensitivity).  This engine's floor
// is the reference's: same weights, same float32 products, float32 accumulation.  The one thing the published code leaves to the
// BLAS kernel, the accumulation order, is fixed here as chains of SSR_TL_KB = 128 fused multiply-adds (ascending k) whose results
// are added in float32 in order - a K-blocked FMA sgemm, the member of the class oracle/tl_chain.c restates bit for bit.
//
// One kernel, a "dual GEMM":  out1 = A1 . B1,  out2 = A2 . B2  on a 64 (frames) x 128 (columns) tile per workgroup of four waves,
// wave w owning columns [32 w, 32 w + 32) of all 64 rows: 2 row groups x {1, 2} = four 32 x 32 accumulator tiles.
//   forward  (STFT.forward):  A1 = A2 = the frames (rows of the reflect-padded signal, stride hop), B1 / B2 = Re / Im weights
//                             [n_fft][bins]; the lane that ends with re[t][k] also holds im[t][k], so spectrogram_phase, the cut and
//                             mag * cos / mag * sin (dsp.py:76-81,112-116; lowpass.py:24-25) run in the epilogue, which stores the
//                             NON-ZERO channels of the Hermitian-mirrored full spectrum compacted in ascending channel order
//                             (ISTFT._get_full_stft): K = cut + min(cut - 1, n_fft/2 - 1) instead of n_fft.
//   inverse  (ISTFT.forward): A1 / A2 = those compact real / imaginary rows, B1 / B2 = the matching ROWS of the transposed inverse
//                             tables [n_fft (channel)][n_fft (sample)]; out = out1 - out2 (s_real = conv_real(.) - conv_imag(.)),
//                             one windowed time frame per row.  All-zero channels are skipped: they add exact zeros.
// k_tl_fold then overlap-adds (F.fold: for one output sample the frames are added in DESCENDING frame order), divides by the folded
// hann^2 (float32, same order) clamped at 1e-11 and trims.
//
// Staging per K chunk of 16: B tiles through LDS-DMA (global_load_lds_dwordx4: a wave instruction moves two 128-float rows, the LDS
// image is linear and the column reads are conflict-free), A tiles through registers into a stride-17 array (32 rows on 32 banks);
// two stages, one barrier per chunk (32 MFMAs per wave); per-item cut / frame counts come from device arrays, so tiles past an
// item's frames or bins exit early.  Block order: column tile slowest, so that the workgroups resident on an XCD share one B tile
// (<= 2 MB) in its L2 while the A tiles stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SSR_TL_BM 64
#define SSR_TL_BN 128
#define SSR_TL_BK 16
#define SSR_TL_KB 128          /* chain length (terms per fused-multiply-add chain); a multiple of SSR_TL_BK */
#define SSR_TL_LDA (SSR_TL_BK + 1)
#define SSR_TL_NT 256

typedef float ssr_f32x16 __attribute__((ext_vector_type(16)));

struct SsrTlParams {
  // batch description (device arrays)
  const int32_t* len;          // samples per item
  const int32_t* cut;          // first zeroed bin per item, or nullptr = no cut (n_bins)
  const int64_t* frame_off;    // first row of item i in every [total_rows, *] matrix
  int n_fft, hop, n_bins, n_items, m_tiles;   // m_tiles = ceil(max frames / 64)
  int pad;                     // samples padded on each side: n_fft / 2 (center = True), 0 (center = False)
  int pad_reflect;             // 1: F.pad(mode="reflect"), 0: "constant" (zeros)
  // forward
  const float* xpad;           // reflect-padded signals; item i starts at ssr_tl_pad_off(...)
  int64_t pad_stride;          // > 0: item i's padded signal starts at i * pad_stride; 0: at frame_off[i] * hop + i * n_fft
  const float* wre_t;          // [n_fft][ldw]  Re weights transposed (row = sample, column = bin)
  const float* wim_t;
  int ldw;
  float* spec_re;              // forward out / inverse in: compact full-spectrum rows [total_rows][n_fft]
  float* spec_im;
  float* out_re;               // MODE_STFT: plain [total_rows][n_bins] outputs
  float* out_im;
  // inverse
  const float* ire_t;          // [n_fft (channel)][n_fft (sample)]
  const float* iim_t;
  float* frames;               // [total_rows][n_fft]
};

enum { SSR_TL_FWD_LOWPASS = 0, SSR_TL_FWD_STFT = 1, SSR_TL_INV = 2 };

__device__ __forceinline__ int ssr_tl_frames_of(int len, int n_fft, int hop, int pad) { return 1 + (len + 2 * pad - n_fft) / hop; }
// items the transform cannot frame are skipped (the fold kernel zeroes their output): torch's reflect padding refuses len <= pad,
// and a frame needs n_fft samples
__device__ __forceinline__ bool ssr_tl_item_ok(int len, int n_fft, int pad, int pad_reflect) {
  return (!pad_reflect || pad == 0 || len > pad) && len + 2 * pad >= n_fft && len >= 1;
}
// non-zero channels of the mirrored full spectrum for a cut: bins [0, c) and the mirrors of bins 1 .. mmax
__device__ __forceinline__ int ssr_tl_mmax(int c, int n_fft) { const int m = c - 1 < n_fft / 2 - 1 ? c - 1 : n_fft / 2 - 1; return m < 0 ? 0 : m; }

__device__ __forceinline__ int64_t ssr_tl_pad_off(int64_t pad_stride, int64_t row0, int hop, int item, int n_fft) {
  return pad_stride > 0 ? (int64_t)item * pad_stride : row0 * hop + (int64_t)item * n_fft;      // (T hop + n_fft >= len + 2 pad for every item)
}

// LDS-DMA, 16 bytes per lane: lane l's four floats land at LDS byte address `lds_wave_base` + 16 l (M0 = the wave-uniform base).
// Inline asm rather than __builtin_amdgcn_global_load_lds: with the builtin in flight the compiler's wait-count pass makes EVERY LDS
// wait of the loop a wait for all outstanding LDS operations (see ssr_resample_rc.h: ssr_lds_dma_dword); the kernel orders the
// transfer itself (s_waitcnt vmcnt(0) before the barrier that publishes the stage).
__device__ __forceinline__ void ssr_tl_glds16(const float* src, char* lds_wave_base) {
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(base), "v"(src) : "memory");
}

// separately rounded float32 operations (hipcc's default -ffp-contract=fast would fuse a * b + c; torch's tensor ops round each)
__device__ __forceinline__ float ssr_tl_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float ssr_tl_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}

template <int MODE>
__device__ __forceinline__ void ssr_tl_gemm_body(const SsrTlParams& p, char* smem) {
  constexpr bool INV = MODE == SSR_TL_INV;
  constexpr int A_FLOATS = SSR_TL_BM * SSR_TL_LDA, B_FLOATS = SSR_TL_BK * SSR_TL_BN;
  constexpr int STAGE_FLOATS = (INV ? 2 : 1) * A_FLOATS + 2 * B_FLOATS;
  float* lds = (float*)smem;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // block -> (column tile, item, row tile), row tile fastest
  const int per_n = p.n_items * p.m_tiles;
  const int ntile = (int)blockIdx.x / per_n, rest = (int)blockIdx.x % per_n;
  const int item = rest / p.m_tiles, mtile = rest % p.m_tiles;
  const int len = p.len[item];
  if (!ssr_tl_item_ok(len, p.n_fft, p.pad, p.pad_reflect)) return;   // (entry-point contract: skipped, output zeroed by the fold kernel)
  const int T = ssr_tl_frames_of(len, p.n_fft, p.hop, p.pad);
  const int m0 = mtile * SSR_TL_BM;
  if (m0 >= T) return;
  int c = p.cut ? p.cut[item] : p.n_bins;
  c = c < 0 ? 0 : (c > p.n_bins ? p.n_bins : c);
  const int mmax = ssr_tl_mmax(c, p.n_fft);
  const int n0 = ntile * SSR_TL_BN;
  const int n_cols = INV ? p.n_fft : (MODE == SSR_TL_FWD_STFT ? p.n_bins : c);
  if (n0 >= n_cols) return;
  const int K = INV ? c + mmax : p.n_fft;
  const int64_t row0 = p.frame_off[item];
  const bool wave_on = n0 + 32 * wave < n_cols;                   // a wave whose 32 columns lie past the item's bins only stages

  // ---- staging assignments ------------------------------------------------------------------------------------
  // A: thread -> k = tid % 16, rows tid / 16 + 16 j
  const int ak = tid & 15, ar = tid >> 4;
  const float* a1p[4];
  const float* a2p[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int t = m0 + ar + 16 * j;
    t = t < T ? t : T - 1;
    if (INV) {
      a1p[j] = p.spec_re + (row0 + t) * (int64_t)p.n_fft + ak;
      a2p[j] = p.spec_im + (row0 + t) * (int64_t)p.n_fft + ak;
    } else {
      a1p[j] = p.xpad + ssr_tl_pad_off(p.pad_stride, row0, p.hop, item, p.n_fft) + (int64_t)t * p.hop + ak;
      a2p[j] = nullptr;
    }
  }
  // B: wave instruction q = 4 wave + i moves rows 2 q' + lane / 32 of B1 (q < 8) or B2; lane covers 4 columns
  const int brow = lane >> 5, bcol = (lane & 31) * 4;
  const float* b1 = INV ? p.ire_t : p.wre_t;
  const float* b2 = INV ? p.iim_t : p.wim_t;
  const int ldb = INV ? p.n_fft : p.ldw;

  ssr_f32x16 acc1[2], acc2[2], tot1[2], tot2[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[g][r] = 0.0f; acc2[g][r] = 0.0f; tot1[g][r] = 0.0f; tot2[g][r] = 0.0f; }

  float ra1[4], ra2[4];
  auto issue = [&](int chunk, int stage) {
    float* st = lds + stage * STAGE_FLOATS;
    float* bs = st + (INV ? 2 : 1) * A_FLOATS;
    const int k0 = chunk * SSR_TL_BK;
#if defined(SSR_TL_EXP_NOB)             /* developer experiment (timing only, wrong results): B staged for the first chunks only */
    if (chunk < 2)
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = 4 * wave + i, which = q >> 3, r = 2 * (q & 7) + brow;
      int kk = k0 + r;
      kk = kk < K ? kk : K - 1;
      const int ch = INV ? (kk < c ? kk : p.n_fft - mmax + (kk - c)) : kk;
      const float* src = (which ? b2 : b1) + (int64_t)ch * ldb + n0 + bcol;
      ssr_tl_glds16(src, (char*)(bs + which * B_FLOATS + 2 * (q & 7) * SSR_TL_BN));
    }
    const bool in_k = k0 + ak < K;
#if defined(SSR_TL_EXP_NOA)             /* developer experiment (timing only, wrong results): A loaded for the first chunks only */
    if (chunk < 2)
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ra1[j] = in_k ? a1p[j][k0] : 0.0f;
      if (INV) ra2[j] = in_k ? a2p[j][k0] : 0.0f;
    }
  };
  auto park = [&](int stage) {
    float* st = lds + stage * STAGE_FLOATS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st[(ar + 16 * j) * SSR_TL_LDA + ak] = ra1[j];
      if (INV) st[A_FLOATS + (ar + 16 * j) * SSR_TL_LDA + ak] = ra2[j];
    }
  };

  const int n_chunks = (K + SSR_TL_BK - 1) / SSR_TL_BK;
  if (n_chunks > 0) {
    issue(0, 0);
    park(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int fi = lane & 31, fk = lane >> 5;
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const int stage = chunk & 1;
    if (chunk + 1 < n_chunks) issue(chunk + 1, stage ^ 1);
    if (wave_on) {
      const float* st = lds + stage * STAGE_FLOATS;
      const float* as1 = st;
      const float* as2 = INV ? st + A_FLOATS : st;
      const float* bs1 = st + (INV ? 2 : 1) * A_FLOATS + 32 * wave + fi;
      const float* bs2 = bs1 + B_FLOATS;
      // One step = one k pair = four matrix instructions (256 cycles of the matrix pipe).  The operands of step s + 1 are requested
      // BEFORE the instructions of step s are issued and first touched after them: the LDS round trip runs under a full step of
      // matrix work (left to itself the compiler put every read right in front of its use - a wait per one or two instructions).
      float fb1[2], fb2[2], fa10[2], fa11[2], fa20[2], fa21[2];
      auto fetch = [&](int s, int slot) {
        const int k = 2 * s + fk;
        fb1[slot] = bs1[k * SSR_TL_BN]; fb2[slot] = bs2[k * SSR_TL_BN];
        fa10[slot] = as1[fi * SSR_TL_LDA + k]; fa11[slot] = as1[(fi + 32) * SSR_TL_LDA + k];
        if (INV) { fa20[slot] = as2[fi * SSR_TL_LDA + k]; fa21[slot] = as2[(fi + 32) * SSR_TL_LDA + k]; }
      };
      fetch(0, 0);
#pragma unroll
      for (int s = 0; s < SSR_TL_BK / 2; ++s) {
        const int cur = s & 1;
        if (s + 1 < SSR_TL_BK / 2) fetch(s + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const float a20 = INV ? fa20[cur] : fa10[cur], a21 = INV ? fa21[cur] : fa11[cur];
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa10[cur], fb1[cur], acc1[0], 0, 0, 0);
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a20, fb2[cur], acc2[0], 0, 0, 0);
        acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa11[cur], fb1[cur], acc1[1], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a21, fb2[cur], acc2[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if ((chunk + 1) % (SSR_TL_KB / SSR_TL_BK) == 0 || chunk + 1 == n_chunks) {   // a chain ends: total += chain, restart from 0
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            tot1[g][r] = ssr_tl_add(tot1[g][r], acc1[g][r]);
            tot2[g][r] = ssr_tl_add(tot2[g][r], acc2[g][r]);
            acc1[g][r] = 0.0f;
            acc2[g][r] = 0.0f;
          }
      }
    }
    if (chunk + 1 < n_chunks) park(stage ^ 1);       // (the loads' results are first touched here: they had the whole chunk to land)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA of this wave has landed before anyone passes the barrier
    __syncthreads();
  }
  if (!wave_on) return;

  // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -------------------------------
  const int col = n0 + 32 * wave + fi;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = m0 + 32 * g + (r & 3) + 8 * (r >> 2) + 4 * fk;
      if (t >= T) continue;
      const int64_t row = row0 + t;
      const float v1 = tot1[g][r], v2 = tot2[g][r];
      if (MODE == SSR_TL_INV) {
        p.frames[row * p.n_fft + col] = v1 - v2;
      } else if (MODE == SSR_TL_FWD_STFT) {
        if (col < p.n_bins) {
          p.out_re[row * p.n_bins + col] = v1;
          p.out_im[row * p.n_bins + col] = v2;
        }
      } else if (col < c) {
        // spectrogram_phase (dsp.py:76-81) with eps = 1e-8 (dsp.py:83), then mag * cos, mag * sin (dsp.py:112-116): float32, every
        // operation rounded on its own as torch's separate tensor ops are
        const float aa = ssr_tl_mul(v1, v1), bb = ssr_tl_mul(v2, v2);
        float ss = ssr_tl_add(aa, bb);
        ss = ss < 1e-8f ? 1e-8f : ss;
        const float mag = sqrtf(ss);                  // correctly rounded (hipcc's default; __fsqrt_rn is the 1-ulp native one)
        const float cs = v1 / mag, sn = v2 / mag;
        const float R = ssr_tl_mul(mag, cs), I = ssr_tl_mul(mag, sn);
        float* sr = p.spec_re + row * p.n_fft;
        float* si = p.spec_im + row * p.n_fft;
        sr[col] = R;
        si[col] = I;
        if (col >= 1 && col <= mmax) {               // the mirrored channel n_fft - col sits at compact index c + mmax - col
          sr[c + mmax - col] = R;
          si[c + mmax - col] = -I;
        }
      }
    }
}

// Padding as torchlibrosa's STFT.forward does it: F.pad(x, (n_fft/2, n_fft/2), mode = "reflect" | "constant") when center, none otherwise.
struct SsrTlPadParams {
  const float* in; const int64_t* in_off; const int32_t* len; const int64_t* frame_off;
  int n_fft, hop; float* xpad; int64_t pad_stride; int pad, pad_reflect;
};
__device__ __forceinline__ void ssr_tl_pad_body(const SsrTlPadParams& p, int item, int64_t i) {
  const int len = p.len[item];
  if (!ssr_tl_item_ok(len, p.n_fft, p.pad, p.pad_reflect) || i >= (int64_t)len + 2 * p.pad) return;
  int64_t s = i - p.pad;
  float v;
  if (s >= 0 && s < len) v = p.in[p.in_off[item] + s];
  else if (!p.pad_reflect) v = 0.0f;                               // F.pad(mode="constant")
  else {
    if (s < 0) s = -s;
    if (s >= len) s = 2 * ((int64_t)len - 1) - s;
    v = p.in[p.in_off[item] + s];
  }
  p.xpad[ssr_tl_pad_off(p.pad_stride, p.frame_off[item], p.hop, item, p.n_fft) + i] = v;
}

// ISTFT given (re, im) [rows][n_bins]: build the compact mirrored rows (cut = n_bins: every channel).
struct SsrTlPackParams {
  const float* re; const float* im; int64_t total_rows; int n_fft, n_bins; float* spec_re; float* spec_im;
};
__device__ __forceinline__ void ssr_tl_pack_body(const SsrTlPackParams& p, int64_t idx) {
  const int64_t row = idx / p.n_fft;
  const int j = (int)(idx % p.n_fft);
  if (row >= p.total_rows) return;
  const int c = p.n_bins, mmax = p.n_fft / 2 - 1;
  float r, i;
  if (j < c) { r = p.re[row * p.n_bins + j]; i = p.im[row * p.n_bins + j]; }
  else { const int src = mmax - (j - c); r = p.re[row * p.n_bins + src]; i = -p.im[row * p.n_bins + src]; }
  p.spec_re[idx] = r;
  p.spec_im[idx] = i;
}

// F.fold + / clamp(folded hann^2, 1e-11) + trim (ISTFT._overlap_add_divide_window_sum, _trim_edges): one thread per output sample.
struct SsrTlFoldParams {
  const float* frames; const int64_t* frame_off; const int32_t* len; const int64_t* out_off; int n_fft, hop;
  const float* w2; float* out; int pad, pad_reflect;
};
__device__ __forceinline__ void ssr_tl_fold_body(const SsrTlFoldParams& p, int item, int s) {
  const int len = p.len[item];
  if (s >= len) return;
  float* out = p.out + p.out_off[item];
  if (!ssr_tl_item_ok(len, p.n_fft, p.pad, p.pad_reflect)) { out[s] = 0.0f; return; }
  const int T = ssr_tl_frames_of(len, p.n_fft, p.hop, p.pad);
  const int q = s + p.pad;                             // ISTFT._trim_edges: start = n_fft / 2 if center else 0
  int t = q / p.hop;
  t = t < T - 1 ? t : T - 1;
  const float* fr = p.frames + p.frame_off[item] * (int64_t)p.n_fft;
  if (q >= (T - 1) * p.hop + p.n_fft) { out[s] = 0.0f; return; }     // past the overlap-added signal (the slice ends there; zero-filled)
  float y = 0.0f, ws = 0.0f;
  for (; t >= 0 && q - t * p.hop < p.n_fft; --t) {
    y = ssr_tl_add(y, fr[(int64_t)t * p.n_fft + (q - t * p.hop)]);
    ws = ssr_tl_add(ws, p.w2[q - t * p.hop]);
  }
  ws = ws < 1e-11f ? 1e-11f : ws;
  out[s] = y / ws;
}
