// K6, reference-arithmetic engine (SSR_LOWPASS_CONV): torchlibrosa's STFT / ISTFT evaluated the way the package evaluates them -
// as DENSE float32 DFT matrix products (ssr_eval/dsp.py:1,21-39 wraps torchlibrosa.stft.STFT / ISTFT = nn.Conv1d modules whose
// weights are the DFT x window matrices computed in float64 and stored float32) - on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: bitwise an ascending-k chain of float32 fused multiply-adds, MI355X_MICROARCH.md).
//
// Why it exists: a hard-low-passed signal's stop band is the transform's own round-off floor, and LSD / log-SISpec take its
// logarithm.  A float64 FFT rounded once puts that floor ~20 dB below a 2048-term float32 dot product, which moves LSD of the
// degraded input by 2-7 %.  This engine's floor is the reference's: same weights, same float32 products, float32 accumulation
// IN THE ORDER torch-CPU's F.conv1d runs them (oneDNN, AVX-512 host, >= 2 threads; established bit for bit on CPU,
// tests/test_oracle.py::test_tl_chain_is_torch_conv1d_bit_for_bit, and restated in oracle/tl_chain.c):
//   forward  (STFT.forward, strided conv, signals of >= 55 frames): ONE ascending chain of n_fft fused multiply-adds per output;
//   inverse  (ISTFT.forward, two 1x1 convs over the n_fft channels of the mirrored spectrum): one chain per BLOCK OF
//            SSR_TL_KBF = 256 CHANNELS of the full spectrum (channel / 256), each from 0, the blocks' results added to a float32
//            total in ascending block order.  All-zero channels (the cut) add exact zeros and are skipped.
//
// Two kernels, both "dual GEMMs" (out1 = A1 . B1, out2 = A2 . B2), K in chunks of 16, operands double-buffered in LDS:
//   k_tl_fwd: A1 = A2 = the frames (rows of the padded signal at stride hop, read where they lie), B1 / B2 = Re / Im weights
//            [n_fft][bins].  128 (frames) x 128 (bins) per workgroup of four waves, wave = 64 x 64 of both products: eight 32 x 32
//            accumulators, no chain totals (one chain); a tile with at most 64 bins left (the last one of most cuts) is split 4 x 1:
//            wave = 32 x 64, half the matrix instructions, no idle wave.  The lane that ends with re[t][k] also holds im[t][k], so
//            spectrogram_phase, the cut and mag * cos / mag * sin (dsp.py:76-81,112-116; lowpass.py:24-25) run in the epilogue,
//            which stores (R, I) TRANSPOSED - [bin][row], four consecutive rows per lane and store - so that the inverse product
//            can stream its A tiles with LDS-DMA.  The mirrored half of the spectrum is never materialised.
//   k_tl_inv<BM>: A1 / A2 = rows [channel][frame] of that transposed spectrum (channel ch > n_fft/2 reads bin n_fft - ch; the
//            sign of the mirrored imaginary part lives in the weight table: rows ch > n_fft/2 of the device copy of the Im table are
//            negated, (-I) w == I (-w) bit for bit), B1 / B2 = rows ch of the transposed inverse tables [n_fft (channel)][n_fft
//            (sample)]; out = out1 - out2 (s_real = conv_real(.) - conv_imag(.)), one windowed time frame per row.
//            BM (64 | 128) x 128 per workgroup of BM / 16 waves, wave = 64 x 32 of both products + the chain totals.
//            EVERY operand tile arrives by LDS-DMA (global_load_lds_dwordx4: linear LDS images [k][row] and [k][col], conflict-free
//            fragment reads, no register staging, no ds_write); a chunk never straddles a 256-channel block (block tails are
//            padded with a zero row of the spectrum buffer: exact zeros), so a chain ends on a chunk boundary.
// k_tl_fold then overlap-adds (F.fold: for one output sample the frames are added in DESCENDING frame order), divides by the folded
// window^2 (float32, same order) clamped at 1e-11 and trims.
//
// Row tiles: with one cut for the whole launch (ssr_fft_lowpass_multi, ssr_istft) the tiles run over the batch's rows
// [0, total_rows) regardless of item boundaries (no ragged last tile per item); with per-item cuts (ssr_fft_lowpass) a tile
// belongs to one item and starts at a multiple of 4 rows (16-byte DMA sources), rows of the neighbours masked.
// Block order: blocks b, b + 8, ... run on one XCD (observed; used for speed only): each XCD works through "super tiles" of 4 row
// tiles x all column tiles, so that the workgroups resident on it share their A and B chunks in its L2 as they move along K.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#define SSR_TL_BN 128
#define SSR_TL_BK 16
#define SSR_TL_KBF 256         /* inverse: channels of the full spectrum per chain block */
#define SSR_TL_FWD_BM 128
#define SSR_TL_FWD_NT 256
#define SSR_TL_LDA (SSR_TL_BK + 1)

typedef float ssr_f32x16 __attribute__((ext_vector_type(16)));
typedef float ssr_f32x4 __attribute__((ext_vector_type(4)));

struct SsrTlParams {
  // batch description (device arrays)
  const int32_t* len;          // samples per item
  const int32_t* cut;          // per-item cuts (first zeroed bin), or nullptr with uniform_cut
  const int64_t* frame_off;    // first row of item i in every [total_rows, *] matrix
  int n_fft, hop, n_bins, n_items;
  int m_tiles;                 // per-item cuts: row tiles per item; uniform cut: row tiles of the batch
  int bm;                      // rows per tile of this launch
  int pad;                     // samples padded on each side: n_fft / 2 (center = True), 0 (center = False)
  int pad_reflect;             // 1: F.pad(mode="reflect"), 0: "constant" (zeros)
  int uniform_cut;             // >= 0: one cut for every row of the launch, row tiles over [0, total_rows); < 0: cut[item]
  int64_t total_rows;
  // forward
  const float* xpad;           // padded signals; item i starts at ssr_tl_pad_off(...)
  int64_t pad_stride;          // > 0: item i's padded signal starts at i * pad_stride; 0: at frame_off[i] * hop + i * n_fft
  const int64_t* rowbase;      // uniform cut: offset of row r's first sample in xpad (k_tl_pad writes it); else nullptr
  const float* wre_t;          // [n_fft][ldw]  Re weights transposed (row = sample, column = bin)
  const float* wim_t;
  int ldw;
  float* spec_re;              // forward out / inverse in: TRANSPOSED low-passed spectrum [n_bins][ldt] (row = bin, column = frame row)
  float* spec_im;
  const float* zero_row;       // [ldt] zeros: the A source of padded chunk slots
  int64_t ldt;
  float* out_re;               // SSR_TL_FWD_STFT: plain [total_rows][n_bins] outputs
  float* out_im;
  // inverse
  const float* ire_t;          // [n_fft (channel)][n_fft (sample)]
  const float* iim_t;          // rows ch > n_fft / 2 negated
  float* frames;               // [total_rows][n_fft]
};

enum { SSR_TL_FWD_LOWPASS = 0, SSR_TL_FWD_STFT = 1 };

__device__ __forceinline__ int ssr_tl_frames_of(int len, int n_fft, int hop, int pad) { return 1 + (len + 2 * pad - n_fft) / hop; }
// items the transform cannot frame are skipped (the fold kernel zeroes their output): torch's reflect padding refuses len <= pad,
// and a frame needs n_fft samples
__device__ __forceinline__ bool ssr_tl_item_ok(int len, int n_fft, int pad, int pad_reflect) {
  return (!pad_reflect || pad == 0 || len > pad) && len + 2 * pad >= n_fft && len >= 1;
}
// mirrored bins 1 .. mmax of a cut c (channels n_fft - mmax .. n_fft - 1 of the full spectrum)
__device__ __host__ __forceinline__ int ssr_tl_mmax(int c, int n_fft) { const int m = c - 1 < n_fft / 2 - 1 ? c - 1 : n_fft / 2 - 1; return m < 0 ? 0 : m; }

__device__ __forceinline__ int64_t ssr_tl_pad_off(int64_t pad_stride, int64_t row0, int hop, int item, int n_fft) {
  return pad_stride > 0 ? (int64_t)item * pad_stride : row0 * hop + (int64_t)item * n_fft;      // (T hop + n_fft >= len + 2 pad for every item)
}

// LDS-DMA, 16 bytes per lane: lane l's four floats land at LDS byte address `lds_wave_base` + 16 l (M0 = the wave-uniform base).
// Inline asm rather than __builtin_amdgcn_global_load_lds: with the builtin in flight the compiler's wait-count pass makes EVERY LDS
// wait of the loop a wait for all outstanding LDS operations (see ssr_resample_rc.h: ssr_lds_dma_dword); the kernel orders the
// transfer itself (s_waitcnt vmcnt before the barrier that publishes the stage).
__device__ __forceinline__ void ssr_tl_glds16(const float* src, unsigned lds_byte_off) {
  const unsigned base = __builtin_amdgcn_readfirstlane(lds_byte_off);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(base), "v"(src) : "memory");
}
__device__ __forceinline__ unsigned ssr_tl_lds_off(const float* lds) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
}
template <int N> __device__ __forceinline__ void ssr_tl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// separately rounded float32 operations (hipcc's default -ffp-contract=fast would fuse a * b + c; torch's tensor ops round each)
__device__ __forceinline__ float ssr_tl_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float ssr_tl_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}

// ---- block -> (row tile, column tile) -------------------------------------------------------------------------------------------
// Blocks b, b + 8, ... share an XCD (observed; used for speed only).  Each XCD works through "super tiles" of SSR_TL_RW row tiles x
// ALL column tiles, row tile fastest: the workgroups resident on it at one time share A chunks (same row tile, different column
// tiles) and B chunks (same column tile, different row tiles) in its L2 as they move along K; super tile g runs on XCD g % 8, so
// every XCD gets the same number of equal super tiles.  The host launches ssr_tl_grid(Mt, Nt) blocks; blocks past the matrix exit.
#define SSR_TL_RW 4
__host__ inline int64_t ssr_tl_grid(int64_t Mt, int Nt) {
  const int64_t supers = (Mt + SSR_TL_RW - 1) / SSR_TL_RW;
  return ((supers + 7) / 8) * 8 * SSR_TL_RW * Nt;
}
__device__ __forceinline__ bool ssr_tl_block_tile(int b, int Mt, int Nt, int& mt, int& ct) {
  const int xcd = b & 7, j = b >> 3, per = SSR_TL_RW * Nt;
  const int g = (j / per) * 8 + xcd, jj = j % per;
  mt = g * SSR_TL_RW + jj % SSR_TL_RW;
  ct = jj / SSR_TL_RW;
  return mt < Mt;
}

// ---- row tile -> rows -------------------------------------------------------------------------------------------------------------
struct SsrTlTile {
  int64_t lo;        // first row the tile's operands cover (a multiple of 4)
  int64_t v_lo, v_hi;   // rows the tile owns: [v_lo, v_hi)
  int c;             // cut of those rows (clamped to [0, n_bins])
  int item;          // per-item cuts: the item; uniform: -1
  int64_t row0;      // per-item cuts: the item's first row
};
__device__ __forceinline__ bool ssr_tl_tile(const SsrTlParams& p, int mt, int BM, SsrTlTile& t) {
  int c;
  if (p.uniform_cut >= 0) {
    t.item = -1; t.row0 = 0;
    t.lo = (int64_t)mt * BM;
    t.v_lo = t.lo;
    t.v_hi = t.lo + BM < p.total_rows ? t.lo + BM : p.total_rows;
    c = p.uniform_cut;
  } else {
    const int item = mt / p.m_tiles, mtile = mt % p.m_tiles;
    const int len = p.len[item];
    if (!ssr_tl_item_ok(len, p.n_fft, p.pad, p.pad_reflect)) return false;   // (entry-point contract: skipped, output zeroed by the fold kernel)
    const int T = ssr_tl_frames_of(len, p.n_fft, p.hop, p.pad);
    t.item = item; t.row0 = p.frame_off[item];
    t.lo = (t.row0 & ~(int64_t)3) + (int64_t)mtile * BM;
    t.v_lo = t.lo > t.row0 ? t.lo : t.row0;
    t.v_hi = t.lo + BM < t.row0 + T ? t.lo + BM : t.row0 + T;
    c = p.cut ? p.cut[item] : p.n_bins;
  }
  t.c = c < 0 ? 0 : (c > p.n_bins ? p.n_bins : c);
  return t.v_lo < t.v_hi;
}

// =====================================================================================================================================
// forward
// =====================================================================================================================================
template <int MODE>
__device__ __forceinline__ void ssr_tl_fwd_body(const SsrTlParams& p, float* lds) {
  constexpr int BM = SSR_TL_FWD_BM, NT = SSR_TL_FWD_NT;
  constexpr int A_FLOATS = BM * SSR_TL_LDA, B_FLOATS = SSR_TL_BK * SSR_TL_BN;
  constexpr int STAGE_FLOATS = A_FLOATS + 2 * B_FLOATS;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;              // wave = rows [64 wr, +64) x columns [64 wc, +64)
  const int Mt = p.uniform_cut >= 0 ? p.m_tiles : p.m_tiles * p.n_items;
  const int Nt_all = (p.n_bins + SSR_TL_BN - 1) / SSR_TL_BN;
  const int Nt = MODE == SSR_TL_FWD_STFT || p.uniform_cut < 0 ? Nt_all : (p.uniform_cut + SSR_TL_BN - 1) / SSR_TL_BN;
  int mt, ct;
  if (!ssr_tl_block_tile((int)blockIdx.x, Mt, Nt, mt, ct)) return;
  SsrTlTile tl;
  if (!ssr_tl_tile(p, mt, BM, tl)) return;
  const int c = tl.c;
  const int n0 = ct * SSR_TL_BN;
  const int n_cols = MODE == SSR_TL_FWD_STFT ? p.n_bins : c;
  if (n0 >= n_cols) return;
  // Wave layout of the tile: 2 x 2 waves of 64 rows x 64 columns - or, when the tile holds no more than 64 columns (the last column
  // tile of most cuts: 683 = 5 x 128 + 43), 4 x 1 waves of 32 rows x 64 columns: half the matrix instructions per wave, no idle wave
  const bool narrow = n_cols - n0 <= 64;
  const int row0 = narrow ? 32 * wave : 64 * wr, col0 = narrow ? 0 : 64 * wc;
  const bool wave_on = n0 + col0 < n_cols;             // a wave whose 64 columns lie past the bins only stages

  // ---- staging assignments ------------------------------------------------------------------------------------
  // A: thread -> k = tid % 16, rows tid / 16 + 16 j (j < 8); rows outside the tile's own range re-read a row inside it (never stored)
  const int ak = tid & 15, ar = tid >> 4;
  const float* ap[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int64_t r = tl.lo + ar + 16 * j;
    r = r < tl.v_lo ? tl.v_lo : (r >= tl.v_hi ? tl.v_hi - 1 : r);
    const int64_t base = p.uniform_cut >= 0 ? p.rowbase[r]
                                            : ssr_tl_pad_off(p.pad_stride, tl.row0, p.hop, tl.item, p.n_fft) + (r - tl.row0) * p.hop;
    ap[j] = p.xpad + base + ak;
  }
  // B: wave instruction q = 4 wave + i moves rows 2 (q % 8) + lane / 32 of B1 (q < 8) or B2; a lane covers 4 columns
  const unsigned lds0 = ssr_tl_lds_off(lds);
  const float* bp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = 4 * wave + i;
    bp[i] = ((q >> 3) ? p.wim_t : p.wre_t) + (int64_t)(2 * (q & 7) + (lane >> 5)) * p.ldw + n0 + (lane & 31) * 4;
  }
  const int64_t bstep = (int64_t)SSR_TL_BK * p.ldw;

  ssr_f32x16 acc1[2][2], acc2[2][2];                    // [row group][column half]
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc1[g][h][r] = 0.0f; acc2[g][h][r] = 0.0f; }

  float ra[8];
  // B transfer i (of this wave's four) of `chunk` into `stage`; A: the thread's eight samples of `chunk` into registers, parked later
  auto issue_b = [&](int i, int chunk, int stage) {
    const int q = 4 * wave + i;
    ssr_tl_glds16(bp[i] + chunk * bstep,
                  lds0 + 4u * (unsigned)(stage * STAGE_FLOATS + A_FLOATS + (q >> 3) * B_FLOATS + 2 * (q & 7) * SSR_TL_BN));
  };
  auto load_a = [&](int j, int chunk) { ra[j] = ap[j][chunk * SSR_TL_BK]; };
  auto park = [&](int stage) {
    float* st = lds + stage * STAGE_FLOATS;
#pragma unroll
    for (int j = 0; j < 8; ++j) st[(ar + 16 * j) * SSR_TL_LDA + ak] = ra[j];
  };

  const int fi = lane & 31, fk = lane >> 5;
  // operands of one step (one k pair = eight matrix instructions, 512 cycles of the matrix pipe): two register sets, the set of step
  // s + 1 is requested BEFORE the matrix instructions of step s are issued and first touched after them
  float fa0[2], fa1[2], fb10[2], fb11[2], fb20[2], fb21[2];
  auto fetch = [&](int stage, int s, int slot, auto narrow_tag) {
    constexpr bool NARROW = decltype(narrow_tag)::value;
    const float* st = lds + stage * STAGE_FLOATS;
    const float* as = st + (row0 + fi) * SSR_TL_LDA;
    const float* bs1 = st + A_FLOATS + col0 + fi;
    const float* bs2 = bs1 + B_FLOATS;
    const int k = 2 * s + fk;
    fa0[slot] = as[k];
    if (!NARROW) fa1[slot] = as[32 * SSR_TL_LDA + k];
    fb10[slot] = bs1[k * SSR_TL_BN]; fb20[slot] = bs2[k * SSR_TL_BN];
    fb11[slot] = bs1[k * SSR_TL_BN + 32]; fb21[slot] = bs2[k * SSR_TL_BN + 32];
  };
  auto mfma8 = [&](int cur, auto narrow_tag) {
    constexpr bool NARROW = decltype(narrow_tag)::value;
    acc1[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur], fb10[cur], acc1[0][0], 0, 0, 0);
    acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur], fb20[cur], acc2[0][0], 0, 0, 0);
    if (!NARROW) {
      acc1[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur], fb10[cur], acc1[1][0], 0, 0, 0);
      acc2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur], fb20[cur], acc2[1][0], 0, 0, 0);
    }
    acc1[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur], fb11[cur], acc1[0][1], 0, 0, 0);
    acc2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur], fb21[cur], acc2[0][1], 0, 0, 0);
    if (!NARROW) {
      acc1[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur], fb11[cur], acc1[1][1], 0, 0, 0);
      acc2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur], fb21[cur], acc2[1][1], 0, 0, 0);
    }
  };

  // ---- the pipeline: three stages, ONE barrier per chunk, in the MIDDLE of the chunk's eight steps (see k_tl_inv) -------------------
  // Second half of chunk c: the B transfers of chunk c + 2 and the thread's A samples of chunk c + 2 (into registers), one transfer and
  // two loads per step; first half of chunk c + 1: those registers are parked in LDS; barrier(c + 1) publishes both.
  const int n_chunks = p.n_fft / SSR_TL_BK;             // (n_fft = 32 m: >= 2)
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_b(i, 0, 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) load_a(j, 0);
  park(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_b(i, 1, 1);
#pragma unroll
  for (int j = 0; j < 8; ++j) load_a(j, 1);
  ssr_tl_wait_vm<12>();                                 // chunk 0's transfers have landed (chunk 1's four + eight loads may be in flight)
  __syncthreads();
  if (wave_on) fetch(0, 0, 0, std::false_type{});
  int stage = 0;
  auto chunk_body = [&](int chunk, auto more_tag, auto on_tag, auto narrow_tag) {
    constexpr bool MORE = decltype(more_tag)::value;     // chunk + 2 exists: request it in the second half
    constexpr bool ON = decltype(on_tag)::value;         // this wave multiplies (else it only stages)
    const int st1 = stage == 2 ? 0 : stage + 1, st2 = stage == 0 ? 2 : stage - 1;     // stages of chunks c + 1, c + 2
    if (ON) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        fetch(stage, s + 1, (s + 1) & 1, narrow_tag);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(s & 1, narrow_tag);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (chunk + 1 < n_chunks) {
      park(st1);                                        // (the loads were issued half a chunk and more ago)
      ssr_tl_wait_vm<0>();
      __syncthreads();
    }
#pragma unroll
    for (int s = 4; s < 8; ++s) {
      if (MORE) {
        issue_b(s - 4, chunk + 2, st2);
        load_a(2 * (s - 4), chunk + 2);
        load_a(2 * (s - 4) + 1, chunk + 2);
      }
      if (ON) {
        if (s < 7) fetch(stage, s + 1, (s + 1) & 1, narrow_tag);
        else if (chunk + 1 < n_chunks) fetch(st1, 0, 0, narrow_tag);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(s & 1, narrow_tag);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    stage = st1;
  };
  if (wave_on && !narrow) {
    int chunk = 0;
    for (; chunk + 2 < n_chunks; ++chunk) chunk_body(chunk, std::true_type{}, std::true_type{}, std::false_type{});
    for (; chunk < n_chunks; ++chunk) chunk_body(chunk, std::false_type{}, std::true_type{}, std::false_type{});
  } else if (wave_on) {
    int chunk = 0;
    for (; chunk + 2 < n_chunks; ++chunk) chunk_body(chunk, std::true_type{}, std::true_type{}, std::true_type{});
    for (; chunk < n_chunks; ++chunk) chunk_body(chunk, std::false_type{}, std::true_type{}, std::true_type{});
  } else {
    int chunk = 0;
    for (; chunk + 2 < n_chunks; ++chunk) chunk_body(chunk, std::true_type{}, std::false_type{}, std::false_type{});
    for (; chunk < n_chunks; ++chunk) chunk_body(chunk, std::false_type{}, std::false_type{}, std::false_type{});
  }
  if (!wave_on) return;

  // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -------------------------------
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int col = n0 + col0 + 32 * h + fi;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (g == 1 && narrow) continue;                                  // (32 rows per wave in the narrow layout)
        const int64_t rq = tl.lo + row0 + 32 * g + 8 * j + 4 * fk;          // rows rq .. rq + 3 (rq is a multiple of 4)
        if (MODE == SSR_TL_FWD_STFT) {
          if (col < p.n_bins)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (rq + e >= tl.v_lo && rq + e < tl.v_hi) {
                p.out_re[(rq + e) * p.n_bins + col] = acc1[g][h][4 * j + e];
                p.out_im[(rq + e) * p.n_bins + col] = acc2[g][h][4 * j + e];
              }
        } else if (col < c) {
          // spectrogram_phase (dsp.py:76-81) with eps = 1e-8 (dsp.py:83), then mag * cos, mag * sin (dsp.py:112-116): float32, every
          // operation rounded on its own as torch's separate tensor ops are
          ssr_f32x4 R, I;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v1 = acc1[g][h][4 * j + e], v2 = acc2[g][h][4 * j + e];
            const float aa = ssr_tl_mul(v1, v1), bb = ssr_tl_mul(v2, v2);
            float ss = ssr_tl_add(aa, bb);
            ss = ss < 1e-8f ? 1e-8f : ss;
            const float mag = sqrtf(ss);                  // correctly rounded (hipcc's default; __fsqrt_rn is the 1-ulp native one)
            const float cs = v1 / mag, sn = v2 / mag;
            R[e] = ssr_tl_mul(mag, cs);
            I[e] = ssr_tl_mul(mag, sn);
          }
          float* sr = p.spec_re + (int64_t)col * p.ldt + rq;
          float* si = p.spec_im + (int64_t)col * p.ldt + rq;
          if (rq >= tl.v_lo && rq + 4 <= tl.v_hi) {
            *(ssr_f32x4*)sr = R;
            *(ssr_f32x4*)si = I;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (rq + e >= tl.v_lo && rq + e < tl.v_hi) { sr[e] = R[e]; si[e] = I[e]; }
          }
        }
      }
  }
}

// =====================================================================================================================================
// inverse
// =====================================================================================================================================
// Chunks of the channel axis for a cut c: three runs of channels - the direct ones [0, c), the mirrored ones up to the next
// multiple of SSR_TL_KBF [m0, m0 + l0), the rest of the mirrored ones [m0 + l0, n_fft) - each walked in steps of 16 channels; the
// last chunk of a run is padded (the padded slots read the zero row of the spectrum).  The first and the last run start on a
// multiple of SSR_TL_KBF, so within them a chain starts wherever the chunk's first channel is such a multiple; the middle run
// starts a chain with its first chunk unless channel m0 - 1 is a direct channel of the same block (c = n_bins only).
struct SsrTlRuns {
  int c, m0, l0, n_fft, total;
  bool joined;
};
__device__ __forceinline__ SsrTlRuns ssr_tl_runs(int c, int n_fft) {
  SsrTlRuns q;
  const int mmax = ssr_tl_mmax(c, n_fft);
  q.c = c; q.n_fft = n_fft;
  q.m0 = n_fft - mmax;
  const int to_edge = SSR_TL_KBF - q.m0 % SSR_TL_KBF;
  q.l0 = mmax < to_edge ? mmax : to_edge;
  const int rest = mmax - q.l0;
  q.total = (c + SSR_TL_BK - 1) / SSR_TL_BK + (q.l0 + SSR_TL_BK - 1) / SSR_TL_BK + (rest + SSR_TL_BK - 1) / SSR_TL_BK;
  q.joined = c > 0 && mmax > 0 && (c - 1) / SSR_TL_KBF == q.m0 / SSR_TL_KBF;
  return q;
}
// a cursor over the chunks (wave-uniform state: scalar registers)
struct SsrTlCursor {
  int ch0, end, run;
  __device__ __forceinline__ void init(const SsrTlRuns& q) { ch0 = 0; end = q.c; run = 0; }
  __device__ __forceinline__ int valid() const { return end - ch0 < SSR_TL_BK ? end - ch0 : SSR_TL_BK; }
  __device__ __forceinline__ bool starts_chain(const SsrTlRuns& q) const {
    return run == 1 ? (ch0 == q.m0 && !q.joined) : ch0 % SSR_TL_KBF == 0;
  }
  __device__ __forceinline__ void advance(const SsrTlRuns& q) {
    ch0 += SSR_TL_BK;
    if (ch0 >= end) {
      if (run == 0 || (run == 1 && q.l0 == 0)) { run = q.l0 > 0 ? 1 : 2; ch0 = q.m0; end = q.m0 + q.l0; }
      if (run == 1 && ch0 >= end) { run = 2; ch0 = q.m0 + q.l0; end = q.n_fft; }
    }
  }
};

template <int BM>
__device__ __forceinline__ void ssr_tl_inv_body(const SsrTlParams& p, float* lds) {
  constexpr int NW = BM / 16;                             // waves: (BM / 64) row groups x 4 column groups
  constexpr int A_FLOATS = SSR_TL_BK * BM, B_FLOATS = SSR_TL_BK * SSR_TL_BN;
  constexpr int STAGE_FLOATS = 2 * A_FLOATS + 2 * B_FLOATS;
  constexpr int KPI = 256 / BM;                           // channel rows one A transfer moves (64 lanes x 4 frames)
  constexpr int NB = 8 / NW;                              // B transfers per wave and table (2 channel rows each)
  constexpr int PER = 2 + 2 * NB;                         // transfers per wave and chunk
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;                // wave = rows [64 wr, +64) x columns [32 wc, +32)
  const int Mt = p.uniform_cut >= 0 ? p.m_tiles : p.m_tiles * p.n_items;
  const int Nt = (p.n_fft + SSR_TL_BN - 1) / SSR_TL_BN;
  int mt, ct;
  if (!ssr_tl_block_tile((int)blockIdx.x, Mt, Nt, mt, ct)) return;
  SsrTlTile tl;
  if (!ssr_tl_tile(p, mt, BM, tl)) return;
  const int c = tl.c, n0 = ct * SSR_TL_BN;
  const bool wave_on = n0 + 32 * wc < p.n_fft;            // (n_fft = 32 m: whole 32-column groups)
  const SsrTlRuns runs = ssr_tl_runs(c, p.n_fft);
  const int n_chunks = runs.total;

  // ---- LDS-DMA assignments -------------------------------------------------------------------------------------
  // A: wave w moves channel rows [KPI w, KPI w + KPI) of the chunk, re and im; lane -> row KPI w + lane / (BM / 4), frames
  // 4 (lane % (BM / 4)).  A chunk lies wholly in the direct run (bin = channel) or in a mirrored one (bin = n_fft - channel), so a
  // lane's source is a per-lane pointer plus a wave-uniform multiple of the row pitch.
  const int a_krow = KPI * wave + lane / (BM / 4);
  const int64_t a_col = tl.lo + 4 * (lane % (BM / 4));
  const float* a_plus = p.spec_re + a_col + (int64_t)a_krow * p.ldt;       // direct: + ch0 ldt
  const float* a_minus = p.spec_re + a_col - (int64_t)a_krow * p.ldt;      // mirrored: + (n_fft - ch0) ldt
  const uint64_t a_im = (uint64_t)(p.spec_im - p.spec_re);
  const float* a_zero = p.zero_row + a_col;
  const uint32_t ldt = (uint32_t)p.ldt;
  // B: transfer i of wave w moves channel rows 2 (w + NW i) + lane / 32 of the chunk; a lane covers 4 columns.  (Rows past the
  // chunk's channels multiply the zero row: the tables carry 16 rows of slack.)
  const float* b_re[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) b_re[i] = p.ire_t + (int64_t)(2 * (wave + NW * i) + (lane >> 5)) * p.n_fft + n0 + 4 * (lane & 31);
  const uint64_t b_im = (uint64_t)(p.iim_t - p.ire_t);
  const unsigned lds0 = ssr_tl_lds_off(lds);

  // transfer `piece` (0, 1: A re / im; 2 + 2 i, 3 + 2 i: B re / im of row pair i) of the chunk under cursor `cu` into `stage`
  auto issue_piece = [&](int piece, const SsrTlCursor& cu, int stage) {
    const unsigned st = lds0 + 4u * (unsigned)(stage * STAGE_FLOATS);
    if (piece < 2) {
      const uint64_t off = (uint64_t)(uint32_t)(cu.run ? p.n_fft - cu.ch0 : cu.ch0) * ldt + (piece ? a_im : 0);
      const float* s1 = (cu.run ? a_minus : a_plus) + off;
      if (cu.end - cu.ch0 < SSR_TL_BK) s1 = a_krow < cu.end - cu.ch0 ? s1 : a_zero;
      ssr_tl_glds16(s1, st + 4u * (unsigned)(piece * A_FLOATS + wave * 256));
    } else {
      const int i = (piece - 2) >> 1, im = (piece - 2) & 1, q = wave + NW * i;
      const uint64_t off = (uint64_t)(uint32_t)cu.ch0 * (uint32_t)p.n_fft + (im ? b_im : 0);
      ssr_tl_glds16(b_re[i] + off, st + 4u * (unsigned)(2 * A_FLOATS + im * B_FLOATS + q * 256));
    }
  };

  ssr_f32x16 acc1[2], acc2[2], tot1[2], tot2[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[g][r] = 0.0f; acc2[g][r] = 0.0f; tot1[g][r] = 0.0f; tot2[g][r] = 0.0f; }

  const int fi = lane & 31, fk = lane >> 5;
  // operands of one step (one k pair = four matrix instructions, 256 cycles of the matrix pipe): two register sets, the set of step
  // s + 1 is requested BEFORE the matrix instructions of step s are issued and first touched after them
  float fb1[2], fb2[2], fa10[2], fa11[2], fa20[2], fa21[2];
  auto fetch = [&](int stage, int s, int slot) {
    const float* st = lds + stage * STAGE_FLOATS;
    const float* as1 = st + 64 * wr + fi;
    const float* as2 = as1 + A_FLOATS;
    const float* bs1 = st + 2 * A_FLOATS + 32 * wc + fi;
    const float* bs2 = bs1 + B_FLOATS;
    const int k = 2 * s + fk;
    fb1[slot] = bs1[k * SSR_TL_BN]; fb2[slot] = bs2[k * SSR_TL_BN];
    fa10[slot] = as1[k * BM]; fa11[slot] = as1[k * BM + 32];
    fa20[slot] = as2[k * BM]; fa21[slot] = as2[k * BM + 32];
  };
  auto mfma4 = [&](int cur) {
    acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa10[cur], fb1[cur], acc1[0], 0, 0, 0);
    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa20[cur], fb2[cur], acc2[0], 0, 0, 0);
    acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa11[cur], fb1[cur], acc1[1], 0, 0, 0);
    acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa21[cur], fb2[cur], acc2[1], 0, 0, 0);
  };

  // ---- the pipeline: three stages, ONE barrier per chunk, in the MIDDLE of the chunk's eight steps -----------------------------------
  // barrier(c) publishes chunk c + 1 (requested half a chunk and more earlier: the wait before it does not stall) and frees the
  // stage chunk c - 1 was read from, which the second half of chunk c then fills with chunk c + 2, one or two transfers per step,
  // between the matrix instructions.  A wave crosses from chunk c to chunk c + 1 without any synchronisation (its first operands
  // are requested under the last step of chunk c), and after the barrier the operands of step 4 are already in registers: the
  // matrix pipe has work on both sides of every wait.
  SsrTlCursor cu;                                       // the chunk being requested (two ahead of the one being multiplied)
  cu.init(runs);
  bool start1 = false, start2 = false;                  // whether chunks c + 1, c + 2 start a chain
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (i < n_chunks) {
#pragma unroll
      for (int piece = 0; piece < PER; ++piece) issue_piece(piece, cu, i);
      (i ? start2 : start1) = cu.starts_chain(runs);
      cu.advance(runs);
    }
  // (chunk 0 starts the first chain by definition: start1 / start2 now describe chunks 0 and 1; shift so that they describe 1 and 2)
  start1 = start2;
  if (n_chunks >= 2) ssr_tl_wait_vm<PER>(); else ssr_tl_wait_vm<0>();
  __syncthreads();
  if (wave_on && n_chunks > 0) fetch(0, 0, 0);
  int stage = 0;
  auto chunk_body = [&](int chunk, bool chain_start, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;     // chunk + 2 exists: request it in the second half
    const int st1 = stage == 2 ? 0 : stage + 1, st2 = stage == 0 ? 2 : stage - 1;     // stages of chunks c + 1, c + 2
    if (chain_start) {                                  // a chain ends: total += chain, the next one starts from 0
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
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fetch(stage, s + 1, (s + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma4(s & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (chunk + 1 < n_chunks) {                         // (wave-uniform and the same for every wave of the workgroup)
      ssr_tl_wait_vm<0>();
      __syncthreads();
    }
#pragma unroll
    for (int s = 4; s < 8; ++s) {
      if (MORE) {
        constexpr int P0 = PER == 4 ? 1 : 2;            // transfers in steps 4 and 5 each; the remaining two go one per step
        if (s < 6) {
#pragma unroll
          for (int j = 0; j < P0; ++j) issue_piece((s - 4) * P0 + j, cu, st2);
        } else {
          issue_piece(2 * P0 + (s - 6), cu, st2);
        }
      }
      if (s < 7) fetch(stage, s + 1, (s + 1) & 1);
      else if (chunk + 1 < n_chunks) fetch(st1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma4(s & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    stage = st1;
  };
  if (wave_on) {
    int chunk = 0;
    bool start0 = false;                                // (chunk 0: nothing to close)
    for (; chunk + 2 < n_chunks; ++chunk) {
      const bool s2 = cu.starts_chain(runs);
      chunk_body(chunk, start0, std::true_type{});
      cu.advance(runs);
      start0 = start1; start1 = s2;
    }
    for (; chunk < n_chunks; ++chunk) {
      chunk_body(chunk, start0, std::false_type{});
      start0 = start1;
    }
  } else {
    // a wave whose 32 columns lie past n_fft (n_fft not a multiple of 128) only moves its share of the operands
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
      if (chunk + 1 < n_chunks) { ssr_tl_wait_vm<0>(); __syncthreads(); }
      if (chunk + 2 < n_chunks) {
        const int st2 = stage == 0 ? 2 : stage - 1;
#pragma unroll
        for (int piece = 0; piece < PER; ++piece) issue_piece(piece, cu, st2);
        cu.advance(runs);
      }
      stage = stage == 2 ? 0 : stage + 1;
    }
    return;
  }

  // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -------------------------------
  const int col = n0 + 32 * wc + fi;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = tl.lo + 64 * wr + 32 * g + (r & 3) + 8 * (r >> 2) + 4 * fk;
      if (row < tl.v_lo || row >= tl.v_hi) continue;
      const float v1 = ssr_tl_add(tot1[g][r], acc1[g][r]), v2 = ssr_tl_add(tot2[g][r], acc2[g][r]);
      p.frames[row * p.n_fft + col] = ssr_tl_add(v1, -v2);
    }
}

// Padding as torchlibrosa's STFT.forward does it: F.pad(x, (n_fft/2, n_fft/2), mode = "reflect" | "constant") when center, none otherwise.
// Also writes rowbase[row] = offset of the row's frame in xpad (uniform-cut launches tile the batch's rows across items).
struct SsrTlPadParams {
  const float* in; const int64_t* in_off; const int32_t* len; const int64_t* frame_off;
  int n_fft, hop; float* xpad; int64_t pad_stride; int pad, pad_reflect; int64_t* rowbase;
};
__device__ __forceinline__ void ssr_tl_pad_body(const SsrTlPadParams& p, int item, int64_t i) {
  const int len = p.len[item];
  if (!ssr_tl_item_ok(len, p.n_fft, p.pad, p.pad_reflect) || i >= (int64_t)len + 2 * p.pad) return;
  const int64_t base = ssr_tl_pad_off(p.pad_stride, p.frame_off[item], p.hop, item, p.n_fft);
  if (p.rowbase && i < ssr_tl_frames_of(len, p.n_fft, p.hop, p.pad)) p.rowbase[p.frame_off[item] + i] = base + i * p.hop;
  int64_t s = i - p.pad;
  float v;
  if (s >= 0 && s < len) v = p.in[p.in_off[item] + s];
  else if (!p.pad_reflect) v = 0.0f;                               // F.pad(mode="constant")
  else {
    if (s < 0) s = -s;
    if (s >= len) s = 2 * ((int64_t)len - 1) - s;
    v = p.in[p.in_off[item] + s];
  }
  p.xpad[base + i] = v;
}

// ISTFT given (re, im) [rows][n_bins]: the transposed spectrum [n_bins][ldt] the inverse product reads (32 x 32 tiles through LDS).
struct SsrTlPackParams {
  const float* re; const float* im; int64_t total_rows; int n_bins; float* spec_re; float* spec_im; int64_t ldt;
};
__device__ __forceinline__ void ssr_tl_pack_body(const SsrTlPackParams& p, float* tile /* [2][32][33] */) {
  const int tx = (int)threadIdx.x & 31, ty = (int)threadIdx.x >> 5;          // 256 threads: 32 x 8
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  const int b0 = (int)blockIdx.y * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t row = r0 + ty + 8 * j;
    const int bin = b0 + tx;
    const bool ok = row < p.total_rows && bin < p.n_bins;
    tile[(ty + 8 * j) * 33 + tx] = ok ? p.re[row * p.n_bins + bin] : 0.0f;
    tile[32 * 33 + (ty + 8 * j) * 33 + tx] = ok ? p.im[row * p.n_bins + bin] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int bin = b0 + ty + 8 * j;
    const int64_t row = r0 + tx;
    if (bin < p.n_bins && row < p.total_rows) {
      p.spec_re[(int64_t)bin * p.ldt + row] = tile[tx * 33 + ty + 8 * j];
      p.spec_im[(int64_t)bin * p.ldt + row] = tile[32 * 33 + tx * 33 + ty + 8 * j];
    }
  }
}

// F.fold + / clamp(folded window^2, 1e-11) + trim (ISTFT._overlap_add_divide_window_sum, _trim_edges).
struct SsrTlFoldParams {
  const float* frames; const int64_t* frame_off; const int32_t* len; const int64_t* out_off; int n_fft, hop;
  const float* w2; float* out; int pad, pad_reflect;
};
// FOUR output samples per thread, 256 apart (s0, s0 + 256, ...): their frame rows are independent chains of loads, so a thread keeps
// up to twenty requests in flight instead of five; each sample's sums run in the same order as before (frames in descending order).
__device__ __forceinline__ void ssr_tl_fold_body(const SsrTlFoldParams& p, int item, int s0) {
  const int len = p.len[item];
  if (s0 >= len) return;
  float* out = p.out + p.out_off[item];
  const bool ok = ssr_tl_item_ok(len, p.n_fft, p.pad, p.pad_reflect);
  const int T = ok ? ssr_tl_frames_of(len, p.n_fft, p.hop, p.pad) : 1;
  const float* fr = p.frames + p.frame_off[item] * (int64_t)p.n_fft;
  const int end = (T - 1) * p.hop + p.n_fft;          // samples of the overlap-added signal
  float y[4], ws[4];
  int q[4], t[4];
  bool live[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int s = s0 + 256 * j;
    q[j] = s + p.pad;                                  // ISTFT._trim_edges: start = n_fft / 2 if center else 0
    live[j] = ok && s < len && q[j] < end;             // (past the overlap-added signal the slice ends: zero-filled)
    int tt = q[j] / p.hop;
    t[j] = tt < T - 1 ? tt : T - 1;
    y[j] = 0.0f; ws[j] = 0.0f;
  }
  // a sample lies in at most ceil(n_fft / hop) frames; the chains advance together, a finished one idles
  for (bool any = true; any;) {
    any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool go = live[j] && t[j] >= 0 && q[j] - t[j] * p.hop < p.n_fft;
      if (go) {
        y[j] = ssr_tl_add(y[j], fr[(int64_t)t[j] * p.n_fft + (q[j] - t[j] * p.hop)]);
        ws[j] = ssr_tl_add(ws[j], p.w2[q[j] - t[j] * p.hop]);
        --t[j];
        any = true;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int s = s0 + 256 * j;
    if (s >= len) continue;
    const float w = ws[j] < 1e-11f ? 1e-11f : ws[j];
    out[s] = live[j] ? y[j] / w : 0.0f;
  }
}
