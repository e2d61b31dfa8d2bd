// Kernel body K7, matrix-core variant (round 3): the polyphase FIR resampler of ssr_resample.h cast as a dense (outputs x input
// window) by (input window x utterances) product on v_mfma_f32_32x32x2_f32 - the "MFMA only if the FIR is cast as a dense
// frame x tap GEMM" of the task statement.
//
//   out[m] = sum_i x[q - i] h[ph + i up],   t = (m + n_pre_remove) down,  q = t / up,  ph = t mod up           (ssr_resample.h)
//
// A block of 32 consecutive outputs m = mb + i of ONE utterance reads the input window s = qb + k, k < K
// (qb = q(mb) - (hpp - 1), K = floor((up - 1 + 31 down) / up) + hpp: 34 for 441 / 160, 50 for 160 / 147), each output through its
// own hpp of the K taps:
//   y[i] = sum_k A[i][k] x[qb + k],     A[i][k] = h[ph_i + (dq_i + hpp - 1 - k) up]  (0 outside the output's hpp taps),
// and A depends on the block only through phi = ((mb + n_pre_remove) down) mod up - the SAME for block b of EVERY utterance.  So
// the 32 x K matrix A of a block multiplies a K x 32 matrix of 32 utterances' windows: D[i][j] = sum_k A[i][k] X_j[qb + k], K / 2
// MFMAs per 32 x 32 tile of outputs, 60 % of A non-zero for 441 / 160.  A is not stored: a lane builds its A[i][k] operand from the
// tap table in LDS (one gather per MFMA), B comes from the utterances' windows staged in LDS (row stride odd: conflict-free), and
// the tile is transposed through a wave-private 32 x 33 LDS array so that every store instruction writes two full 128-byte
// runs of two utterances.
//
// Arithmetic: the matrix core evaluates D = fma(a_k, b_k, D) in ascending k - ascending input index, SciPy's order - with ONE
// rounding per tap where SciPy's upfirdn rounds the product and the sum separately.  The result is therefore NOT bit-identical to
// scipy.signal.resample_poly (ssr_resample.h is, and stays the default); it is the float32 fused-multiply-add evaluation of the
// same sum, within ~1 ulp per tap of it (tests: <= 4e-7 x sum |h| max |x|).  Metrics that take the logarithm of the resampler's
// own stop-band leakage move with those ulps: LSD by up to 1.2e-5 relative over cfg-5's 12,500 white-noise utterances (7e-7 on
// speech), log-SISpec by 2.5e-5 - which is why this kernel is an option and not the default.
// The multiplications by the zeros of A add exact zeros.
//
// Workgroup = four waves, persistent: stages the tap table once, then walks passes of (group of 32 utterances) x (NB blocks of
// 32 outputs); the next pass's 32 windows are requested into registers before the current pass's tiles and written to LDS after
// them.  Wave w takes blocks w, w + 4, ... of the pass.
#pragma once
#include "ssr_resample.h"

constexpr int SSR_RMF_NT = 256, SSR_RMF_WAVES = 4, SSR_RMF_T = 32;
constexpr int SSR_RMF_TB = SSR_RMF_T * (SSR_RMF_T + 1);
constexpr int SSR_RMF_MAXCH = 7;                 // 64-float chunks of a window a lane prefetches per utterance: W <= 448

struct SsrResampleMfmaGeom {
  int hpp, K, NB, W, WS, ok;
  int pad, hl_len;          // tap table stored with one spare word per 32 (see ssr_rmf_slot); its length in LDS
  size_t lds_bytes;
};
// The lanes of an A gather read taps `down` apart (tap index = phi + i down + const for output row i): down = 160 = 5 x 32 put
// all 32 rows on ONE bank - 64 LDS cycles per gather, the CU's LDS pipe saturated at a tenth of the matrix rate (measured: 5.2 ms
// where the bit-exact kernel takes 6.5).  An even `down` gets the table stored with a spare word every 32 (stride 165: odd).
SSR_HD int ssr_rmf_slot(int t, int pad) { return pad ? t + (t >> 5) : t; }
// host and device agree on the geometry through this function
SSR_HD SsrResampleMfmaGeom ssr_resample_mfma_geom(int up, int down, int n_taps) {
  SsrResampleMfmaGeom g;
  g.hpp = (n_taps + up - 1) / up;
  g.K = (int)(((int64_t)(up - 1) + 31 * (int64_t)down) / up) + g.hpp;
  g.K += g.K & 1;
  g.ok = 0; g.NB = 0; g.W = 0; g.WS = 0; g.lds_bytes = 0;
  g.pad = (down % 2 == 0) ? 1 : 0;
  g.hl_len = ssr_rmf_slot(g.hpp * up, g.pad) + 1;
  if ((int64_t)g.hpp * up > 24 * 1024 || g.K + 2 > 64 * SSR_RMF_MAXCH) return g;     // tap table beyond 96 KB / window beyond the registers
  // blocks per pass: a multiple of four, as many as keep the window inside the prefetch registers and two workgroups on a CU
  for (int nb = 32; nb >= 4; nb -= 4) {
    const int w = (int)(((int64_t)32 * (nb - 1) * down + (up - 1)) / up) + g.K + 1;
    const int ws = w | 1;
    const size_t lds = sizeof(float) * ((size_t)g.hl_len + (size_t)SSR_RMF_T * ws + SSR_RMF_WAVES * SSR_RMF_TB) + 32 * 16;
    if (w <= 64 * SSR_RMF_MAXCH && (lds <= 78 * 1024 || nb == 4)) {
      if (lds > 156 * 1024) return g;
      g.NB = nb; g.W = w; g.WS = ws; g.lds_bytes = lds; g.ok = 1;
      return g;
    }
  }
  return g;
}

struct SsrResampleMfmaParams {
  SsrResampleParams rp;     // signals, lengths, up / down, taps (groups / taps_in_lds unused)
  int n_items_total;
  int max_out_len;
  int n_groups;             // ceil(n_items / 32)
  int passes_per_group;     // ceil(max_out_len / (32 NB))
};

struct SsrResampleMfmaLds {
  float* hl; float* xs; float* tb; int64_t* oo; int32_t* ol;
  SSR_MEMBER SsrResampleMfmaLds(char* base, const SsrResampleMfmaGeom& g, int up) {
    oo = reinterpret_cast<int64_t*>(base);                    // [32] output offsets of the pass's utterances
    ol = reinterpret_cast<int32_t*>(oo + 32);                 // [32] output lengths (0: no such utterance)
    hl = reinterpret_cast<float*>(base + 32 * 16);            // [hpp up] taps, zero beyond n_taps
    xs = hl + (size_t)g.hl_len;                               // [32][WS] windows
    tb = xs + (size_t)SSR_RMF_T * g.WS;                       // [4 waves][32][33] tile transposition
  }
};

template <int MAXCH> struct SsrResampleMfmaRegs { float nx[8 * MAXCH]; };

// first output, first-block quotient / phase and window origin of a pass
struct SsrResampleMfmaPass { int group, pass; int64_t m_lo, q_lo; unsigned phi0; };
SSR_DEV SsrResampleMfmaPass ssr_resample_mfma_pass(const SsrResampleMfmaParams& p, const SsrResampleMfmaGeom& g, int idx) {
  SsrResampleMfmaPass s;
  s.group = idx / p.passes_per_group;
  s.pass = idx - s.group * p.passes_per_group;
  s.m_lo = (int64_t)s.pass * 32 * g.NB;
  const int64_t T0 = (s.m_lo + p.rp.n_pre_remove) * p.rp.down;
  const int64_t Q0 = T0 / p.rp.up;
  s.phi0 = (unsigned)(T0 - Q0 * p.rp.up);
  s.q_lo = Q0 - (g.hpp - 1);
  return s;
}

// A[row][kk] of the block with phase phib (row < 32): the tap output mb + row applies to window sample kk
SSR_DEV float ssr_resample_mfma_a(const float* hl, int pad, int up, int down, int hpp, unsigned phib, int row, int kk) {
  const unsigned a = phib + (unsigned)row * (unsigned)down;
  const unsigned dq = a / (unsigned)up;
  const int ph = (int)(a - dq * (unsigned)up);
  const int ii = (int)dq + hpp - 1 - kk;
  return ((unsigned)ii < (unsigned)hpp) ? hl[ssr_rmf_slot(ph + ii * up, pad)] : 0.0f;
}

// MAXCH: 64-sample chunks of a window a lane can prefetch per utterance (5: windows up to 320 samples - 441 / 160 and 160 / 147;
// SSR_RMF_MAXCH = 7 for the down-sampling plans: sixteen more registers and sixteen more guarded requests per pass)
template <int MAXCH, typename BLK>
SSR_BODY void ssr_resample_mfma_body(const SsrResampleMfmaParams& p, BLK& blk, int first, int stride, int total, char* lds_base) {
  constexpr int NT = SSR_RMF_NT, T = SSR_RMF_T;
  static_assert(MAXCH <= SSR_RMF_MAXCH, "geometry admits windows of 64 SSR_RMF_MAXCH samples");
  const int up = p.rp.up, down = p.rp.down;
  const SsrResampleMfmaGeom g = ssr_resample_mfma_geom(up, down, p.rp.n_taps);
  const int hpp = g.hpp, K = g.K, NB = g.NB, W = g.W, WS = g.WS;
  const int n_ch = (W + 63) / 64;
  SsrResampleMfmaLds L(lds_base, g, up);
  if (first >= total) return;

  // window of pass s -> registers: wave w serves utterances 8 w .. 8 w + 7 of the group, 64 consecutive samples per request
#define SSR_RMF_FETCH(s_)                                                                                     \
  {                                                                                                           \
    const int wv = ssr_wave_of(tid), lane = tid & 63;                                                         \
    SSR_UNROLL for (int jj = 0; jj < 8; ++jj) {                                                               \
      const int item = (s_).group * T + 8 * wv + jj;                                                          \
      const bool have = item < p.n_items_total;                                                               \
      const int ic = have ? item : 0;                                                                         \
      const SsrView<float> vx(p.rp.in + p.rp.in_off[ic], have ? p.rp.in_len[ic] : 0);                         \
      SSR_UNROLL for (int c = 0; c < MAXCH; ++c)                                                              \
        if (c < n_ch) R.nx[jj * MAXCH + c] = vx.at_or_zero((unsigned)((s_).q_lo + lane + 64 * c));            \
    }                                                                                                         \
  }
#define SSR_RMF_PUT()                                                                                         \
  {                                                                                                           \
    const int wv = ssr_wave_of(tid), lane = tid & 63;                                                         \
    SSR_UNROLL for (int jj = 0; jj < 8; ++jj)                                                                 \
      SSR_UNROLL for (int c = 0; c < MAXCH; ++c)                                                              \
        if (c < n_ch && lane + 64 * c < W) L.xs[(8 * wv + jj) * WS + lane + 64 * c] = R.nx[jj * MAXCH + c];   \
  }
#define SSR_RMF_DESC(s_)                                                                                      \
  if (tid < T) {                                                                                              \
    const int item = (s_).group * T + tid;                                                                    \
    const bool have = item < p.n_items_total;                                                                 \
    L.oo[tid] = have ? p.rp.out_off[item] : 0;                                                                \
    L.ol[tid] = have ? p.rp.out_len[item] : 0;                                                                \
  }

  SSR_REGS(SsrResampleMfmaRegs<MAXCH>, regs, blk);
  SsrResampleMfmaPass cur = ssr_resample_mfma_pass(p, g, first);
  SSR_PHASE(blk, regs, {
    for (int i = tid; i < hpp * up; i += NT) L.hl[ssr_rmf_slot(i, g.pad)] = (i < p.rp.n_taps) ? p.rp.taps[i] : 0.0f;
    SSR_RMF_FETCH(cur);
  });
  SSR_PHASE(blk, regs, {
    SSR_RMF_PUT();
    SSR_RMF_DESC(cur);
  });
  for (int idx = first; idx < total; idx += stride) {
    const bool more = idx + stride < total;
    const SsrResampleMfmaPass nxt = more ? ssr_resample_mfma_pass(p, g, idx + stride) : cur;
    SSR_WPHASE(blk, regs, { if (more) SSR_RMF_FETCH(nxt); });      // (registers only: no barrier)
    // ---- the pass's tiles: wave w, round r -> block 4 r + w
    for (int r4 = 0; r4 < NB; r4 += SSR_RMF_WAVES) {
      SSR_WPHASE(blk, regs, {
        const int wv = ssr_wave_of(tid), lane = tid & 63, bl = r4 + wv;
        const int64_t mb = cur.m_lo + 32 * bl;
        if (mb < p.max_out_len) {
          const unsigned ab = cur.phi0 + (unsigned)bl * 32u * (unsigned)down;       // < up + 32 NB down: 32 bits
          const unsigned Qb = ab / (unsigned)up, phib = ab - Qb * (unsigned)up;
          float* tb = L.tb + wv * SSR_RMF_TB;
#ifndef SSR_HOST_EMU
          typedef float f16v __attribute__((ext_vector_type(16)));
          f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          const int i = lane & 31, half = lane >> 5;
          const unsigned a = phib + (unsigned)i * (unsigned)down;
          const unsigned dq = a / (unsigned)up;
          const int ph = (int)(a - dq * (unsigned)up);
          const float* xrow = L.xs + i * WS + (int)Qb + half;
          int ii = (int)dq + hpp - 1 - half;                       // tap row of window sample kk = 2 kap + half
          // (the gather is unconditional on a clamped row and the zero is a select: no branch in the loop, so the operands of
          // the following MFMAs are requested while the current one runs)
          const int pad = g.pad;
          int kap = 0;
          for (; kap + 4 <= K / 2; kap += 4) {                     // four MFMAs' operands in flight
            float a4[4]; float b4[4];
            SSR_UNROLL for (int u = 0; u < 4; ++u) {
              const bool in_taps = (unsigned)(ii - 2 * u) < (unsigned)hpp;
              const float tv = L.hl[ssr_rmf_slot(ph + (in_taps ? ii - 2 * u : 0) * up, pad)];
              a4[u] = in_taps ? tv : 0.0f;
              b4[u] = xrow[2 * (kap + u)];
            }
            SSR_UNROLL for (int u = 0; u < 4; ++u) {
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], b4[u], acc, 0, 0, 0);
            }
            ii -= 8;
          }
          for (; kap < K / 2; ++kap) {
            const bool in_taps = (unsigned)ii < (unsigned)hpp;
            const float tv = L.hl[ssr_rmf_slot(ph + (in_taps ? ii : 0) * up, pad)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(in_taps ? tv : 0.0f, xrow[2 * kap], acc, 0, 0, 0);
            ii -= 2;
          }
          SSR_UNROLL for (int rg = 0; rg < 16; ++rg) tb[i * (T + 1) + (rg & 3) + 8 * (rg >> 2) + 4 * half] = acc[rg];
#else
          // the matrix core's arithmetic restated per element: D[row][col] = fma(A[row][k], B[k][col], D) in ascending k
          const int col = lane & 31, half = lane >> 5;
          for (int rg = 0; rg < 16; ++rg) {
            const int row = (rg & 3) + 8 * (rg >> 2) + 4 * half;
            float d = 0.0f;
            for (int kk = 0; kk < K; ++kk)
              d = fmaf(ssr_resample_mfma_a(L.hl, g.pad, up, down, hpp, phib, row, kk), L.xs[col * WS + (int)Qb + kk], d);
            tb[col * (T + 1) + row] = d;
          }
#endif
        }
      });
      // the tile leaves through the wave's 32 x 33 array: every store instruction writes two full 128-byte runs (two
      // utterances).  (Four 16-byte stores per lane straight from the accumulator layout - 32 bytes of each of 32 utterances per
      // instruction, no LDS - were measured slower: 0.6 ms of stores per 4096 utterances against 0.23.)
      SSR_WPHASE(blk, regs, {
        const int wv = ssr_wave_of(tid), lane = tid & 63, bl = r4 + wv;
        const int64_t mb = cur.m_lo + 32 * bl;
        if (mb < p.max_out_len) {
          const float* tb = L.tb + wv * SSR_RMF_TB;
          const int64_t m = mb + (lane & 31);
          SSR_UNROLL for (int pr = 0; pr < 16; ++pr) {
            const int jj = 2 * pr + (lane >> 5);
            if (m < L.ol[jj]) p.rp.out[L.oo[jj] + m] = tb[jj * (T + 1) + (lane & 31)];
          }
        }
      });
    }
    SSR_PHASE(blk, regs, {});                          // every wave is done with the windows and the descriptors
    SSR_PHASE(blk, regs, {
      if (more) {
        SSR_RMF_PUT();
        SSR_RMF_DESC(nxt);
      }
    });
    cur = nxt;
  }
#undef SSR_RMF_FETCH
#undef SSR_RMF_PUT
#undef SSR_RMF_DESC
}
