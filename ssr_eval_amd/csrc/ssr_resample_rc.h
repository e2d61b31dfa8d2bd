// K7, round 4: the polyphase resampler with one lane per RESIDUE CLASS of outputs.  Same arithmetic as ssr_resample.h - every
// output accumulated in ascending input index with a separately rounded float32 multiply and add, SciPy's _upfirdn_apply loop
// (scipy.signal.resample_poly = librosa.resample(res_type="polyphase"), ssr_eval/eval.py:144-150) - hence the same bits.
//
// What changed.  out[m] = sum_k x[q(m) - (HPP-1) + k] * h[ph(m) + (HPP-1-k) up] with (m + n_pre_remove) down = q up + ph: the phase
// depends on m mod up only, and q(r + j up) = q(r) + j down.  So lane r of a workgroup owns the outputs r, r + up, r + 2 up, ...:
//   * its HPP (= 21 for every up-sampling plan of resample_poly) taps are loop-invariant REGISTERS - no tap table in LDS, no tap
//     reads, no tap index arithmetic in the loop (the round-2/3 kernel read one tap per 8 multiply-adds from a 37 KB table that
//     capped the CU at three workgroups, with 3-way bank conflicts for 441/160);
//   * the input of output j starts `down` samples after that of output j - 1: all LDS addresses of a block of JB outputs are one
//     per-lane base plus compile-time-known strides;
//   * the window is staged TWICE (the second copy shifted by one sample, on the other half of the banks), so that every lane reads
//     its 21 consecutive samples as 11 ALIGNED 8-byte loads whatever the parity of its first index: ds_read_b64 moves twice the
//     bytes per LDS cycle of ds_read_b32 - the old kernel's 1.125 four-byte reads per multiply-add were 41 % of the CU's LDS
//     rate at three waves per SIMD; this one issues 0.52 eight-byte reads per multiply-add, the same cycles as the multiply-adds
//     themselves take on the VALU;
//   * consecutive lanes = consecutive outputs: stores are full 256-byte runs; consecutive lanes read input indices 0.36
//     (441/160) or 0.92 (160/147) apart: broadcasts, no bank conflict;
//   * a workgroup is ceil(up / 64) waves streaming over one item's outputs in blocks of JB steps; the next block's window goes
//     global -> LDS by LDS-DMA while the current one is computed (no staging registers, no ds_write; two LDS stages of ~12 KB:
//     four workgroups per CU), one barrier per block.
// Device code only (the host emulation keeps exercising ssr_resample.h, which remains the kernel of every plan this one does not
// take: taps per phase != 21, up < 33 or > 1024, float64 signals).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssr_resample.h"

#define SSR_RC_JB 8            /* outputs per lane and block */

struct SsrResampleRcParams {
  const float* in;
  const int64_t* in_off;
  const int32_t* in_len;
  const int64_t* out_off;
  const int32_t* out_len;
  int up, down, n_taps, n_pre_remove;
  const float* taps;
  int blocks_per_chunk, n_chunks;     // a workgroup walks `blocks_per_chunk` blocks of JB steps of ONE item
  int seg_floats;                     // floats per window copy (even, and = 32 mod 64: the two copies sit on opposite bank halves)
  float* out;
};

// window of a block of JB steps starting at step j0: inputs [lo, lo + len)
__host__ __device__ inline int ssr_rc_seg_len(int up, int down, int hpp) {
  return (SSR_RC_JB - 1) * down + (int)(((int64_t)(up - 1) * down + (up - 1)) / up) + hpp + 6;      // (+ the last chain's over-read)
}

template <int HPP>
__device__ __forceinline__ void ssr_resample_rc_body(const SsrResampleRcParams& p, char* smem) {
  constexpr int JB = SSR_RC_JB, NP = (HPP + 1) / 2;
  const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
  const int item = (int)blockIdx.x / p.n_chunks, chunk = (int)blockIdx.x % p.n_chunks;
  const int up = p.up, down = p.down;
  const int n_in = p.in_len[item], n_out = p.out_len[item];
  const float* x = p.in + p.in_off[item];
  float* y = p.out + p.out_off[item];
  const int steps = (n_out + up - 1) / up;                              // outputs r + j up, j < steps
  const int blk0 = chunk * p.blocks_per_chunk;
  if (blk0 * JB >= steps) return;
  int blk1 = blk0 + p.blocks_per_chunk;
  if (blk1 * JB > steps) blk1 = (steps + JB - 1) / JB;
  typedef __attribute__((address_space(3))) float lds_float;               // 32-bit LDS addresses (a generic pointer costs a register pair)
  typedef float ssr_v2f __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) ssr_v2f lds_float2;
  lds_float* lds = (lds_float*)smem;
  const int SEG = p.seg_floats;                                         // stage s: copies at lds + s * 2 SEG and + s * 2 SEG + SEG

  // lane constants: phase, first input index, taps (k ascending = input ascending = tap index descending)
  const bool active = tid < up;
  const int r = active ? tid : 0;
  const unsigned t0 = (unsigned)(r + p.n_pre_remove) * (unsigned)down;
  const int q0 = (int)(t0 / (unsigned)up), ph = (int)(t0 - (unsigned)q0 * (unsigned)up);
  float tap[HPP];
  {
    const SsrView<float> vt(p.taps, p.n_taps);                            // (an index past the table loads 0: the zero padding of the last phase rows)
#pragma unroll
    for (int k = 0; k < HPP; ++k) tap[k] = vt.at_or_zero((unsigned)(ph + (HPP - 1 - k) * up));
  }
  const SsrRwView<float> vy(y, n_out);
  const int qmin0 = (int)(((unsigned)p.n_pre_remove * (unsigned)down) / (unsigned)up);    // q of residue 0 at step 0
  const int seg_len = ssr_rc_seg_len(up, down, HPP);

  // Staging.  A block whose window lies inside the signal goes global -> LDS directly (LDS-DMA, 4 bytes per lane: a wave deposits
  // 64 consecutive samples per instruction, once for copy A[i] = x[lo + i] and once, one float lower, for copy B[i] = x[lo + i + 1]):
  // no staging registers, no ds_write, the transfer runs under the current block's multiply-adds.  The first / last blocks of an
  // item, whose windows reach outside [0, n_in), take ordinary loads with the zero extension upfirdn applies.
  auto stage = [&](int blk, int s) {
    const int lo = qmin0 + blk * JB * down - (HPP - 1);
    lds_float* a = lds + s * 2 * SEG;
    lds_float* b = a + SEG;
    if (lo >= 0 && lo + seg_len + 1 + 64 <= n_in) {                       // block-uniform
      const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
      for (int i0 = wave * 64; i0 < seg_len + 1; i0 += nw * 64) {         // wave-uniform trip count
        const float* src = x + lo + i0 + lane;
#ifndef SSR_RC_DEBUG_SLOWA
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(a + i0), 4, 0, 0);
#else
        a[i0 + lane] = src[0];
#endif
#ifndef SSR_RC_DEBUG_SLOWB
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 1),
                                         (__attribute__((address_space(3))) void*)(b + i0), 4, 0, 0);
#else
        b[i0 + lane] = src[1];
#endif
      }
    } else {
      for (int i = tid; i < seg_len + 1; i += nt) {
        const int g = lo + i;
        const float v = (g >= 0 && g < n_in) ? x[g] : 0.0f;
        if (i < seg_len) a[i] = v;
        if (i >= 1) b[i - 1] = v;
      }
    }
  };

  stage(blk0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int blk = blk0; blk < blk1; ++blk) {
    const int s = (blk - blk0) & 1;
    if (blk + 1 < blk1) stage(blk + 1, s ^ 1);
    {                                                                     // (lanes past `up` compute residue 0 again; their stores are dropped)
      const lds_float* a = lds + s * 2 * SEG;
      const int b0 = q0 - qmin0;                                          // index (in copy A) of the first input of the block's first output
      // JB independent accumulation chains (k ascending within each: SciPy's order), one aligned sample PAIR per chain and step.
      // The pairs are single ds_read_b64 on purpose (256 B/clk, banks mod 64: the two window copies never collide); the compiler
      // would merge neighbouring pairs of a chain into ds_read2_b64 - half the rate, banks mod 32: 45 % of the LDS cycles were bank
      // conflicts - so a chain's address goes through an empty asm statement every step (distinct, opaque bases cannot be merged).
      // The order is pinned through the data: step i + 1's loads are issued before step i's multiply-adds.
      float acc[JB];
      unsigned ad[JB];                                                    // LDS byte address of the chain's first pair
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const int bj = b0 + j * down;
        // even index: copy A; odd index: copy B, one slot lower (B[bj - 1] = A[bj])
        ad[j] = (unsigned)(uintptr_t)(a + ((bj & 1) ? SEG + bj - 1 : bj));
        acc[j] = 0.0f;
      }
      float cur[JB][2], nxt[JB][2];
      auto fetch = [&](float (*dst)[2], int i) {
#pragma unroll
        for (int j = 0; j < JB; ++j) {
          asm volatile("" : "+v"(ad[j]));
          const ssr_v2f v = *(const lds_float2*)(uintptr_t)(ad[j] + 8u * (unsigned)i);
          dst[j][0] = v.x; dst[j][1] = v.y;
        }
      };
      fetch(cur, 0);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if (i + 1 < NP) fetch(nxt, i + 1);
#pragma unroll
        for (int j = 0; j < JB; ++j) { ssr_touch(cur[j][0]); ssr_touch(cur[j][1]); }     // step i's values are consumed from here on ...
#pragma unroll
        for (int j = 0; j < JB; ++j) {
          acc[j] = ssr_fadd_rn(acc[j], ssr_fmul_rn(cur[j][0], tap[2 * i]));
          if (2 * i + 1 < HPP) acc[j] = ssr_fadd_rn(acc[j], ssr_fmul_rn(cur[j][1], tap[2 * i + 1]));
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) ssr_touch(acc[j]);                   // ... and its multiply-adds are complete here
        if (i + 1 < NP) {
#pragma unroll
          for (int j = 0; j < JB; ++j) { cur[j][0] = nxt[j][0]; cur[j][1] = nxt[j][1]; }
        }
      }
      // JB stores per wave, ALWAYS issued (a buffer view drops what lies past the item's end or belongs to a lane past `up`): the
      // wait below counts on exactly JB vector-memory instructions being younger than the LDS-DMA of the next block
      const int m0 = r + blk * JB * up;
#pragma unroll
      for (int j = 0; j < JB; ++j) vy.st_raw(active ? 4 * (m0 + j * up) : -1, acc[j]);
    }
    // the next block's LDS-DMA (issued before the multiply-adds) has landed - the stores just issued may still be in flight (vmcnt
    // retires in order) - and this wave's DS operations are complete, before anyone passes the barrier
    static_assert(JB == 8, "the immediate below");
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  }
}
