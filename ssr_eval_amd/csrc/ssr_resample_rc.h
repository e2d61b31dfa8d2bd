// K7, round 4: the polyphase resampler with one lane per RESIDUE CLASS of outputs.  Same arithmetic as ssr_resample.h - every
// output accumulated in ascending input index with a separately rounded float32 multiply and add, SciPy's _upfirdn_apply loop
// (scipy.signal.resample_poly = librosa.resample(res_type="polyphase"), ssr_eval/eval.py:144-150) - hence the same bits.
//
// What changed.  out[m] = sum_k x[q(m) - (HPP-1) + k] * h[ph(m) + (HPP-1-k) up] with (m + n_pre_remove) down = q up + ph: the phase
// depends on m mod up only, and q(r + j up) = q(r) + j down.  So lane r of a workgroup owns the outputs r, r + up, r + 2 up, ...:
//   * its HPP (= 21 for every up-sampling plan of resample_poly) taps are loop-invariant REGISTERS - no tap table in LDS, no tap
//     reads, no tap index arithmetic in the loop (the round-2/3 kernel read one tap per 8 multiply-adds from a 37 KB table that
//     capped the CU at three workgroups, with 3-way bank conflicts for 441/160);
//   * two successive outputs of a lane read inputs exactly `down` samples apart, and the window sits in LDS as PAIRS
//     L[i] = (x[lo + i], x[lo + i + down]): ONE aligned ds_read_b64 at pair i hands the lane sample k of output j AND sample k of
//     output j + 1 in a register pair - the operand of a packed multiply by (tap k, tap k) and a packed add into
//     (acc j, acc j + 1): v_pk_mul_f32 + v_pk_add_f32, each half rounded on its own exactly like the scalar multiply and add
//     (no fused multiply-add: SciPy's two roundings), at HALF the vector instructions - a wave64 float32 instruction occupies its
//     SIMD for 4 cycles packed or not, and the scalar version of this kernel sat at 62 % VALU occupancy;
//   * 0.5 eight-byte LDS reads per multiply-add, every read aligned whatever the lane's first index, 32 consecutive lanes on at
//     most 31 consecutive pairs = 62 of the 64 banks: no conflict (ds_read_b64: 256 B/clk; the compiler's ds_read2_b64 merge is
//     half that rate on 32 banks and is defeated by passing a chain's address through an empty asm statement every step);
//   * consecutive lanes = consecutive outputs: stores are full 256-byte runs;
//   * a workgroup is G lane groups of `up` residues (G up lanes in whole waves: 441 -> 1 x 441 of 448 lanes, 160 -> 2 x 160 = 320)
//     streaming over one item's outputs in blocks of G x 8 steps; the next block's window goes
//     global -> LDS by LDS-DMA while the current one is computed (no staging registers, no ds_write; two LDS stages), one
//     barrier per block, and the wait before it leaves the block's own stores in flight (counted vmcnt).
// Device code only (the host emulation keeps exercising ssr_resample.h, which remains the kernel of every plan this one does not
// take: taps per phase != 21, up < 33 or > 1024, float64 signals).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssr_resample.h"

#define SSR_RC_JB 8            /* outputs per lane and block (JB / 2 packed chains) */

struct SsrResampleRcParams {
  const float* in;
  const int64_t* in_off;
  const int32_t* in_len;
  const int64_t* out_off;
  const int32_t* out_len;
  int up, down, n_taps, n_pre_remove;
  const float* taps;
  int blocks_per_chunk, n_chunks;     // a workgroup walks `blocks_per_chunk` blocks of JB steps of ONE item
  int stage_floats;                   // floats per LDS stage (2 per pair; a multiple of 64)
  int groups;                         // G: lane groups of `up` residues per workgroup; a block is G * JB steps, group g its steps g JB ..
  float* out;
};

// LDS-DMA of one dword per lane: lane l's dword lands at LDS byte address `lds_wave_base` + 4 l (M0 = the wave-uniform base).
// As inline asm, not __builtin_amdgcn_global_load_lds: with the builtin in flight the compiler's wait-count pass treats every
// LDS wait of the loop as a wait for ALL outstanding LDS operations ("pending flat": s_waitcnt lgkmcnt(0) in front of every second
// step of the multiply-add chains, i.e. the one-step-ahead ds_read prefetch was waited for at once).  The kernels order the
// transfer themselves - counted s_waitcnt vmcnt before the barrier that publishes the stage.  M0 has no other user here.
__device__ __forceinline__ void ssr_lds_dma_dword(const float* src_lane, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_wave_base), "v"(src_lane) : "memory");
}

// pairs a block's window holds: pair i = (x[lo + i], x[lo + i + down]), i < n_pairs; the chains of steps 0, 2, .., G JB - 2 start at
// pair (q(r) - q(0)) + j down and read HPP pairs
__host__ __device__ inline int ssr_rc_pairs(int up, int down, int hpp, int groups) {
  return (groups * SSR_RC_JB - 2) * down + (int)(((int64_t)(up - 1) * down + (up - 1)) / up) + hpp + 2;
}

template <int HPP>
__device__ __forceinline__ void ssr_resample_rc_body(const SsrResampleRcParams& p, char* smem) {
  constexpr int JB = SSR_RC_JB, NC = JB / 2;
  typedef float ssr_v2f __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) float lds_float;              // 32-bit LDS addresses (a generic pointer costs a register pair)
  typedef __attribute__((address_space(3))) ssr_v2f lds_float2;
  const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
  const int item = (int)blockIdx.x / p.n_chunks, chunk = (int)blockIdx.x % p.n_chunks;
  const int up = p.up, down = p.down;
  const int n_in = p.in_len[item], n_out = p.out_len[item];
  const float* x = p.in + p.in_off[item];
  float* y = p.out + p.out_off[item];
  const int steps = (n_out + up - 1) / up;                              // outputs r + j up, j < steps
  const int G = p.groups, SB = G * JB;                                  // steps per block
  const int blk0 = chunk * p.blocks_per_chunk;
  if (blk0 * SB >= steps) return;
  int blk1 = blk0 + p.blocks_per_chunk;
  if (blk1 * SB > steps) blk1 = (steps + SB - 1) / SB;
  lds_float* lds = (lds_float*)smem;
  const int STAGE = p.stage_floats;

  // lane constants: phase, first input index, taps (k ascending = input ascending = tap index descending)
  // lane -> (group g, residue r): `up` = 160 fills 2 x 160 = 320 lanes = five whole waves (three waves per group of one would
  // leave 17 % of the lanes idle); lanes past G up compute group 0 / residue 0 again and their stores are dropped
  const bool active = tid < G * up;
  const int g = active ? tid / up : 0;
  const int r = active ? tid - g * up : 0;
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
  const int n_pairs = ssr_rc_pairs(up, down, HPP, G);

  // Staging.  A block whose window lies inside the signal goes global -> LDS directly (LDS-DMA, 4 bytes per lane: a wave deposits 32
  // consecutive PAIRS per instruction - its even lanes fetch x[lo + i], its odd lanes x[lo + i + down]): no staging registers, no
  // ds_write, the transfer runs under the current block's multiply-adds.  The first / last blocks of an item, whose windows reach
  // outside [0, n_in), take ordinary loads with the zero extension upfirdn applies.
  auto stage = [&](int blk, int s) {
    const int lo = qmin0 + blk * SB * down - (HPP - 1);
    lds_float* a = lds + s * STAGE;
    if (lo >= 0 && lo + n_pairs + 64 + down <= n_in) {                    // block-uniform
      const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
      const float* src = x + lo + (lane >> 1) + ((lane & 1) ? down : 0);
      for (int i0 = wave * 32; i0 < n_pairs; i0 += nw * 32)               // wave-uniform trip count
        ssr_lds_dma_dword(src + i0, __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(a + 2 * i0)));
    } else {
      for (int i = tid; i < 2 * n_pairs; i += nt) {
        const int g = lo + (i >> 1) + ((i & 1) ? down : 0);
        a[i] = (g >= 0 && g < n_in) ? x[g] : 0.0f;
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
      const lds_float* a = lds + s * STAGE;
      // NC packed chains: chain c = outputs of steps g JB + 2 c and g JB + 2 c + 1 of this block; its pair k sits at pair index
      // (q0 - qmin0) + (g JB + 2 c) down + k.  The order is pinned through the data: step k + 1's loads are issued before step k's
      // multiply-adds, and a chain's address goes through an empty asm statement every step (opaque bases cannot be merged into
      // ds_read2_b64).
      ssr_v2f acc[NC];
      unsigned ad[NC];                                                    // LDS byte address of the chain's first pair
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        ad[c] = (unsigned)(uintptr_t)(a + 2 * ((q0 - qmin0) + (g * JB + 2 * c) * down));
        acc[c] = (ssr_v2f){0.0f, 0.0f};
      }
      ssr_v2f cur[NC], nxt[NC];
      auto fetch = [&](ssr_v2f* dst, int k) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          asm volatile("" : "+v"(ad[c]));
          dst[c] = *(const lds_float2*)(uintptr_t)(ad[c] + 8u * (unsigned)k);
        }
      };
      fetch(cur, 0);
      static_assert(NC == 4, "the asm below");
#pragma unroll
      for (int k = 0; k < HPP; ++k) {
        if (k + 1 < HPP) fetch(nxt, k + 1);
        // Step k of the four chains: products first, sums second (a packed float32 instruction that reads the result of the one
        // just before it costs wait states - the compiler's own schedule paired every product with its sum and padded each pair
        // with two s_nop), the tap broadcast to both halves through op_sel (taps stay 11 register pairs instead of 21).
        // v_pk_mul_f32 then v_pk_add_f32: every half is multiplied, rounded, added, rounded - SciPy's two roundings per tap.
        ssr_v2f p0, p1, p2, p3;
        const ssr_v2f tp = {tap[k & ~1], tap[(k | 1) < HPP ? (k | 1) : k]};
        if (k & 1)
          asm volatile("v_pk_mul_f32 %4, %8, %12 op_sel:[0,1]\n\tv_pk_mul_f32 %5, %9, %12 op_sel:[0,1]\n\t"
                       "v_pk_mul_f32 %6, %10, %12 op_sel:[0,1]\n\tv_pk_mul_f32 %7, %11, %12 op_sel:[0,1]\n\t"
                       "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                       : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
                       : "v"(cur[0]), "v"(cur[1]), "v"(cur[2]), "v"(cur[3]), "v"(tp));
        else
          asm volatile("v_pk_mul_f32 %4, %8, %12 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %5, %9, %12 op_sel_hi:[1,0]\n\t"
                       "v_pk_mul_f32 %6, %10, %12 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %7, %11, %12 op_sel_hi:[1,0]\n\t"
                       "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                       : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
                       : "v"(cur[0]), "v"(cur[1]), "v"(cur[2]), "v"(cur[3]), "v"(tp));
        if (k + 1 < HPP) {
#pragma unroll
          for (int c = 0; c < NC; ++c) cur[c] = nxt[c];
        }
      }
      // JB stores per wave, ALWAYS issued (a buffer view drops what lies past the item's end or belongs to a lane past `up`): the
      // wait below counts on exactly JB vector-memory instructions being younger than the LDS-DMA of the next block
      const int m0 = r + (blk * SB + g * JB) * up;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        vy.st_raw(active ? 4 * (m0 + 2 * c * up) : -1, acc[c].x);
        vy.st_raw(active ? 4 * (m0 + (2 * c + 1) * up) : -1, acc[c].y);
      }
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
