// K7 x 2, round 4: TWO polyphase resamplers in one kernel - scipy.signal.resample_poly(resample_poly(x, up1, down1), up2, down2) with the
// intermediate signal living in LDS only (BASELINE cfg-5: 16 kHz -> 44.1 kHz -> 48 kHz = 441/160 then 160/147; the reference's
// librosa.resample(res_type="polyphase") call sites, ssr_eval/eval.py:144-150).  Both stages are ssr_resample_rc.h's residue-class
// kernel - lane = residue of the output index modulo `up`, 21 taps in registers, the window as (x[i], x[i + down]) PAIRS read with one
// ds_read_b64 per packed multiply + packed add, SciPy's two roundings per tap - hence SciPy's bits for the intermediate and, from
// those, for the result.  Unfused, the chain moves 30.6 GB per 12,500 utterances of 4 s (12.8 GB algorithmic): the 44.1 kHz signal
// is written and read back once (17.6 GB).
//
// Geometry.  JB1 up1 = G2 JB2 down2 (8 x 441 = 2 x 12 x 147 = 3528): a block of 8 stage-1 steps produces exactly the 3528 intermediate
// samples that 24 stage-2 steps consume.  A workgroup is 7 PRODUCER waves (441 residues of stage 1) and 5 CONSUMER waves (2 lane
// groups x 160 residues of stage 2) streaming over one item's blocks:
//   iteration i:  producers compute block P = b0 - 1 + i of the intermediate from stage s = i & 1 of the input window (the next
//                 window is on its way global -> LDS by LDS-DMA) and deposit every sample TWICE in pair buffer s - as .x of pair j
//                 and as .y of pair j - down2 -, the last 20 also in a carry array; consumers compute block P - 1 of the output from
//                 pair buffer s ^ 1 and store it;
//   one barrier;  the carry (the 20 samples both windows share: HPP - 1) enters the head of the next buffer at the start of the
//                 next producer iteration.
// The two roles never touch the same buffer between two barriers; a role's waves run ahead of the other's inside an iteration as
// the SIMD schedules them.  Zero extension (upfirdn's, at both ends of both signals): the input window's slow staging path, and
// the producers write 0 for intermediate indices outside [0, mid_len).
// LDS: 2 x 9.2 KB (input windows) + 2 x 28.4 KB (pair buffers) + carry = 75.5 KB: two workgroups (24 waves) per CU.
// Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssr_resample_rc.h"

#define SSR_RCC_JB1 8            /* stage-1 steps per block (4 packed chains per lane) */
#define SSR_RCC_G2 2             /* stage-2 lane groups */
#define SSR_RCC_JB2 12           /* stage-2 steps per lane group and block (2 passes of 3 packed chains) */
#define SSR_RCC_NT1 448          /* producer threads (up1 <= 448) */
#define SSR_RCC_NT2 320          /* consumer threads (2 up2 <= 320) */
#define SSR_RCC_NT (SSR_RCC_NT1 + SSR_RCC_NT2)

struct SsrResampleChainParams {
  const float* in;
  const int64_t* in_off;
  const int32_t* in_len;
  const int32_t* mid_len;             // [n_items] length of the intermediate signal (resample_poly's n_out for stage 1)
  const int64_t* out_off;
  const int32_t* out_len;
  int up1, down1, n_taps1, npr1;
  const float* taps1;
  int up2, down2, n_taps2, npr2;
  const float* taps2;
  int blocks_per_chunk, n_chunks;     // a workgroup walks `blocks_per_chunk` blocks (G2 JB2 stage-2 steps each) of ONE item
  int x_stage_floats;                 // floats per LDS stage of the stage-1 window (2 per pair; whole 32-pair deposits)
  int y_pairs;                        // pairs per intermediate buffer: JB1 up1 + HPP - 1, rounded up to 8
  float* out;
};

typedef float ssr_rcc_v2f __attribute__((ext_vector_type(2)));
template <bool M> struct SsrRccMask { static constexpr bool value = M; };     // deposit(): zero the samples outside [0, mid_len)?

// NC packed chains of HPP taps: chain c reads pair k at LDS byte address ad[c] + 8 k; acc[c] = (output of the chain's even step,
// of its odd step).  The loop of ssr_resample_rc.h (see there for why the order is pinned through the data and the step is one asm
// block: products first, sums second, the tap broadcast through op_sel), for 3 or 4 chains.
template <int NC, int HPP>
__device__ __forceinline__ void ssr_rcc_chains(unsigned (&ad)[NC], const float (&tap)[HPP], ssr_rcc_v2f (&acc)[NC]) {
  typedef __attribute__((address_space(3))) ssr_rcc_v2f lds_float2;
  static_assert(NC == 3 || NC == 4, "asm blocks below");
  ssr_rcc_v2f cur[NC], nxt[NC];
  auto fetch = [&](ssr_rcc_v2f* dst, int k) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      asm volatile("" : "+v"(ad[c]));
      dst[c] = *(const lds_float2*)(uintptr_t)(ad[c] + 8u * (unsigned)k);
    }
  };
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = (ssr_rcc_v2f){0.0f, 0.0f};
  fetch(cur, 0);
#pragma unroll
  for (int k = 0; k < HPP; ++k) {
    if (k + 1 < HPP) fetch(nxt, k + 1);
    const ssr_rcc_v2f tp = {tap[k & ~1], tap[(k | 1) < HPP ? (k | 1) : k]};
    // (the products overwrite the samples: `cur` is dead after the step - four registers pairs fewer than separate product temporaries)
    if constexpr (NC == 4) {
      if (k & 1)
        asm volatile("v_pk_mul_f32 %4, %4, %8 op_sel:[0,1]\n\tv_pk_mul_f32 %5, %5, %8 op_sel:[0,1]\n\t"
                     "v_pk_mul_f32 %6, %6, %8 op_sel:[0,1]\n\tv_pk_mul_f32 %7, %7, %8 op_sel:[0,1]\n\t"
                     "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])
                     : "v"(tp));
      else
        asm volatile("v_pk_mul_f32 %4, %4, %8 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %5, %5, %8 op_sel_hi:[1,0]\n\t"
                     "v_pk_mul_f32 %6, %6, %8 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %7, %7, %8 op_sel_hi:[1,0]\n\t"
                     "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])
                     : "v"(tp));
    } else {
      if (k & 1)
        asm volatile("v_pk_mul_f32 %3, %3, %6 op_sel:[0,1]\n\tv_pk_mul_f32 %4, %4, %6 op_sel:[0,1]\n\t"
                     "v_pk_mul_f32 %5, %5, %6 op_sel:[0,1]\n\t"
                     "v_pk_add_f32 %0, %0, %3\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %5"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2])
                     : "v"(tp));
      else
        asm volatile("v_pk_mul_f32 %3, %3, %6 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %4, %4, %6 op_sel_hi:[1,0]\n\t"
                     "v_pk_mul_f32 %5, %5, %6 op_sel_hi:[1,0]\n\t"
                     "v_pk_add_f32 %0, %0, %3\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %5"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2])
                     : "v"(tp));
    }
    if (k + 1 < HPP) {
#pragma unroll
      for (int c = 0; c < NC; ++c) cur[c] = nxt[c];
    }
  }
}

template <int HPP>
__device__ __forceinline__ void ssr_resample_chain_body(const SsrResampleChainParams& p, char* smem) {
  constexpr int JB1 = SSR_RCC_JB1, G2 = SSR_RCC_G2, JB2 = SSR_RCC_JB2, SB2 = G2 * JB2, NT1 = SSR_RCC_NT1;
  typedef __attribute__((address_space(3))) float lds_float;
  const int tid = (int)threadIdx.x;
  const int item = (int)blockIdx.x / p.n_chunks, chunk = (int)blockIdx.x % p.n_chunks;
  const int n_in = p.in_len[item], n_mid = p.mid_len[item], n_out = p.out_len[item];
  const int up1 = p.up1, down1 = p.down1, up2 = p.up2, down2 = p.down2;
  const int steps2 = (n_out + up2 - 1) / up2;
  const int nblk = (steps2 + SB2 - 1) / SB2;
  const int b0 = chunk * p.blocks_per_chunk;
  if (b0 >= nblk) return;                                               // (workgroup-uniform)
  const int b1 = b0 + p.blocks_per_chunk < nblk ? b0 + p.blocks_per_chunk : nblk;
  const int n_iter = b1 - b0 + 2;
  const int B = JB1 * up1;                                              // intermediate samples per block (= SB2 down2)

  lds_float* xs = (lds_float*)smem;                                     // [2][x_stage_floats]
  lds_float* yb = xs + 2 * p.x_stage_floats;                            // [2][2 y_pairs]
  lds_float* carry = yb + 4 * p.y_pairs;                                // [2][32]
  const int YB = 2 * p.y_pairs;
  // stage 2's window of block b starts at intermediate index lo2(b) = qmin2 + b B - (HPP - 1); stage-1 block P covers the indices
  // [P B + qmin2, (P + 1) B + qmin2): pair index of its sample (residue r, step s) in its iteration's buffer is r + s up1 + (HPP - 1)
  const int qmin2 = (int)(((unsigned)p.npr2 * (unsigned)down2) / (unsigned)up2);

  if (tid < NT1) {
    // ------------------------------------------------------------------------------------------------ producers (stage 1)
    const float* x = p.in + p.in_off[item];
    const bool active = tid < up1;
    const int r = active ? tid : 0;
    const int shift = qmin2 + p.npr1;                                   // output origin moved by qmin2 (see above)
    const unsigned t0 = (unsigned)(r + shift) * (unsigned)down1;
    const int q0 = (int)(t0 / (unsigned)up1), ph = (int)(t0 - (unsigned)q0 * (unsigned)up1);
    float tap[HPP];
    {
      const SsrView<float> vt(p.taps1, p.n_taps1);
#pragma unroll
      for (int k = 0; k < HPP; ++k) tap[k] = vt.at_or_zero((unsigned)(ph + (HPP - 1 - k) * up1));
    }
    const int qmin1 = (int)(((unsigned)shift * (unsigned)down1) / (unsigned)up1);
    const int n_pairs = ssr_rc_pairs(up1, down1, HPP, 1);
    auto stage = [&](int P, int s) {
      const int lo = qmin1 + P * JB1 * down1 - (HPP - 1);
      lds_float* a = xs + s * p.x_stage_floats;
      if (lo >= 0 && lo + n_pairs + 64 + down1 <= n_in) {                // block-uniform
        const int wave = tid >> 6, lane = tid & 63;
        const float* src = x + lo + (lane >> 1) + ((lane & 1) ? down1 : 0);
        for (int i0 = wave * 32; i0 < n_pairs; i0 += (NT1 / 64) * 32)
          ssr_lds_dma_dword(src + i0, __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(a + 2 * i0)));
      } else {
        for (int i = tid; i < 2 * n_pairs; i += NT1) {
          const int gi = lo + (i >> 1) + ((i & 1) ? down1 : 0);
          a[i] = (gi >= 0 && gi < n_in) ? x[gi] : 0.0f;
        }
      }
    };
    stage(b0 - 1, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    for (int it = 0; it < n_iter; ++it) {
      const int P = b0 - 1 + it, s = it & 1;
      if (P < b1) {
        if (P + 1 < b1) stage(P + 1, s ^ 1);
        lds_float* yw = yb + s * YB;
        if (it >= 1 && tid < HPP - 1) yw[2 * tid] = carry[32 * (s ^ 1) + tid];      // the samples the two windows share
        unsigned ad[4];
        ssr_rcc_v2f acc[4];
        const lds_float* a = xs + s * p.x_stage_floats;
#pragma unroll
        for (int c = 0; c < 4; ++c) ad[c] = (unsigned)(uintptr_t)(a + 2 * ((q0 - qmin1) + 2 * c * down1));
        ssr_rcc_chains<4, HPP>(ad, tap, acc);
        if (active) {
          // Pair index of (r, step st): j = r + st up1 + HPP - 1.  j >= down2 (the sample is also the .y of pair j - down2) fails for
          // step 0 only (up1 = 3 down2): those lanes' second write goes to a spare slot; j >= B (the carry) needs step 7.  The
          // lane's base address passes through an empty asm statement every iteration: hoisted out of the loop, the sixteen
          // per-step addresses and conditions cost sixteen registers and spilled.
          const int m0 = qmin2 + r + P * B;                             // intermediate index of (r, step 0)
          unsigned wb = (unsigned)(uintptr_t)(yw + 2 * (r + HPP - 1));
          asm volatile("" : "+v"(wb));
          const unsigned spare = (unsigned)(uintptr_t)(carry + 32 * s + 24 + (tid & 7));
          const bool inside = P * B + qmin2 >= 0 && (P + 1) * B + qmin2 <= n_mid;       // block-uniform: no sample to zero
          auto deposit = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int st = 2 * c + h;
                float v = h ? acc[c].y : acc[c].x;
                if constexpr (MASKED) v = ((unsigned)(m0 + st * up1) < (unsigned)n_mid) ? v : 0.0f;
                const unsigned ax = wb + 8u * (unsigned)(st * up1);
                *(lds_float*)(uintptr_t)ax = v;
                const unsigned ay = ax - 8u * (unsigned)down2 + 4u;
                *(lds_float*)(uintptr_t)((st == 0 && r + HPP - 1 < down2) ? spare : ay) = v;
                if (st == JB1 - 1 && r >= up1 - (HPP - 1)) carry[32 * s + (r - (up1 - (HPP - 1)))] = v;
              }
            }
          };
          if (inside) deposit(SsrRccMask<false>{}); else deposit(SsrRccMask<true>{});
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
  } else {
    // ------------------------------------------------------------------------------------------------ consumers (stage 2)
    const int ct = tid - NT1;
    float* y = p.out + p.out_off[item];
    const bool active = ct < G2 * up2;
    const int g = active ? ct / up2 : 0;
    const int r = active ? ct - g * up2 : 0;
    const unsigned t0 = (unsigned)(r + p.npr2) * (unsigned)down2;
    const int q0 = (int)(t0 / (unsigned)up2), ph = (int)(t0 - (unsigned)q0 * (unsigned)up2);
    float tap[HPP];
    {
      const SsrView<float> vt(p.taps2, p.n_taps2);
#pragma unroll
      for (int k = 0; k < HPP; ++k) tap[k] = vt.at_or_zero((unsigned)(ph + (HPP - 1 - k) * up2));
    }
    const SsrRwView<float> vy(y, n_out);
    const int vo0 = active ? 4 * (r + g * JB2 * up2) : 0x40000000;        // lane part of the store offset; an idle lane's stays out of range
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    for (int it = 0; it < n_iter; ++it) {
      const int Cb = b0 - 2 + it;                                       // the block produced in the previous iteration
      if (it >= 2) {
        const lds_float* a = yb + ((it - 1) & 1) * YB;
#pragma unroll
        for (int ps = 0; ps < JB2 / 6; ++ps) {                          // passes of 3 packed chains = 6 steps
          unsigned ad[3];
          ssr_rcc_v2f acc[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) ad[c] = (unsigned)(uintptr_t)(a + 2 * ((q0 - qmin2) + (g * JB2 + ps * 6 + 2 * c) * down2));
          ssr_rcc_chains<3, HPP>(ad, tap, acc);
          const int mb = 4 * (Cb * SB2 + ps * 6) * up2;                   // wave-uniform part
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            vy.st_raw(vo0 + mb + 8 * c * up2, acc[c].x);
            vy.st_raw(vo0 + mb + 4 * (2 * c + 1) * up2, acc[c].y);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
  }
}
