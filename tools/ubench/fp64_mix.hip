// Developer micro-benchmark (gfx950): the ceiling of a kernel with k_stft_wave's FP64 instruction mix and LDS exchange pattern and
// NO global memory.  One "unit" = what the hot loop of k_stft_wave<double, false, true, true> issues per frame
// pair (static count of its loop body, tools/isa_stats.py: 620 v_add_f64, 234 v_mul_f64, 300 v_fma(c)_f64, 64 v_cvt_f64_f32,
// 80 ds_write2_b64, 64 ds_read2_b64, four LDS exchanges): here 620 + 248 + 308 + 64 FP64 instructions on 32 independent registers
// (no dependent pair closer than 32 instructions: ideal ILP) in four phases, each followed by 20 ds_write2_b64, a wait, 16
// ds_read2_b64 and a wait - one wave per workgroup with its own LDS slice, as in the kernel.  Prints units/s for 1, 2 and 3 waves
// per SIMD over the whole chip, the FP64 instruction rate per SIMD, and the sustained clock; second part: the issue rate of
// v_mfma_f64_16x16x4_f64 (is the FP64 matrix pipe any faster than the FP64 vector pipe?).
//   hipcc --offload-arch=gfx950 -O2 -o fp64_mix fp64_mix.hip && ./fp64_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool WITH_LDS, int MODE = 0> __global__ __launch_bounds__(64) void k_mix(unsigned long long* out, int units, double seed) {
  __shared__ double lds[64 * 16];          // 16 rows of 64 lanes: element e of lane l at e * 64 + l (conflict-free, as the kernel's exchanges)
  double x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = seed + threadIdx.x + i;
  float f[4] = {(float)seed, 1.5f, 2.5f, 3.5f};
  const double c0 = 1e-9, c1 = 1.0000001;
  unsigned base[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) base[j] = (unsigned)(unsigned long long)(__attribute__((address_space(3))) double*)(lds + threadIdx.x + 256 * j);
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int u = 0; u < units; ++u) {
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
      for (int g = 0; g < 31; ++g) {
#pragma unroll
        for (int k = 0; k < 5; ++k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[(g * 10 + k) & 31]) : "v"(c0));
#pragma unroll
        for (int k = 5; k < 7; ++k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[(g * 10 + k) & 31]) : "v"(c1));
#pragma unroll
        for (int k = 7; k < 9 + (g & 1); ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[(g * 10 + k) & 31]) : "v"(c1), "v"(c0));
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[(ph * 16 + k) & 31]) : "v"(f[k & 3]));
      if (WITH_LDS) {
#pragma unroll
        for (int k = 0; k < 20; ++k) {         // elements (2k, 2k + 1) mod 16: row group (2k mod 16) / 4, rows 64 and 128 eight-byte units apart
          if (MODE & 2) {                      // MODE bit 1: two ds_write_b64 instead of one ds_write2_b64
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(base[((2 * k) & 15) >> 2]), "v"(x[k & 31]), "n"(((2 * k) & 3) * 512) : "memory");
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(base[((2 * k) & 15) >> 2]), "v"(x[(k + 7) & 31]), "n"(((2 * k + 1) & 3) * 512) : "memory");
          } else {
            asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(base[((2 * k) & 15) >> 2]), "v"(x[k & 31]), "v"(x[(k + 7) & 31]),
                         "n"(((2 * k) & 3) * 64), "n"(((2 * k + 1) & 3) * 64) : "memory");
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (MODE & 1) {                          // MODE bit 0: two ds_read_b64 (256 B/clk in the guide's table) instead of one ds_read2_b64 (128)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[(2 * k) & 31]) : "v"(base[((2 * k) & 15) >> 2]), "n"(((2 * k) & 3) * 512) : "memory");
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[(2 * k + 1) & 31]) : "v"(base[((2 * k) & 15) >> 2]), "n"(((2 * k + 1) & 3) * 512) : "memory");
          } else {
            d2 v;                                  // (a 128-bit result: two doubles)
            asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(base[((2 * k) & 15) >> 2]), "n"(((2 * k) & 3) * 64),
                         "n"(((2 * k + 1) & 3) * 64) : "memory");
            x[(2 * k) & 31] = v.x; x[(2 * k + 1) & 31] = v.y;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
  if (s == 12345.678) out[0] = 0;
}

__global__ __launch_bounds__(64) void k_mfma64(unsigned long long* out, int iters, double seed) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const double va = seed + threadIdx.x, vb = 1.0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, a3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (a0.x + a1.x + a2.x + a3.x == 12345.678) out[0] = 0;
}

// issue rate of the two conversions (the windowed-sinc resampler issues three per tap next to four FP64 add / mul): WHICH 0 =
// v_cvt_f64_f32, 1 = v_cvt_f32_f64, 2 = v_add_f64 (the yardstick), 32 independent registers
template <int WHICH> __global__ __launch_bounds__(64) void k_cvt(unsigned long long* out, int iters, double seed) {
  double x[32];
  float f[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) { x[i] = seed + threadIdx.x + i; f[i] = (float)x[i]; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int u = 0; u < iters; ++u) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (WHICH == 0) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[i]) : "v"(f[i]));
      else if (WHICH == 1) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(x[i]));
      else asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i] + f[i];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s == 12345.678) out[0] = 0;
}

int main() {
  unsigned long long* d;
  (void)hipMalloc(&d, 1 << 22);
  static unsigned long long h[1 << 18];
  const int units = 2000, per_unit_f64 = 4 * (31 * 5 + 31 * 2 + 31 * 2 + 15 + 16);      // 1240
  for (int with_lds = 1; with_lds >= 0; --with_lds)
    for (int wps = 1; wps <= 3; ++wps) {
      const int blocks = 256 * 4 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        if (with_lds) k_mix<true><<<blocks, 64>>>(d, rep ? units : 10, 1.0); else k_mix<false><<<blocks, 64>>>(d, rep ? units : 10, 1.0);
        (void)hipDeviceSynchronize();
      }
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      if (with_lds) k_mix<true><<<blocks, 64>>>(d, units, 1.0); else k_mix<false><<<blocks, 64>>>(d, units, 1.0);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      (void)hipMemcpy(h, d, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
      unsigned long long cmax = 0;
      for (int b = 0; b < blocks; ++b) cmax = h[2 * b] > cmax ? h[2 * b] : cmax;
      const double units_s = (double)blocks * units / (ms * 1e-3);
      printf("%s, %d wave(s)/SIMD: %.1f M units/s on the chip, %.2f shader cycles per FP64 instruction and SIMD, clock %.0f MHz, %.1f TFLOP/s-equivalent (FP64 instr x 64 lanes)\n",
             with_lds ? "FP64 mix + 4 LDS exchanges/unit" : "FP64 mix alone              ", wps, units_s / 1e6,
             (double)cmax / ((double)units * per_unit_f64 * wps), 100.0 * h[0] / h[1], units_s * per_unit_f64 * 64 / 1e12);
    }
  // the same unit with the exchanges' DS instructions split: reads as 2 x ds_read_b64, writes as 2 x ds_write_b64, both (2 waves per SIMD)
  for (int mode = 1; mode <= 3; ++mode) {
    const int blocks = 256 * 4 * 2;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      if (mode == 1) k_mix<true, 1><<<blocks, 64>>>(d, units, 1.0); else if (mode == 2) k_mix<true, 2><<<blocks, 64>>>(d, units, 1.0); else k_mix<true, 3><<<blocks, 64>>>(d, units, 1.0);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("DS variant %d (%s, %s), 2 waves/SIMD: %.1f M units/s\n", mode, (mode & 1) ? "2 x ds_read_b64" : "ds_read2_b64", (mode & 2) ? "2 x ds_write_b64" : "ds_write2_b64",
           (double)blocks * units / (best * 1e-3) / 1e6);
  }
  for (int which = 0; which < 3; ++which) {
    const int blocks = 256 * 4 * 2, iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      if (which == 0) k_cvt<0><<<blocks, 64>>>(d, rep ? iters : 10, 1.0); else if (which == 1) k_cvt<1><<<blocks, 64>>>(d, rep ? iters : 10, 1.0);
      else k_cvt<2><<<blocks, 64>>>(d, rep ? iters : 10, 1.0);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, d, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    unsigned long long cmax = 0;
    for (int b = 0; b < blocks; ++b) cmax = h[b] > cmax ? h[b] : cmax;
    printf("%s, 2 waves/SIMD: %.2f shader cycles per instruction and SIMD\n", which == 0 ? "v_cvt_f64_f32" : which == 1 ? "v_cvt_f32_f64" : "v_add_f64    ",
           (double)cmax / ((double)iters * 32 * 2));
  }
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * 4 * wps, iters = 4000;
    k_mfma64<<<blocks, 64>>>(d, 10, 1.0);
    (void)hipDeviceSynchronize();
    k_mfma64<<<blocks, 64>>>(d, iters, 1.0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    unsigned long long cmax = 0;
    for (int b = 0; b < blocks; ++b) cmax = h[b] > cmax ? h[b] : cmax;
    printf("v_mfma_f64_16x16x4_f64, %d wave(s)/SIMD: %.1f shader cycles per instruction and SIMD = %.1f FLOP/cycle/SIMD (FP64 vector FMA: 32)\n", wps,
           (double)cmax / ((double)iters * 64 * wps), 2048.0 / ((double)cmax / ((double)iters * 64 * wps)));
  }
  return 0;
}
