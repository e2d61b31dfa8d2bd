// Developer micro-benchmark (gfx950): issue cadence of dependent / independent FP64 and FP32 VALU chains for one wave and for
// two waves per SIMD, and the rate of the s_memtime counter against the 100 MHz s_memrealtime.
//   hipcc --offload-arch=gfx950 -O2 -o valu_cadence valu_cadence.hip && ./valu_cadence
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE> __global__ void k(unsigned long long* out, int iters) {
  double a = 1.0000001, b = 1e-9;
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, fa = 1.0001f;
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));) }
    if (MODE == 1) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                                       "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                                       : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));) }
    if (MODE == 2) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x0) : "v"(b));) }
    if (MODE == 3) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_mul_f64 %1, %1, %9\n v_add_f64 %2, %2, %8\n v_mul_f64 %3, %3, %9\n"
                                       "v_add_f64 %4, %4, %8\n v_mul_f64 %5, %5, %9\n v_add_f64 %6, %6, %8\n v_mul_f64 %7, %7, %9"
                                       : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));) }
    if (MODE == 4) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(fa));) }
    if (MODE == 5) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n"
                                       "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                                       : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa));) }
  }
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0;
  }
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f0 + f1 + f2 + f3 == 12345.678) out[0] = 0;
}

template <int MODE> void run(const char* name, unsigned long long* d, int threads, int blocks) {
  const int iters = 20000, n = 64 * iters, waves = blocks * threads / 64;
  k<MODE><<<blocks, threads>>>(d, 10);
  (void)hipDeviceSynchronize();
  k<MODE><<<blocks, threads>>>(d, iters);
  (void)hipDeviceSynchronize();
  static unsigned long long h[2 * 256 * 16];
  (void)hipMemcpy(h, d, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull, tmax = 0, rmax = 0;
  for (int w = 0; w < waves; ++w) { tmin = h[2 * w] < tmin ? h[2 * w] : tmin; tmax = h[2 * w] > tmax ? h[2 * w] : tmax; rmax = h[2 * w + 1] > rmax ? h[2 * w + 1] : rmax; }
  printf("%-32s %4d thr x %3d blk: %6.2f .. %6.2f shader cycles / instr / wave, clock %.0f MHz, SIMD rate 1 instr / %.2f cycles\n", name, threads, blocks,
         (double)tmin / n, (double)tmax / n, 100.0 * h[0] / h[1], (double)tmax / n / (threads / 256.0 > 1 ? threads / 256.0 : 1));
}

int main() {
  unsigned long long* d;
  (void)hipMalloc(&d, 1 << 20);
  const int cfgs[6][2] = {{64, 1}, {256, 1}, {512, 1}, {1024, 1}, {512, 256}, {1024, 256}};
  for (auto& c : cfgs) {
    run<0>("v_fma_f64 dependent chain", d, c[0], c[1]);
    run<1>("v_fma_f64 8 independent chains", d, c[0], c[1]);
    run<3>("v_add/mul_f64 8 independent", d, c[0], c[1]);
    run<4>("v_fma_f32 dependent chain", d, c[0], c[1]);
    run<5>("v_fma_f32 4 independent chains", d, c[0], c[1]);
  }
  return 0;
}
