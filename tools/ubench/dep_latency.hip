// Developer micro-benchmark: issue-to-issue latency of DEPENDENT vector instructions on gfx950, one wave per SIMD (what a serial
// recurrence such as ssr_iir.h's step sees).  C independent chains interleaved in one wave: cycles per instruction of a chain.
//   hipcc --offload-arch=gfx950 -O2 -o dep_latency dep_latency.hip && ./dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP, int C> __global__ __launch_bounds__(64) void k_chain(double* out, int iters, double c0, unsigned long long* cyc) {
  double x[C];
  float f[C];
  for (int c = 0; c < C; ++c) { x[c] = c0 + c + threadIdx.x; f[c] = (float)x[c]; }
  const double m = c0 * 1.0000001;
  const float mf = (float)m;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(m));
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(m));
        if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[c]) : "v"(m));
        if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(mf));
        if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[c]) : "v"(mf));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
  for (int c = 0; c < C; ++c) s += x[c] + (double)f[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the nine float64 operations of one sosfilt step (ssr_iir.h, pipelined order), G = 1: sixteen steps on sixteen register inputs per iteration
__global__ __launch_bounds__(64) void k_iir_steps(double* out, int iters, double c0) {
#pragma clang fp contract(off)
  float xs[16];
  for (int k = 0; k < 16; ++k) xs[k] = (float)(c0 * (k + 1) + threadIdx.x * 1e-3);
  const double b0 = 0.2 * c0, b1 = 0.4 * c0, b2 = 0.2 * c0, a1 = -0.3 * c0, a2 = 0.1 * c0;
  double z0 = 0.0, z1 = 0.0, acc = 0.0;
  for (int i = 0; i < iters; ++i) {
    double xin = (double)xs[0], p0 = b0 * xin;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double yo = p0 + z0;
      double xin_n = 0.0, p0_n = 0.0;
      if (k + 1 < 16) { xin_n = (double)xs[k + 1]; p0_n = b0 * xin_n; }
      z0 = (b1 * xin - a1 * yo) + z1;
      z1 = b2 * xin - a2 * yo;
      xin = xin_n; p0 = p0_n;
      if (k == 15) acc += yo;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(xs[k]));      // (opaque per iteration: the products with x are not loop-invariant)
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc + z0 + z1;
}
static void run_iir(int blocks) {
  double* out;
  (void)hipMalloc(&out, sizeof(double) * 64 * blocks);
  const int iters = 20000;
  hipLaunchKernelGGL(k_iir_steps, dim3(blocks), dim3(64), 0, 0, out, 10, 1.0);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_iir_steps, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("sosfilt step (9 float64 operations + 1 conversion, registers only), waves %4d: %.2f ns per step\n", blocks, 1e6 * ms / ((double)iters * 16));
  (void)hipFree(out);
}

template <int OP, int C> static void run(const char* name, int blocks) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * 64 * blocks); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  const int iters = 2000;
  hipLaunchKernelGGL((k_chain<OP, C>), dim3(blocks), dim3(64), 0, 0, out, 10, 1.0, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_chain<OP, C>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * 64;          // instructions per chain
  printf("%-12s chains %d, waves %4d: %.2f ns and %.1f shader-clock counts per instruction of a chain (%.2f ns per instruction issued)\n", name, C, blocks,
         1e6 * ms / n, (double)h / n, 1e6 * ms / (n * C));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 1>("v_add_f64", 1); run<0, 2>("v_add_f64", 1); run<0, 4>("v_add_f64", 1); run<0, 8>("v_add_f64", 1);
  run<1, 1>("v_mul_f64", 1); run<1, 4>("v_mul_f64", 1);
  run<2, 1>("v_fma_f64", 1); run<2, 4>("v_fma_f64", 1);
  run<3, 1>("v_add_f32", 1); run<3, 4>("v_add_f32", 1);
  run<4, 1>("v_fma_f32", 1);
  run<0, 1>("v_add_f64", 1024); run<0, 1>("v_add_f64", 2048); run<0, 4>("v_add_f64", 1024);
  run_iir(1); run_iir(256); run_iir(1024); run_iir(2048);
  return 0;
}
