// Developer / measurement tool (NOT part of libssrhip.so): what one MI355X sustains on a plain streaming read and on a copy, with
// 16-byte loads and stores and several independent loads in flight per lane - the "measured ceiling" SURVEY 8(d) asks to quote
// next to the 8 TB/s data-sheet peak.  bench.py loads tools/_build/libhbmprobe.so (built by __graft_entry__.build()) and runs both
// kernels on buffers larger than the 256 MB Infinity Cache in the process that prints the bench line.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_probe_read(const f4* __restrict__ a, float* sink, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  f4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f4 v0 = __builtin_nontemporal_load(a + i), v1 = __builtin_nontemporal_load(a + i + stride);
    const f4 v2 = __builtin_nontemporal_load(a + i + 2 * stride), v3 = __builtin_nontemporal_load(a + i + 3 * stride);
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; i < n4; i += stride) s0 += a[i];
  const f4 s = s0 + s1 + s2 + s3;
  if (s.x + s.y + s.z + s.w == 1.2345678e30f) sink[0] = 1.0f;          // keeps the loads alive; never true for finite data
}

__global__ __launch_bounds__(256) void k_probe_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f4 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
    b[i] = v0; b[i + stride] = v1; b[i + 2 * stride] = v2; b[i + 3 * stride] = v3;
  }
  for (; i < n4; i += stride) b[i] = a[i];
}

// mode 0: read `bytes` of a; mode 1: copy a -> b.  Enqueues ONE launch on `stream`; the caller times it with events.
extern "C" int hbm_probe_launch(const void* a, void* b, size_t bytes, int mode, int blocks, void* stream) {
  const size_t n4 = bytes / 16;
  if (!a || !b || n4 == 0 || blocks <= 0) return -1;
  if (mode == 0) hipLaunchKernelGGL(k_probe_read, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f4*)a, (float*)b, n4);
  else hipLaunchKernelGGL(k_probe_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f4*)a, (f4*)b, n4);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
