// Block-execution abstraction shared by every kernel body in this directory.
//
// A kernel body is written once as a sequence of PHASES.  Inside a phase every thread of the
// workgroup runs the same statement on its own `tid`; threads only READ what earlier phases wrote
// and only WRITE locations no other thread touches in that phase.  Consecutive phases are separated
// by a workgroup barrier.
//
//  * hipcc (gfx950):   SSR_PHASE(...) = { statement } __syncthreads();
//  * g++ -DSSR_HOST_EMU: SSR_PHASE(...) = for (tid = 0 .. NT-1) { statement }   (sequential emulation)
//
// The emulation build exists so that the arithmetic of every kernel body (FFT passes, Bluestein,
// SSIM windows, polyphase indices ...) is parity-tested against the oracle on a CPU-only machine
// (tests/test_emu_*.py).  It is test infrastructure; the shipped library contains device code only.
#pragma once
#include <stdint.h>
#include <math.h>

#ifdef SSR_HOST_EMU
#include <vector>
#include <cstring>
#define SSR_DEV static inline
#define SSR_BODY static inline
#define SSR_MEMBER inline
#define SSR_HD static inline
#define SSR_SCHED_FENCE() do {} while (0)
#define SSR_SCHED_BARRIER() do {} while (0)
#define SSR_VMEM_DRAIN() do {} while (0)
template <typename V> static inline void ssr_touch(V&) {}
#define SSR_UNROLL
#define SSR_UNROLL4
struct SsrBlk { int nt; };
static inline void ssr_launder(SsrBlk&) {}
static inline unsigned ssr_launder_index(unsigned i) { return i; }
#define SSR_REGS(TYPE, name, blk) std::vector<TYPE> name((blk).nt)
#define SSR_PHASE(blk, regs, ...)                                  \
  for (int tid = 0; tid < (blk).nt; ++tid) {                       \
    auto& R = (regs)[tid]; (void)R;                                \
    __VA_ARGS__;                                                   \
  }
#define SSR_WPHASE SSR_PHASE
// inside a phase: dst[tid / 64] = sum of `val` over the lanes of that wave (host: tids run in ascending order)
#define SSR_WAVE_SUM_STORE(tid, NT_, val, dst)              \
  do {                                                      \
    if (((tid) & 63) == 0) (dst)[(tid) >> 6] = 0.0;         \
    (dst)[(tid) >> 6] += (val);                             \
  } while (0)
#define SSR_WAVE_SUM_ADD(tid, NT_, val, dst) do { (dst)[(tid) >> 6] += (val); } while (0)
// lane-private accumulator in LDS: *p += v, nothing returned
#define SSR_LDS_ACCUM(p, v) do { *(p) += (v); } while (0)
// inside a phase: SSR_WAVE_ANY(pred) = "pred holds on some lane of this wave" (device: a scalar; host: the lane's own
// pred, folded over the wave by the store); SSR_WAVE_FLAG_STORE: dst[tid / 64] = that flag
#define SSR_WAVE_ANY(pred) ((pred) ? 1 : 0)
#define SSR_WAVE_FLAG_STORE(tid, flag, dst)                 \
  do {                                                      \
    if (((tid) & 63) == 0) (dst)[(tid) >> 6] = 0;           \
    if (flag) (dst)[(tid) >> 6] = 1;                        \
  } while (0)
// (the wave-wide vote is taken BEFORE the single-lane store: inside that branch only one lane would be voting)
#define SSR_WAVE_ANY_STORE(tid, pred, dst)                  \
  do {                                                      \
    const int any_ = SSR_WAVE_ANY(pred);                    \
    SSR_WAVE_FLAG_STORE(tid, any_, dst);                    \
  } while (0)
// wave number of a thread as a value the compiler knows to be wave-uniform (scalar register on the device)
static inline int ssr_wave_of(int tid) { return tid >> 6; }
static inline float ssr_fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float ssr_fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline double ssr_fmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double ssr_fadd_rn(double a, double b) { volatile double r = a + b; return r; }
#else
#include <hip/hip_runtime.h>
#if defined(SSR_FULL_BARRIER)
#define SSR_BARRIER() __syncthreads()
#else
// Workgroup barrier that orders LDS traffic only.  All cross-thread hand-offs inside a kernel body go
// through LDS; __syncthreads() would additionally drain every outstanding GLOBAL access (s_waitcnt
// vmcnt(0)), i.e. expose the full HBM store/load latency at every phase boundary.
#define SSR_BARRIER()                                                   \
  do {                                                                  \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");     \
    __builtin_amdgcn_s_barrier();                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");     \
  } while (0)
#endif
#define SSR_DEV __device__ __forceinline__
#define SSR_BODY __device__ __forceinline__
#define SSR_MEMBER __device__ __forceinline__
#define SSR_HD __host__ __device__ __forceinline__
// compiler-only fence: keeps the scheduler from hoisting the NEXT group of loads above this point (caps
// the number of table values live at once in the 16-points-per-thread Bluestein phases)
#define SSR_SCHED_FENCE() asm volatile("" ::: "memory")
// instruction-scheduler barrier: nothing is moved across this point
#define SSR_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// s_waitcnt vmcnt(0) (expcnt / lgkmcnt untouched): every vector-memory operation of the wave has completed
#define SSR_VMEM_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)
// "this register is needed here": a zero-instruction read-modify-write that makes the compiler place the wait for an
// outstanding load that fills it at this point - and makes the value a product of this point of the program, no
// longer of the load (nothing downstream is tied to the memory counters any more)
SSR_DEV void ssr_touch(float& v) { asm volatile("" : "+v"(v)); }
SSR_DEV void ssr_touch(double& v) { asm volatile("" : "+v"(v)); }
// full unroll (usable inside SSR_PHASE macro arguments): register arrays must only be indexed statically
#define SSR_UNROLL _Pragma("unroll")
#define SSR_UNROLL4 _Pragma("unroll 4")
struct SsrBlk { int tid; };
// Make the thread index opaque to the optimiser for the code that follows.  Loop-invariant code motion
// otherwise hoists EVERY per-register LDS / table address of every pass out of the frame loop; at 16
// points per thread that is several hundred values which end up in scratch memory.
SSR_DEV void ssr_launder(SsrBlk& b) { asm volatile("" : "+v"(b.tid)); }
// the same for one index: the address built from it is recomputed where it is used instead of living (or being
// spilled) across the whole frame loop
SSR_DEV unsigned ssr_launder_index(unsigned i) { asm volatile("" : "+v"(i)); return i; }
#define SSR_REGS(TYPE, name, blk) TYPE name
#define SSR_PHASE(blk, regs, ...)                                  \
  {                                                                \
    const int tid = (blk).tid; auto& R = (regs); (void)R; (void)tid; \
    __VA_ARGS__;                                                   \
  }                                                                \
  SSR_BARRIER();
// Phase of a body whose hand-offs stay inside ONE wave (one-wave workgroups, or per-wave LDS arrays): a wave-scope ordering
// point - it constrains the compiler and emits nothing (same-wave LDS traffic is ordered by the hardware), where SSR_PHASE's
// workgroup fence costs an s_waitcnt lgkmcnt(0) per boundary.
#define SSR_WAVE_SYNC()                                             \
  do {                                                              \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          \
    __builtin_amdgcn_wave_barrier();                                \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          \
  } while (0)
#define SSR_WPHASE(blk, regs, ...)                                   \
  {                                                                  \
    const int tid = (blk).tid; auto& R = (regs); (void)R; (void)tid; \
    __VA_ARGS__;                                                     \
  }                                                                  \
  SSR_WAVE_SYNC();
// wave-level sum through cross-lane shuffles (no LDS round trip, no barrier); W = active lanes (<= 64)
template <int W> SSR_DEV double ssr_wave_sum(double v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
  return v;
}
// wave number as a SCALAR (recomputed where it is used: a per-lane copy would be loop-invariant, get hoisted out of the
// frame loop and occupy - or spill - a vector register for the whole kernel)
SSR_DEV int ssr_wave_index(int tid) { return __builtin_amdgcn_readfirstlane(tid) >> 6; }
SSR_DEV int ssr_wave_of(int tid) { return ssr_wave_index(tid); }
#define SSR_WAVE_SUM_STORE(tid, NT_, val, dst)                                     \
  do {                                                                             \
    const double s_ = ssr_wave_sum<((NT_) < 64 ? (NT_) : 64)>(val);                \
    if (((tid) & 63) == 0) (dst)[ssr_wave_index(tid)] = s_;                        \
  } while (0)
// lane-private accumulator in LDS: *p += v as one ds_add_f64 (nothing comes back: no register, no wait)
#define SSR_LDS_ACCUM(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT))
#define SSR_WAVE_SUM_ADD(tid, NT_, val, dst)                                       \
  do {                                                                             \
    const double s_ = ssr_wave_sum<((NT_) < 64 ? (NT_) : 64)>(val);                \
    if (((tid) & 63) == 0) (dst)[ssr_wave_index(tid)] += s_;                       \
  } while (0)
#define SSR_WAVE_ANY(pred) (__builtin_amdgcn_ballot_w64(pred) != 0ull ? 1 : 0)
#define SSR_WAVE_FLAG_STORE(tid, flag, dst)                                        \
  do {                                                                             \
    const int flag_ = (flag);   /* evaluated by the WHOLE wave, outside the branch */ \
    if (((tid) & 63) == 0) (dst)[ssr_wave_index(tid)] = flag_;                     \
  } while (0)
// (the wave-wide vote is taken BEFORE the single-lane store: inside that branch only one lane would be voting)
#define SSR_WAVE_ANY_STORE(tid, pred, dst)                  \
  do {                                                      \
    const int any_ = SSR_WAVE_ANY(pred);                    \
    SSR_WAVE_FLAG_STORE(tid, any_, dst);                    \
  } while (0)
// Separately rounded multiply and add.  HIP's __fmul_rn/__fadd_rn are plain `*` / `+` and hipcc's default
// -ffp-contract=fast would fuse them into one v_fma_f32; the pragma strips the `contract` flag from
// these two instructions so they can never be fused (needed for bit-identity with SciPy's upfirdn).
SSR_DEV float ssr_fmul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
SSR_DEV float ssr_fadd_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
SSR_DEV double ssr_fmul_rn(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
SSR_DEV double ssr_fadd_rn(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}
#endif

// Two float32 values in a 64-bit register pair, processed by ONE packed instruction (v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32: a wave-wide packed float32 instruction takes the same issue slot as a scalar-per-lane one).  Host emulation: a
// plain pair with elementwise operators.
#ifdef SSR_HOST_EMU
struct f2 { float x, y; };
static inline f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
static inline f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
static inline f2 operator*(f2 a, f2 b) { return {a.x * b.x, a.y * b.y}; }
static inline f2 f2_fma(f2 a, f2 b, f2 c) { return {fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#else
typedef float f2 __attribute__((ext_vector_type(2)));
SSR_DEV f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif
SSR_DEV f2 f2_make(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
SSR_DEV f2 f2_splat(float v) { return f2_make(v, v); }

// magnitude bits of a sample (sign dropped): non-zero iff the sample is not +-0 (NaN and Inf count as non-zero)
SSR_DEV unsigned ssr_mag_bits(float v) {
  unsigned u;
  memcpy(&u, &v, 4);
  return u << 1;
}
SSR_DEV unsigned ssr_mag_bits(double v) {
  unsigned long long u;
  memcpy(&u, &v, 8);
  return (unsigned)(u & 0xffffffffu) | (unsigned)((u >> 32) & 0x7fffffffu);
}

// ------------------------------------------------------------------------------------------------
// A read-only device array as the streaming loads see it: element `idx` (per lane, 32-bit) past a block-uniform
// element offset.  Device: a raw buffer resource, i.e. scalar base + scalar offset + 32-bit lane offset in ONE
// instruction (the compiler otherwise keeps a per-lane 64-bit pointer per load and advances each one with a vector
// add every loop trip).  Host emulation: a plain pointer.  Arrays are limited to 4 GiB (checked on the host).
template <typename E> struct SsrView {
#ifdef SSR_HOST_EMU
  const E* base; int64_t n;
  SSR_MEMBER SsrView(const E* p, int64_t n_elems) : base(p), n(n_elems) {}
  SSR_MEMBER E at(unsigned idx, int64_t uniform_off = 0) const { return base[uniform_off + idx]; }
  SSR_MEMBER E at_or_zero(unsigned idx) const { return ((int64_t)idx < n) ? base[idx] : E{}; }
  // four consecutive elements from (16-byte aligned) element idx
  SSR_MEMBER void at4(unsigned idx, int64_t uniform_off, E* out) const { for (int j = 0; j < 4; ++j) out[j] = base[uniform_off + idx + j]; }
#else
  __amdgpu_buffer_rsrc_t rsrc;
  SSR_MEMBER SsrView(const E* p, int64_t n_elems)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(unsigned)(n_elems * (int64_t)sizeof(E)), 0x00020000)) {}
  SSR_MEMBER E at(unsigned idx, int64_t uniform_off = 0) const {
    const int vo = (int)(idx * (unsigned)sizeof(E)), so = (int)(unsigned)(uniform_off * (int64_t)sizeof(E));
    if constexpr (sizeof(E) == 4) return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, so, 0));
    else if constexpr (sizeof(E) == 8) return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo, so, 0));
    else return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, so, 0));
  }
  // element idx, or 0 where idx >= n_elems (the hardware's range check; idx may be a negative int cast to unsigned as
  // long as |idx| * sizeof(E) < 2^31)
  SSR_MEMBER E at_or_zero(unsigned idx) const { return at(idx); }
  // four consecutive 4-byte elements from (16-byte aligned) element idx: one buffer_load_dwordx4
  SSR_MEMBER void at4(unsigned idx, int64_t uniform_off, E* out) const {
    static_assert(sizeof(E) == 4, "dword elements");
    const int vo = (int)(idx * 4u), so = (int)(unsigned)(uniform_off * 4);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, so, 0);
    out[0] = __builtin_bit_cast(E, v.x); out[1] = __builtin_bit_cast(E, v.y); out[2] = __builtin_bit_cast(E, v.z); out[3] = __builtin_bit_cast(E, v.w);
  }
#endif
};

// A read-WRITE device array with predicated, range-checked element access: ld returns 0 and st is dropped when `ok` is
// false or idx >= n_elems.  Device: raw buffer resource; a false predicate selects an out-of-range offset, so the
// hardware's bounds check does the masking (no branch, no exec-mask change).  idx may be a NEGATIVE int cast to unsigned
// (it stays out of range after the scaling by sizeof(E) as long as |idx| * sizeof(E) < 2^31).
template <typename E> struct SsrRwView {
#ifdef SSR_HOST_EMU
  E* base; int64_t n;
  SSR_MEMBER SsrRwView(E* p, int64_t n_elems) : base(p), n(n_elems) {}
  SSR_MEMBER E ld(unsigned idx, bool ok) const { return (ok && (int64_t)idx < n) ? base[idx] : (E)0; }
  SSR_MEMBER void st(unsigned idx, E v, bool ok) const { if (ok && (int64_t)idx < n) base[idx] = v; }
  // the same by BYTE offset; any negative offset is out of range (callers OR an all-ones mask into it to switch a lane off)
  SSR_MEMBER E ld_raw(int off) const { return ld((unsigned)off / (unsigned)sizeof(E), off >= 0); }
  SSR_MEMBER void st_raw(int off, E v) const { st((unsigned)off / (unsigned)sizeof(E), v, off >= 0); }
  SSR_MEMBER void st_raw_nt(int off, E v) const { st_raw(off, v); }
  SSR_MEMBER void st_raw2(int off, E a, E b) const { st_raw(off, a); st_raw(off + (int)sizeof(E), b); }
  SSR_MEMBER void st_raw4(int off, E a, E b, E c, E d) const { st_raw2(off, a, b); st_raw2(off + 2 * (int)sizeof(E), c, d); }
#else
  __amdgpu_buffer_rsrc_t rsrc;
  SSR_MEMBER SsrRwView(E* p, int64_t n_elems)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(unsigned)(n_elems * (int64_t)sizeof(E)), 0x00020000)) {}
  SSR_MEMBER E ld(unsigned idx, bool ok) const {
    const int vo = ok ? (int)(idx * (unsigned)sizeof(E)) : -1;
    if constexpr (sizeof(E) == 4) return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, 0, 0));
    else return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo, 0, 0));
  }
  SSR_MEMBER void st(unsigned idx, E v, bool ok) const {
    const int vo = ok ? (int)(idx * (unsigned)sizeof(E)) : -1;
    if constexpr (sizeof(E) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, vo, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, v), rsrc, vo, 0, 0);
  }
  SSR_MEMBER E ld_raw(int off) const {
    if constexpr (sizeof(E) == 4) return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
    else return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
  }
  SSR_MEMBER void st_raw(int off, E v) const {
    if constexpr (sizeof(E) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, v), rsrc, off, 0, 0);
  }
  // two / four consecutive 4-byte elements with one buffer_store_dwordx2 / x4 (off: 8- / 16-byte aligned)
  SSR_MEMBER void st_raw2(int off, E a, E b) const {
    static_assert(sizeof(E) == 4, "dword elements");
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 v; v.x = __builtin_bit_cast(unsigned, a); v.y = __builtin_bit_cast(unsigned, b);
    __builtin_amdgcn_raw_buffer_store_b64(v, rsrc, off, 0, 0);
  }
  SSR_MEMBER void st_raw4(int off, E a, E b, E c, E d) const {
    static_assert(sizeof(E) == 4, "dword elements");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 v; v.x = __builtin_bit_cast(unsigned, a); v.y = __builtin_bit_cast(unsigned, b); v.z = __builtin_bit_cast(unsigned, c); v.w = __builtin_bit_cast(unsigned, d);
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 0);
  }
  // streaming store (nt): written once, not read again by this kernel - do not displace what the caches are kept for
  SSR_MEMBER void st_raw_nt(int off, E v) const {
    if constexpr (sizeof(E) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, off, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, v), rsrc, off, 0, 2);
  }
#endif
};

// ------------------------------------------------------------------------------------------------
// complex helpers
template <typename T> struct cx { T x, y; };
template <typename T> SSR_DEV cx<T> cadd(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> SSR_DEV cx<T> csub(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> SSR_DEV cx<T> cmul(cx<T> a, cx<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename T> SSR_DEV cx<T> cmul_negi(cx<T> a) { return {a.y, -a.x}; }   // a * (-i)

// LDS index padding for the FFT arrays: one spare element per 2^SSR_PAD_SHIFT.  An ADDITIVE pad keeps
// pad(j + q*stride) = pad(j) + q*stride' for the power-of-two strides of the passes, so the compiler folds
// the eight per-register offsets into the ds_read/ds_write immediate field (one address VALU op per
// pass).  An XOR swizzle with fewer modelled bank conflicts was measured 28 % SLOWER on MI355X because it
// needs per-access address arithmetic on a VALU-bound kernel (profiles/r01_notes.md).
#ifndef SSR_PAD_SHIFT
#define SSR_PAD_SHIFT 4
#endif
SSR_DEV int ssr_pad(int i) { return i + (i >> SSR_PAD_SHIFT); }
SSR_HD constexpr int ssr_padded_len(int n) { return n + (n >> SSR_PAD_SHIFT) + 1; }

// centred-STFT reflect padding of sample index s into [0, n): numpy.pad(mode="reflect"), i.e. reflection about both
// ends repeated with period 2 (n - 1) when the pad is longer than the signal (n = 1: constant).  The modulo is only
// reached by samples beyond the end.
SSR_DEV int ssr_reflect(int s, int n) {
  if (s < 0) s = -s;
  if (s >= n) {
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    s %= period;
    if (s >= n) s = period - s;
  }
  return s;
}

// Block-wide sum of NV doubles per thread, one value at a time (3 phases each).
// sc0: NT doubles, sc1: 16 doubles, result: out[0..NV) (LDS), valid for every thread after the call.
#define SSR_BLOCK_SUM(blk, regs, NT_, NV_, sc0, sc1, out, GETTER)                            \
  for (int q_ = 0; q_ < (NV_); ++q_) {                                                       \
    SSR_PHASE(blk, regs, (sc0)[tid] = GETTER(q_));                                           \
    SSR_PHASE(blk, regs, if (tid < 16) {                                                     \
      double s_ = 0.0;                                                                       \
      for (int i_ = tid; i_ < (NT_); i_ += 16) s_ += (sc0)[i_];                              \
      (sc1)[tid] = s_;                                                                       \
    });                                                                                      \
    SSR_PHASE(blk, regs, if (tid == 0) {                                                     \
      double s_ = 0.0;                                                                       \
      for (int i_ = 0; i_ < 16; ++i_) s_ += (sc1)[i_];                                       \
      (out)[q_] = s_;                                                                        \
    });                                                                                      \
  }
