// Kernel bodies K1-K4: framed-window STFT of TWO real sequences per complex FFT, magnitude /
// complex output, and the fused LSD + SISpec / log-SISpec accumulation.
//
// Reference semantics reproduced (ssr_eval/metrics.py:26-30 -> librosa.stft; ssr_eval/dsp.py:72-81 ->
// torchlibrosa STFT): centred frames with reflect padding of n_fft//2, periodic Hann window,
// frame stride `hop`, bins 0..n_fft/2; float64 transform rounded once to float32 (precision f64) or
// a float32 transform (precision f32).  |X| is taken as hypotf on the rounded float32 (re, im), the
// way numpy.abs acts on complex64.
//
// Two-for-one packing: z = x_a + i*x_b is transformed once; X_a[k] = (Z[k] + conj(Z[n-k]))/2,
// X_b[k] = (Z[k] - conj(Z[n-k]))/(2i).
//   mode PAIR   : x_a = est frame t, x_b = target frame t        (metrics path)
//   mode SINGLE : x_a = frame 2g,    x_b = frame 2g+1 of one signal (wav_to_spectrogram, FDomainHelper)
//
// Three transform engines share the epilogue:
//   direct    : n_fft = 2^LOGN, FFT of length n_fft.
//   bluestein : any n_fft; chirp-z through two FFTs of length M = 2^LOGM >= 2*n_fft-1
//               (743 / 1114 / 1486 for AudioMetrics at 16 / 24 / 32 kHz).
//   radix-3   : n_fft = 3q (2229 = 3*743 for AudioMetrics(48000)); ssr_stft_r3.h.
#pragma once
#include "ssr_fft.h"
// unsigned index: lets the compiler address tables as scalar base + 32-bit lane offset
#define SSR_UIDX(...) ((unsigned)(__VA_ARGS__))

enum { SSR_MODE_PAIR = 0, SSR_MODE_SINGLE = 1 };

// points per thread of the FFT engine for a given transform length: 16 for the 8192-point Bluestein
// transforms (512 threads x up to 256 VGPRs instead of 1024 threads x 128 VGPRs with spills), else 8.
#ifndef SSR_PPT_2048
#define SSR_PPT_2048 8   /* developer A/B knob for the direct 2048-point engine */
#endif
SSR_HD constexpr int ssr_stft_ppt(int logn, bool bluestein) {
  return logn >= 13 ? 16 : ((logn == 11 && !bluestein) ? SSR_PPT_2048 : 8);
}
enum { SSR_OUT_NONE = 0, SSR_OUT_MAG = 1, SSR_OUT_COMPLEX = 2 };
enum { SSR_M_LSD = 1, SSR_M_LOG_SISPEC = 2, SSR_M_SISPEC = 4, SSR_M_SSIM = 8 };

#define SSR_NPART 8  // doubles per (item, chunk) partial record
// partial record layout: [0] sum over frames of sqrt(mean_f d^2)   (LSD numerator)
//                        [1] Sdd [2] Stt [3] Sdt            (raw magnitudes; d = est - target, elementwise, exact in float64)
//                        [4] Sdd [5] Stt [6] Sdt            (the same on log10(mag + 1e-12))
//                        [7] unused
// The SISpec sums are kept on the DIFFERENCE d = e - t rather than on e: the noise energy ||e - alpha t||^2 then comes out
// as Sdd + 2 (1 - alpha) Sdt + (1 - alpha)^2 Stt (ssr_sispec_from_sums), which stays accurate when est is within 1e-6 of
// target - from (See, Stt, Set) the same quantity is a difference of nearly equal sums, good to ~1e-16 See at best.

template <typename T> struct SsrStftParams {
  const float* a;            // signal buffer A (est, or the only signal in SINGLE mode)
  const double* a64;         // IN64 kernels only: the estimate as float64 samples (a is unused then)
  const double* b64;         // IN64 == 3 kernels only: the target as float64 samples (b is unused then)
  const float* b;            // signal buffer B (target); unused in SINGLE mode
  const int64_t* a_off;      // [n_items] element offset of item i in a
  const int64_t* b_off;      // [n_items] element offset of item i in b
  const int32_t* len;        // [n_items] samples per item
  const int64_t* frame_off;  // [n_items] first output row (frame) of item i
  int mode, out_kind, metric_mask;
  int n_fft, hop, n_bins;
  int units_per_chunk;       // frames (PAIR) or frame pairs (SINGLE) per workgroup
  int n_chunks;              // gridDim.x
  int interleave;            // wave engine (ssr_stft_wave.h): S chunks of a group take every S-th unit of the group's span
                             // (0 / 1: a chunk's units are consecutive); n_chunks is then a multiple of S
  const T* window;           // [n_fft] 0.5 * periodic Hann (direct engine; the 1/2 belongs to the separation)
  const cx<T>* tw;           // [N or M] twiddles
  // bluestein tables (null for the direct engine)
  const cx<T>* wchirp;       // [n_fft]  window[m] * exp(-i*pi*m^2/n_fft)
  const cx<T>* bfilt;        // [M]      FFT_M(exp(+i*pi*m^2/n_fft) wrapped) / M
  const cx<T>* chirp;        // [n_fft]  0.5 * exp(-i*pi*k^2/n_fft)
  float* out_a;              // PAIR: est magnitudes [frames, F]; SINGLE: mag or re
  float* out_b;              // PAIR: target magnitudes;          SINGLE: im (COMPLEX) or unused
  int out_pitch;             // floats between output rows (0: F).  The pair pipeline pads its magnitude rows to a multiple of 4
                             // so that k_ssim reads them with aligned 16-byte loads (ssr_metrics.h, CONTIG)
  double* part;              // [n_items, n_chunks, SSR_NPART] or null
};

// Sample types of the two signals of a pair.  IN64: 0 = both float32, 1 = float64 estimate against a float32
// target, 3 = both float64.  (A float32 estimate against a float64 target is widened on the host and runs as 3.)
// 7 = TWO float64 ESTIMATES in one complex transform (ssr_pair_metrics_multi_est64: no target, no metric terms - the two float64
// magnitudes rounded once into the two image rows; ssr_stft_r3_rot.h).
enum { SSR_IN_F32 = 0, SSR_IN_EST64 = 1, SSR_IN_BOTH64 = 3, SSR_IN_EST64X2 = 7 };
template <bool F64> struct SsrSample { typedef float type; };
template <> struct SsrSample<true> { typedef double type; };

template <typename T, bool SUMS = false, int PPT = 8> struct SsrStftRegs {
  cx<T> v[PPT];              // FFT points
  double sums[SUMS ? 6 : 1]; // SISpec / log-SISpec running sums (kernel variants that do not need them carry none)
  cx<T> twp[3 * (PPT >= 8 ? PPT / 8 : 1)];   // table twiddles of the next pass, requested a phase early (direct engine)
};

// Direct engine: the NEXT frame's samples and window values are requested at the top of the current frame's epilogue,
// ahead of its magnitude stores (vmcnt retires in order on this hardware: a load issued after those stores cannot be
// waited for without also waiting for the stores' acknowledgements).  Only half of the window values are loaded: the
// (half-scaled) periodic Hann window satisfies w[m + N/2] = 1/2 - w[m], and a thread's registers come in such pairs.
template <typename T, bool SUMS, int PPT, typename SA, typename SB> struct SsrStftRegsPf : SsrStftRegs<T, SUMS, PPT> {
  SA pa[PPT];
  SB pb[PPT];
  T pw[PPT / 2];
};
template <bool BLUESTEIN, typename T, bool SUMS, int PPT, typename SA, typename SB> struct SsrStftPickRegs {
  typedef SsrStftRegsPf<T, SUMS, PPT, SA, SB> type;
};
template <typename T, bool SUMS, int PPT, typename SA, typename SB> struct SsrStftPickRegs<true, T, SUMS, PPT, SA, SB> {
  typedef SsrStftRegs<T, SUMS, PPT> type;
};


SSR_DEV int ssr_num_frames_dev(int n, int n_fft, int hop) { return 1 + (n + 2 * (n_fft / 2) - n_fft) / hop; }

// ---- shared epilogue: one bin of the separated spectra ---------------------------------------------
template <typename T> struct SsrBinOut { float ar, ai, br, bi; };

// zk, zn already carry the factor 1/2 (folded into the window / chirp tables; exact in binary FP)
template <typename T> SSR_DEV SsrBinOut<T> ssr_separate(cx<T> zk, cx<T> zn) {
  SsrBinOut<T> o;
  o.ar = (float)(zk.x + zn.x);
  o.ai = (float)(zk.y - zn.y);
  o.br = (float)(zk.y + zn.y);
  o.bi = (float)(zn.x - zk.x);
  return o;
}

// |re + i im| for float32 parts: numpy.abs(complex64) is hypotf (a plain sqrtf(re^2 + im^2) measured slower
// in this kernel, profiles/r01_notes.md).
SSR_DEV float ssr_cabsf(float re, float im) {
#ifdef SSR_FASTABS_ALL   /* developer A/B: the fast magnitude in every engine */
  return sqrtf(fmaf(re, re, im * im));
#else
  return hypotf(re, im);
#endif
}

// The same magnitude without ocml's hypotf (~25 instructions of scaling logic: 1/5 of the wave engine's epilogue): one
// float32 multiply, one fused multiply-add and the 1-ulp hardware square root - within 1.5 ulp of the correctly rounded
// value (numpy's hypotf: within 1).  Squares of parts below ~1e-19 underflow: such bins sit 7 orders of magnitude under
// the 1e-12 guards of every metric, and audio never gets near the 1e19 overflow side.  The spectrogram tests bound the
// difference (<= 2e-7 max|X|).
SSR_DEV float ssr_cabsf_fast(float re, float im) {
#ifdef SSR_HOST_EMU
  return sqrtf(fmaf(re, re, im * im));
#else
  return __builtin_amdgcn_sqrtf(__builtin_fmaf(re, re, im * im));
#endif
}

// float32 helpers of the wave engine's epilogue (FASTM variants).  Both stay within ~1 ulp of the correctly rounded result,
// i.e. inside the rounding noise the reference's own float32 tensor arithmetic has (metrics.py:110, utils.py:43-44), and cost
// a third of the library sequences:
//  * ssr_divf_fast : a / b by the hardware reciprocal (1 ulp) plus one residual correction; the operands here are squares of
//    magnitudes guarded by 1e-12, nowhere near the ranges where div_scale / div_fixup would matter.
//  * ssr_log10f_fast: log10(x) = log2(x) * log10(2) with the constant split in two (the sequence LLVM uses for
//    llvm.log10.f32), without the denormal-input rescaling: every argument carries a +1e-12 guard.
SSR_DEV float ssr_divf_fast(float a, float b) {
#ifdef SSR_HOST_EMU
  return a / b;
#else
  const float r = __builtin_amdgcn_rcpf(b);
  const float q = a * r;
  return __builtin_fmaf(__builtin_fmaf(-q, b, a), r, q);
#endif
}
SSR_DEV float ssr_log10f_fast(float x) {
#ifdef SSR_HOST_EMU
  return log10f(x);
#else
  const float y = __builtin_amdgcn_logf(x);                       // log2(x), 1 ulp
  const float c = 0x1.344134p-2f, cc = 0x1.09f79ep-26f;           // log10(2) = c + cc
  const float r = y * c;
  return r + __builtin_fmaf(y, cc, __builtin_fmaf(y, c, -r));
#endif
}

// The same three sequences on two values at a time: every multiply / add / fused multiply-add is one packed instruction for
// the pair (the square root, reciprocal and logarithm units take one value per instruction).  Component for component the
// arithmetic - and so every bit of the result - is that of the scalar versions.
SSR_DEV f2 ssr_cabsf_fast2(f2 re, f2 im) {
  const f2 s = f2_fma(re, re, im * im);
#ifdef SSR_HOST_EMU
  return f2_make(sqrtf(s.x), sqrtf(s.y));
#else
  return f2_make(__builtin_amdgcn_sqrtf(s.x), __builtin_amdgcn_sqrtf(s.y));
#endif
}
SSR_DEV f2 ssr_divf_fast2(f2 a, f2 b) {
#ifdef SSR_HOST_EMU
  return f2_make(a.x / b.x, a.y / b.y);
#else
  const f2 r = f2_make(__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y));
  const f2 q = a * r;
  return f2_fma(f2_fma(f2_make(-q.x, -q.y), b, a), r, q);
#endif
}
SSR_DEV f2 ssr_log10f_fast2(f2 x) {
#ifdef SSR_HOST_EMU
  return f2_make(log10f(x.x), log10f(x.y));
#else
  const f2 y = f2_make(__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y));
  const f2 c = f2_splat(0x1.344134p-2f), cc = f2_splat(0x1.09f79ep-26f);
  const f2 r = y * c;
  return r + f2_fma(y, cc, f2_fma(y, c, f2_make(-r.x, -r.y)));
#endif
}

// LSD term and SISpec sums for one (est, target) magnitude pair, float32 elementwise arithmetic in
// the order of ssr_eval/metrics.py:110 and ssr_eval/utils.py:43-44; accumulation in float64.
template <bool FASTM = false>
SSR_DEV void ssr_accumulate_metrics(float e, float t, int mask, double* acc) {
  const float EPSF = 1e-12f;
  if (mask & SSR_M_LSD) {
    const float ee = e + EPSF;
    // block engines: IEEE division + the library's log10f (cheaper variants measured no faster there - they are not bound
    // by instruction issue); wave engine (FASTM): the two ~1-ulp sequences above
    const float r = (FASTM ? ssr_divf_fast(t * t, ee * ee) : (t * t) / (ee * ee)) + EPSF;
    const float d = FASTM ? ssr_log10f_fast(r) : log10f(r);
    acc[0] += (double)(d * d);
  }
  if (mask & SSR_M_SISPEC) {
    const double td = (double)t, d = (double)e - td;
    acc[1] += d * d;
    acc[2] += td * td;
    acc[3] += d * td;
  }
  if (mask & SSR_M_LOG_SISPEC) {
    const float le = FASTM ? ssr_log10f_fast(e + EPSF) : log10f(e + EPSF);
    const float lt = FASTM ? ssr_log10f_fast(t + EPSF) : log10f(t + EPSF);
    const double td = (double)lt, d = (double)le - td;
    acc[4] += d * d;
    acc[5] += td * td;
    acc[6] += d * td;
  }
}

// The same terms when the estimate is a float64 signal.  librosa.stft then returns complex128 and the reference's
// est spectrogram is a float64 tensor against a float32 target, so torch's type promotion applies:
//   lsd        : target**2 stays float32, (est + EPS)**2, the quotient, + EPS and log10 are float64;
//   log-sispec : to_log(est) is float64, to_log(target) float32 (promoted when the two meet);
//   sispec     : products in float64.
// Round 6 (SSR_EST64_FAST, default): the two LOGARITHMS of a bin run as the wave engine's float32 sequences on the float64 magnitude
// rounded ONCE - log10 is what made this epilogue as expensive as the six transforms in front of it (ocml's float64 log10 x 2 and a
// float64 division per bin: ~400 instructions against ~60).  What that changes: d = log10(t^2 / (e + EPS)^2 + EPS) by <= 5e-8
// absolute (6e-8 relative on e + EPS, times 2 / ln 10), log10(e + EPS) by one float32 ulp of a value <= 12 in magnitude - both
// random per bin, four orders of magnitude inside the 1e-5 bar and inside the 1e-6 the est64 tests hold against the float64
// oracle; the SISpec products and every accumulation stay float64, the magnitude itself is the float64 one.
#ifndef SSR_EST64_FAST
#define SSR_EST64_FAST 1
#endif
SSR_DEV void ssr_accumulate_metrics(double e, float t, int mask, double* acc) {
  const double EPS = 1e-12;
  const float EPSF = 1e-12f;
#if SSR_EST64_FAST
  const float ee = (float)(e + EPS);
  if (mask & SSR_M_LSD) {
    const float d = ssr_log10f_fast(ssr_divf_fast(t * t, ee * ee) + EPSF);
    acc[0] += (double)d * (double)d;
  }
  if (mask & SSR_M_SISPEC) {
    const double td = (double)t, d = e - td;
    acc[1] += d * d;
    acc[2] += td * td;
    acc[3] += d * td;
  }
  if (mask & SSR_M_LOG_SISPEC) {
    const double le = (double)ssr_log10f_fast(ee), lt = (double)ssr_log10f_fast(t + EPSF), d = le - lt;
    acc[4] += d * d;
    acc[5] += lt * lt;
    acc[6] += d * lt;
  }
  return;
#endif
  if (mask & SSR_M_LSD) {
    const double ee = e + EPS;
    const double d = log10((double)(t * t) / (ee * ee) + EPS);
    acc[0] += d * d;
  }
  if (mask & SSR_M_SISPEC) {
    const double td = (double)t, d = e - td;
    acc[1] += d * d;
    acc[2] += td * td;
    acc[3] += d * td;
  }
  if (mask & SSR_M_LOG_SISPEC) {
    const double le = log10(e + EPS), lt = (double)log10f(t + EPSF), d = le - lt;
    acc[4] += d * d;
    acc[5] += lt * lt;
    acc[6] += d * lt;
  }
}

// numpy.abs(complex128) for the float64-estimate path: a float64 fused multiply-add and square root instead of ocml's hypot (its
// scaling logic guards ranges audio spectra never reach: |Z| < 1e6, and squares of parts below 1e-150 would have to underflow);
// within one float64 ulp of hypot's result.
SSR_DEV double ssr_cabs_d(double re, double im) {
#if SSR_EST64_FAST
  return sqrt(fma(re, re, im * im));
#else
  return hypot(re, im);
#endif
}

// Both signals float64: every tensor of metrics.py:109-121 is float64.
SSR_DEV void ssr_accumulate_metrics(double e, double t, int mask, double* acc) {
  const double EPS = 1e-12;
  if (mask & SSR_M_LSD) {
    const double ee = e + EPS;
    const double d = log10((t * t) / (ee * ee) + EPS);
    acc[0] += d * d;
  }
  if (mask & SSR_M_SISPEC) {
    const double d = e - t;
    acc[1] += d * d;
    acc[2] += t * t;
    acc[3] += d * t;
  }
  if (mask & SSR_M_LOG_SISPEC) {
    const double le = log10(e + EPS), lt = log10(t + EPS), d = le - lt;
    acc[4] += d * d;
    acc[5] += lt * lt;
    acc[6] += d * lt;
  }
}

// One (est, target) bin of a pair: the two magnitudes out, the metric terms accumulated.  zk = Z[k], zn = Z[(n-k) mod n].
// An all-zero frame has an exactly zero spectrum in the reference (separate real FFTs).  In the packed transform
// the other signal leaks into it at round-off level (1e-16 of ITS magnitude), which is not negligible against
// the 1e-12 guards of the metrics - so the outputs of an all-zero frame are forced to exact zeros (a_nz / b_nz
// are block-uniform).
template <typename T, int IN64, bool FASTABS = false>
SSR_DEV void ssr_pair_bin(int mask, double* acc, cx<T> zk, cx<T> zn, bool a_nz, bool b_nz, float& e_out, float& t_out) {
  SsrBinOut<T> o = ssr_separate<T>(zk, zn);
  if (!a_nz) { o.ar = 0.0f; o.ai = 0.0f; }
  if (!b_nz) { o.br = 0.0f; o.bi = 0.0f; }
  if constexpr (IN64 == SSR_IN_BOTH64) {
    const double e = a_nz ? hypot((double)zk.x + (double)zn.x, (double)zk.y - (double)zn.y) : 0.0;
    const double t = b_nz ? hypot((double)zk.y + (double)zn.y, (double)zn.x - (double)zk.x) : 0.0;
    e_out = (float)e; t_out = (float)t;
    ssr_accumulate_metrics(e, t, mask, acc);
  } else if constexpr (IN64 == SSR_IN_EST64) {
    // numpy.abs(complex128) of the unrounded est spectrum; the SSIM image keeps its float32 layout (the
    // rounding moves SSIM by < 2e-7, tests/test_gpu_parity.py)
    const double e = a_nz ? ssr_cabs_d((double)zk.x + (double)zn.x, (double)zk.y - (double)zn.y) : 0.0;
    const float t = ssr_cabsf(o.br, o.bi);
    e_out = (float)e; t_out = t;
    ssr_accumulate_metrics(e, t, mask, acc);
  } else {
    const float e = FASTABS ? ssr_cabsf_fast(o.ar, o.ai) : ssr_cabsf(o.ar, o.ai);
    const float t = FASTABS ? ssr_cabsf_fast(o.br, o.bi) : ssr_cabsf(o.br, o.bi);
    e_out = e; t_out = t;
    ssr_accumulate_metrics<FASTABS>(e, t, mask, acc);
  }
}

// Two bins of a pair whose frames both hold signal (no zero forcing), float32 signals, the wave engine's ~1-ulp sequences:
// the fast path of ssr_stft_wave.h's epilogue.  Bin for bin the values and the order in which they reach the accumulators are
// those of ssr_pair_bin<T, 0, true>(mask, ...) with mask = LSD (SUMS false) or LSD | SISPEC | LOG_SISPEC (SUMS true); the
// float32 arithmetic runs on (bin 0, bin 1) register pairs.  e_out / t_out: the two estimate / target magnitudes.
template <typename T, bool SUMS>
SSR_DEV void ssr_pair_bins2_fast(double* acc, cx<T> zk0, cx<T> zn0, cx<T> zk1, cx<T> zn1, f2& e_out, f2& t_out) {
  const f2 EPS2 = f2_splat(1e-12f);
  const f2 ar = f2_make((float)(zk0.x + zn0.x), (float)(zk1.x + zn1.x)), ai = f2_make((float)(zk0.y - zn0.y), (float)(zk1.y - zn1.y));
  const f2 br = f2_make((float)(zk0.y + zn0.y), (float)(zk1.y + zn1.y)), bi = f2_make((float)(zn0.x - zk0.x), (float)(zn1.x - zk1.x));
  const f2 e = ssr_cabsf_fast2(ar, ai), t = ssr_cabsf_fast2(br, bi);
  e_out = e; t_out = t;
  {
    const f2 ee = e + EPS2;
    const f2 d = ssr_log10f_fast2(ssr_divf_fast2(t * t, ee * ee) + EPS2);
    const f2 dd = d * d;
    acc[0] += (double)dd.x;
    if constexpr (!SUMS) acc[0] += (double)dd.y;
    if constexpr (SUMS) {
      // (bin 0's six sums, then bin 1's LSD term and sums: the order of the scalar path)
      const f2 le = ssr_log10f_fast2(e + EPS2), lt = ssr_log10f_fast2(t + EPS2);
      {
        const double td = (double)t.x, d0 = (double)e.x - td;
        acc[1] += d0 * d0; acc[2] += td * td; acc[3] += d0 * td;
        const double ld = (double)lt.x, l0 = (double)le.x - ld;
        acc[4] += l0 * l0; acc[5] += ld * ld; acc[6] += l0 * ld;
      }
      acc[0] += (double)dd.y;
      {
        const double td = (double)t.y, d0 = (double)e.y - td;
        acc[1] += d0 * d0; acc[2] += td * td; acc[3] += d0 * td;
        const double ld = (double)lt.y, l0 = (double)le.y - ld;
        acc[4] += l0 * l0; acc[5] += ld * ld; acc[6] += l0 * ld;
      }
    }
  }
}

// Emit bin k of the current unit.  out_a_row / out_b_row: block-uniform row base pointers (scalar base + 32-bit lane
// offset addressing).
// PAIR  : row_a0 = est magnitudes row,  row_b0 = target magnitudes row
// SINGLE: row_a0 / row_a1 = rows of frames 2g / 2g+1 in out_a; row_b0 / row_b1 the same rows in out_b
template <typename T, int MODE, int IN64 = 0>
SSR_DEV void ssr_emit_bin(const SsrStftParams<T>& p, double* acc, unsigned k, cx<T> zk, cx<T> zn,
                          float* row_a0, float* row_a1, float* row_b0, float* row_b1, bool b_valid, bool a_nz, bool b_nz) {
  if constexpr (MODE == SSR_MODE_PAIR) {
    float e, t;
    ssr_pair_bin<T, IN64>(p.metric_mask, acc, zk, zn, a_nz, b_nz, e, t);
    if (p.out_kind == SSR_OUT_MAG) {
      row_a0[k] = e;
      row_b0[k] = t;
    }
  } else {
    SsrBinOut<T> o = ssr_separate<T>(zk, zn);
    if (!a_nz) { o.ar = 0.0f; o.ai = 0.0f; }
    if (!b_nz) { o.br = 0.0f; o.bi = 0.0f; }
    if (p.out_kind == SSR_OUT_MAG) {
      row_a0[k] = ssr_cabsf(o.ar, o.ai);
      if (b_valid) row_a1[k] = ssr_cabsf(o.br, o.bi);
    } else if (p.out_kind == SSR_OUT_COMPLEX) {
      row_a0[k] = o.ar;
      row_b0[k] = o.ai;
      if (b_valid) {
        row_a1[k] = o.br;
        row_b1[k] = o.bi;
      }
    }
  }
}

// Direct engine epilogue: F = N/2 + 1 = (PPT/2) * NT + 1 -> PPT/2 full, unrolled rounds (all LDS reads in flight
// together) + the Nyquist bin on thread 0.
template <typename T, int LOGN, int MODE, int PPT, int IN64 = 0>
SSR_DEV void ssr_epilogue_direct(const SsrStftParams<T>& p, double* acc, int tid, const T* re, const T* im,
                                 float* ra0, float* ra1, float* rb0, float* rb1, bool b_ok, bool a_nz, bool b_nz) {
  constexpr int N = 1 << LOGN, NT = N / PPT, RND = PPT / 2;   // F = N/2 + 1 = RND * NT + 1
  cx<T> zk[RND], zn[RND];
#pragma unroll
  for (int i = 0; i < RND; ++i) {
    const int k = tid + i * NT, kn = (N - k) & (N - 1);
    zk[i] = {re[ssr_pad(k)], im[ssr_pad(k)]};
    zn[i] = {re[ssr_pad(kn)], im[ssr_pad(kn)]};
  }
#pragma unroll
  for (int i = 0; i < RND; ++i)
    ssr_emit_bin<T, MODE, IN64>(p, acc, (unsigned)(tid + i * NT), zk[i], zn[i], ra0, ra1, rb0, rb1, b_ok, a_nz, b_nz);
  if (tid == 0) {
    const cx<T> zq = {re[ssr_pad(N / 2)], im[ssr_pad(N / 2)]};
    ssr_emit_bin<T, MODE, IN64>(p, acc, (unsigned)(N / 2), zq, zq, ra0, ra1, rb0, rb1, b_ok, a_nz, b_nz);
  }
}

// PAIR mode, direct engine: the same bins, but the magnitudes stay in registers - the caller stores them after it has
// consumed the prefetched samples of the next frame (see the frame loop).
template <typename T, int LOGN, int PPT, int IN64>
SSR_DEV void ssr_epilogue_direct_pair(int mask, double* acc, int tid, const T* re, const T* im, bool a_nz, bool b_nz,
                                      float* ev, float* tv, float& eq, float& tq) {
  constexpr int N = 1 << LOGN, NT = N / PPT, RND = PPT / 2;
  cx<T> zk[RND], zn[RND];
#pragma unroll
  for (int i = 0; i < RND; ++i) {
    const int k = tid + i * NT, kn = (N - k) & (N - 1);
    zk[i] = {re[ssr_pad(k)], im[ssr_pad(k)]};
    zn[i] = {re[ssr_pad(kn)], im[ssr_pad(kn)]};
  }
#pragma unroll
  for (int i = 0; i < RND; ++i) ssr_pair_bin<T, IN64>(mask, acc, zk[i], zn[i], a_nz, b_nz, ev[i], tv[i]);
  if (tid == 0) {
    const cx<T> zq = {re[ssr_pad(N / 2)], im[ssr_pad(N / 2)]};
    ssr_pair_bin<T, IN64>(mask, acc, zq, zq, a_nz, b_nz, eq, tq);
  }
}

// LDS carve-out (doubles first so every array stays 8-byte aligned)
template <typename T, int LOGN, int PPT = 8> struct SsrStftLds {
  static constexpr int NT = (1 << LOGN) / PPT;
  static constexpr int PN = ssr_padded_len(1 << LOGN);
  static constexpr int NW = (NT + 63) / 64;
  // sc1: per-wave LSD sums of the current frame; wacc: per-wave running SISpec sums [6][NW];
  // res[0]: running sum over frames of the per-frame LSD (thread 0);
  // nz: per-wave "this frame of signal A / B has a non-zero sample" flags [2][16]; the direct engine (<= 8 waves)
  // keeps two sets [2][2][8], written one frame ahead
  static constexpr size_t bytes() { return sizeof(double) * (16 + 6 * 16 + 8 + 16) + sizeof(T) * 2 * PN; }
  double* sc1; double* wacc; double* res; int* nz; T* re; T* im;
  SSR_MEMBER explicit SsrStftLds(char* base) {
    sc1 = reinterpret_cast<double*>(base);
    wacc = sc1 + 16;
    res = wacc + 6 * 16;
    nz = reinterpret_cast<int*>(res + 8);
    re = reinterpret_cast<T*>(res + 8 + 16);
    im = re + PN;
  }
  // block-uniform: does the current frame of signal A (which = 0) / B (which = 1) contain any non-zero sample?
  SSR_MEMBER bool any_nonzero(int which, int par = 0) const {
    int f = 0;
    for (int w = 0; w < NW; ++w) f |= nz[which * 16 + par * 8 + w];
#ifndef SSR_HOST_EMU
    f = __builtin_amdgcn_readfirstlane(f);      // every lane read the same words: make the uniformity explicit (scalar branch)
#endif
    return f != 0;
  }
};

// Direct engine: request unit u's samples (first-pass order) and half of its window values into the prefetch
// registers.  Branch-free, always-valid addresses: a frame that does not exist re-reads the last one (the consumer
// scales it by 0); frames that touch the signal ends go through the reflection index.
template <typename T, int LOGN, int MODE, int PPT, typename SA, typename SB, typename REGS>
SSR_DEV void ssr_stft_prefetch(const SsrStftParams<T>& p, REGS& R, int tid, const SsrView<SA>& va, const SsrView<SB>& vb,
                               const SsrView<T>& vw, int u, int n, int n_frames) {
  using P = SsrFftPlan<LOGN, PPT>;
  constexpr int N = 1 << LOGN, R0 = P::R0;
  const int ta = (MODE == SSR_MODE_PAIR) ? u : 2 * u;
  const int tb = (MODE == SSR_MODE_PAIR) ? u : 2 * u + 1;
  const int ta_c = (ta < n_frames) ? ta : n_frames - 1, tb_c = (tb < n_frames) ? tb : n_frames - 1;
  const int base_a = ta_c * p.hop - N / 2, base_b = tb_c * p.hop - N / 2;
  // block-uniform: both frames lie fully inside the signal -> no reflection arithmetic at all
  const bool interior = base_a >= 0 && base_b >= 0 && base_a + N <= n && base_b + N <= n;
  if (interior) {
    SSR_UNROLL for (int r = 0; r < PPT; ++r) {
      const unsigned m = SSR_UIDX(ssr_fft_first_index<LOGN, PPT>(tid, r));
      R.pa[r] = va.at(m, base_a);
      R.pb[r] = vb.at(m, base_b);
    }
  } else {
    SSR_UNROLL for (int r = 0; r < PPT; ++r) {
      const int m = ssr_fft_first_index<LOGN, PPT>(tid, r);
      R.pa[r] = va.at(SSR_UIDX(ssr_reflect(base_a + m, n)));
      R.pb[r] = vb.at(SSR_UIDX(ssr_reflect(base_b + m, n)));
    }
  }
  // registers (b, q) and (b, q + R0/2) are N/2 samples apart: load the lower one of each pair
  SSR_UNROLL for (int r = 0; r < PPT; ++r)
    if ((r % R0) < R0 / 2) R.pw[(r / R0) * (R0 / 2) + (r % R0)] = vw.at(SSR_UIDX(ssr_fft_first_index<LOGN, PPT>(tid, r)));
}

// Direct engine: silent-frame flags of the PREFETCHED unit (OR of the samples' magnitude bits per signal, reduced per
// wave with a ballot) into flag set `par`.  Register 0 of thread 0 is sample m = 0, whose window weight is exactly 0:
// it never reaches the transform and does not count.  Touching the prefetched registers (samples and window values)
// here is also what places the wait for those loads (s_waitcnt vmcnt) at this point of the program.
template <int PPT, typename REGS>
SSR_DEV void ssr_stft_prefetched_flags(REGS& R, int tid, int* nz, int par) {
  SSR_UNROLL for (int r = 0; r < PPT; ++r) { ssr_touch(R.pa[r]); ssr_touch(R.pb[r]); }
  SSR_UNROLL for (int i = 0; i < PPT / 2; ++i) ssr_touch(R.pw[i]);
  unsigned ora = 0u, orb = 0u;
  SSR_UNROLL for (int r = 1; r < PPT; ++r) { ora |= ssr_mag_bits(R.pa[r]); orb |= ssr_mag_bits(R.pb[r]); }
  ora |= (tid == 0) ? 0u : ssr_mag_bits(R.pa[0]);
  orb |= (tid == 0) ? 0u : ssr_mag_bits(R.pb[0]);
  SSR_WAVE_ANY_STORE(tid, ora != 0u, nz + par * 8);
  SSR_WAVE_ANY_STORE(tid, orb != 0u, nz + 16 + par * 8);
}

// ---------------------------------------------------------------------------------------------------
// The body.  LOGN: FFT length of the engine (n_fft for direct, M for bluestein).
// grid = (n_chunks, n_items); block = 2^LOGN / 8 threads.
template <typename T, int LOGN, bool BLUESTEIN, int MODE, bool SUMS, int PPT, int IN64, typename BLK>
SSR_BODY void ssr_stft_body(const SsrStftParams<T>& p, BLK& blk, int chunk, int item, char* lds_base) {
  using P = SsrFftPlan<LOGN, PPT>;
  constexpr int N = P::N, NT = P::NT, LAST = P::NPASS - 1, NW = (NT + 63) / 64;
  static_assert(IN64 == 0 || MODE == SSR_MODE_PAIR, "float64 signals exist on the pair path only");
  using SA = typename SsrSample<(IN64 & 1) != 0>::type;
  using SB = typename SsrSample<(IN64 & 2) != 0>::type;
  using Regs = typename SsrStftPickRegs<BLUESTEIN, T, SUMS, PPT, SA, SB>::type;
  constexpr bool PF = PPT <= 8;      // early twiddle requests: everywhere but the register-critical 16-point engine
  SsrStftLds<T, LOGN, PPT> L(lds_base);

  const int n = p.len[item];
  const int n_fft = BLUESTEIN ? p.n_fft : N, hop = p.hop, F = n_fft / 2 + 1;
  const int n_frames = ssr_num_frames_dev(n, n_fft, hop);
  const int n_units = (MODE == SSR_MODE_PAIR) ? n_frames : (n_frames + 1) / 2;
  const int u0 = chunk * p.units_per_chunk;
  const int u1 = (u0 + p.units_per_chunk < n_units) ? u0 + p.units_per_chunk : n_units;
  const SA* sa;
  if constexpr (IN64 & 1) sa = p.a64 + p.a_off[item];
  else sa = p.a + p.a_off[item];
  const SB* sb;
  if constexpr (IN64 & 2) sb = p.b64 + p.b_off[item];
  else sb = (MODE == SSR_MODE_PAIR) ? p.b + p.b_off[item] : p.a + p.a_off[item];
  const int64_t row0 = p.frame_off[item];
  double* part = p.part ? p.part + ((int64_t)item * p.n_chunks + chunk) * SSR_NPART : nullptr;
  const bool want_lsd = (MODE == SSR_MODE_PAIR) && (p.metric_mask & SSR_M_LSD);
  const int pad = n_fft / 2;
  const SsrView<SA> va(sa, n);                 // the item's two signals and the window table as the frame loads see them
  const SsrView<SB> vb(sb, n);
  const SsrView<T> vw(p.window, BLUESTEIN ? 0 : N);
  const SsrView<cx<T>> vt(p.tw, N);           // twiddle table exp(-2 pi i k / N), k < N
  const SsrView<cx<T>> vwc(p.wchirp, BLUESTEIN ? n_fft : 0), vbf(p.bfilt, BLUESTEIN ? N : 0), vch(p.chirp, BLUESTEIN ? n_fft : 0);

  SSR_REGS(Regs, regs, blk);
  SSR_PHASE(blk, regs, {
    for (int i = tid; i < 6 * 16; i += NT) L.wacc[i] = 0.0;   // NT may be as small as 32
    if (tid == 0) L.res[0] = 0.0;
    for (int q = 0; q < (SUMS ? 6 : 1); ++q) R.sums[q] = 0.0;
    if constexpr (!BLUESTEIN)
      if (u0 < u1) {
        ssr_stft_prefetch<T, LOGN, MODE, PPT>(p, R, tid, va, vb, vw, u0, n, n_frames);
        ssr_stft_prefetched_flags<PPT>(R, tid, L.nz, 0);
      }
  });

  BLK blk0 = blk;
  for (int u = u0; u < u1; ++u) {
#if defined(SSR_LAUNDER_ALL)
    blk = blk0; ssr_launder(blk);
#else
    if constexpr (PPT > 8) { blk = blk0; ssr_launder(blk); }   // see ssr_launder: keeps addresses out of scratch
#endif
    const int ta = (MODE == SSR_MODE_PAIR) ? u : 2 * u;
    const int tb = (MODE == SSR_MODE_PAIR) ? u : 2 * u + 1;
    const bool a_ok = ta < n_frames, b_ok = tb < n_frames;       // a missing frame re-reads the last one, scaled by 0
    const int ta_c = a_ok ? ta : n_frames - 1, tb_c = b_ok ? tb : n_frames - 1;
    const int base_a = ta_c * hop - pad, base_b = tb_c * hop - pad;
    // block-uniform: both frames lie fully inside the signal -> no reflection arithmetic at all
    const bool interior = base_a >= 0 && base_b >= 0 && base_a + n_fft <= n && base_b + n_fft <= n;

    // ---- phase 1: HBM -> registers (first-pass order), window, pass 0, store.
    // All 24 loads of a thread (8 samples of each signal + 8 window values) are issued back to back with
    // branch-free, always-valid addresses, so the frame pays ONE memory latency instead of 24 dependent
    // ones.  Thread 0 also closes the PREVIOUS frame's LSD (per-wave sums left in sc1 by its epilogue).
    SSR_PHASE(blk, regs, {
      if constexpr (BLUESTEIN) {
        // Four points at a time (sample pair + window*chirp value each), fenced, so that at 16 points per
        // thread the address and table registers of one group die before the next group is issued.
        // Groups whose indices all lie beyond n_fft (at least the upper half of M) are skipped outright.
        bool nza = false, nzb = false;
        SSR_UNROLL for (int r0 = 0; r0 < PPT; r0 += 4) {
          const int m_lo = ssr_fft_first_index<LOGN, PPT>(0, r0);   // smallest index any thread's group can hold
          bool any = false;
          SSR_UNROLL for (int r = 0; r < 4; ++r) any = any || (ssr_fft_first_index<LOGN, PPT>(0, r0 + r) < n_fft);
          (void)m_lo;
          if (any) {
            SA fa[4];
            SB fb[4];
            cx<T> wc[4];
            SSR_UNROLL for (int r = 0; r < 4; ++r) {
              const int m = ssr_fft_first_index<LOGN, PPT>(tid, r0 + r);
              const int mc = (m < n_fft) ? m : n_fft - 1;
              const int ia = interior ? base_a + mc : ssr_reflect(base_a + mc, n);
              const int ib = interior ? base_b + mc : ssr_reflect(base_b + mc, n);
              fa[r] = va.at(SSR_UIDX(ia));
              fb[r] = vb.at(SSR_UIDX(ib));
              wc[r] = vwc.at(SSR_UIDX(mc));
            }
            SSR_UNROLL for (int r = 0; r < 4; ++r) {
              const int m = ssr_fft_first_index<LOGN, PPT>(tid, r0 + r);
              const cx<T> z = cmul(cx<T>{a_ok ? (T)fa[r] : (T)0, b_ok ? (T)fb[r] : (T)0}, wc[r]);
              R.v[r0 + r] = (m < n_fft) ? z : cx<T>{(T)0, (T)0};
              nza = nza || (m < n_fft && m != 0 && fa[r] != 0);      // sample m = 0: window weight exactly 0 (periodic Hann)
              nzb = nzb || (m < n_fft && m != 0 && fb[r] != 0);
            }
          } else {
            SSR_UNROLL for (int r = 0; r < 4; ++r) R.v[r0 + r] = cx<T>{(T)0, (T)0};
          }
          SSR_SCHED_FENCE();
        }
        SSR_WAVE_ANY_STORE(tid, nza, L.nz);
        SSR_WAVE_ANY_STORE(tid, nzb, L.nz + 16);
      } else {
        // samples and window values were requested one frame ahead (ssr_stft_prefetch)
        constexpr int R0 = P::R0;
        SSR_UNROLL for (int r = 0; r < PPT; ++r) {
          const T wl = R.pw[(r / R0) * (R0 / 2) + (r % R0) % (R0 / 2)];
          const T w = ((r % R0) < R0 / 2) ? wl : (T)0.5 - wl;           // w[m + N/2] = 1/2 - w[m]
          R.v[r] = {a_ok ? (T)R.pa[r] * w : (T)0, b_ok ? (T)R.pb[r] * w : (T)0};
        }
      }
      ssr_fft_compute<T, LOGN, 0, PPT>(tid, R.v, vt);
      ssr_fft_store<T, LOGN, 0, PPT>(tid, L.re, L.im, R.v);
      if constexpr (PF) ssr_fft_load_tw<T, LOGN, 1, PPT>(tid, vt, R.twp);   // pass 1's twiddles, in flight across the barrier
      if (want_lsd && u > u0 && tid == 0) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += L.sc1[w];
        L.res[0] += sqrt(s / (double)F);
      }
    });
    // remaining forward passes; last pass stays in registers
    ssr_fft_mid_passes<T, LOGN, 1, PPT, PF>(blk, regs, L.re, L.im, vt);

    if constexpr (BLUESTEIN) {
      // forward spectrum * filter, stored as the INPUT of the inverse transform.  The inverse is the
      // forward engine on exchanged (im, re) arrays.
      SSR_PHASE(blk, regs, {
        SSR_UNROLL for (int r0 = 0; r0 < PPT; r0 += 4) {
          SSR_UNROLL for (int r = r0; r < r0 + 4; ++r) {
            const int k = ssr_fft_out_index<LOGN, LAST, PPT>(tid, r);
            const cx<T> y = cmul(R.v[r], vbf.at(SSR_UIDX(k)));
            L.re[ssr_pad(k)] = y.x;
            L.im[ssr_pad(k)] = y.y;
          }
          SSR_SCHED_FENCE();
        }
      });
      SSR_PHASE(blk, regs, ssr_fft_load<T, LOGN, 0, PPT>(tid, L.im, L.re, R.v);
                ssr_fft_compute<T, LOGN, 0, PPT>(tid, R.v, vt));
      SSR_PHASE(blk, regs, ssr_fft_store<T, LOGN, 0, PPT>(tid, L.im, L.re, R.v);
                if constexpr (PF) ssr_fft_load_tw<T, LOGN, 1, PPT>(tid, vt, R.twp));
      ssr_fft_mid_passes<T, LOGN, 1, PPT, PF>(blk, regs, L.im, L.re, vt);
      // registers hold swap(IFFT*M): true real part = .y, true imaginary part = .x
      SSR_PHASE(blk, regs, {
        SSR_UNROLL for (int r = 0; r < PPT; ++r) {
          const int k = ssr_fft_out_index<LOGN, LAST, PPT>(tid, r);
          if (k < n_fft) {
            const cx<T> zk = cmul(cx<T>{R.v[r].y, R.v[r].x}, vch.at(SSR_UIDX(k)));
            L.re[ssr_pad(k)] = zk.x;
            L.im[ssr_pad(k)] = zk.y;
          }
        }
      });
    } else {
      SSR_PHASE(blk, regs, ssr_fft_store<T, LOGN, LAST, PPT>(tid, L.re, L.im, R.v));
    }

    // ---- epilogue: separate the two spectra, emit, accumulate; leave per-wave LSD sums in sc1.
    const int64_t OP = p.out_pitch ? p.out_pitch : F;      // floats between output rows
    float* ra0 = p.out_a ? p.out_a + (row0 + ta) * OP : nullptr;   // block-uniform row pointers
    float* ra1 = p.out_a ? p.out_a + (row0 + tb) * OP : nullptr;
    float* rb0 = p.out_b ? p.out_b + (row0 + ta) * OP : nullptr;
    float* rb1 = p.out_b ? p.out_b + (row0 + tb) * OP : nullptr;
    if (MODE == SSR_MODE_PAIR) { ra1 = ra0; rb1 = rb0; }
    SSR_PHASE(blk, regs, {
      if constexpr (!BLUESTEIN)
        if (u + 1 < u1) ssr_stft_prefetch<T, LOGN, MODE, PPT>(p, R, tid, va, vb, vw, u + 1, n, n_frames);
      double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const int par = BLUESTEIN ? 0 : ((u - u0) & 1);               // flag set of this unit (direct engine: two sets)
      const bool a_nz = L.any_nonzero(0, par), b_nz = L.any_nonzero(1, par);
      if constexpr (!BLUESTEIN && MODE == SSR_MODE_PAIR) {
        // The magnitudes stay in registers until the next unit's prefetched samples have been consumed (their flags):
        // vmcnt retires in order, so loads that are waited for BEFORE this unit's stores are issued never wait on the
        // stores' acknowledgements - and the next unit's first phase finds its samples already there.
        constexpr int RND = PPT / 2;
        float ev[RND], tv[RND], eq = 0.0f, tq = 0.0f;
        // block-uniform branch: the common case (no silent frame) carries no zero-forcing selects at all
        if (a_nz && b_nz) ssr_epilogue_direct_pair<T, LOGN, PPT, IN64>(p.metric_mask, acc, tid, L.re, L.im, true, true, ev, tv, eq, tq);
        else ssr_epilogue_direct_pair<T, LOGN, PPT, IN64>(p.metric_mask, acc, tid, L.re, L.im, a_nz, b_nz, ev, tv, eq, tq);
        if (u + 1 < u1) ssr_stft_prefetched_flags<PPT>(R, tid, L.nz, par ^ 1);
        SSR_SCHED_BARRIER();
        if (p.out_kind == SSR_OUT_MAG) {
          SSR_UNROLL for (int i = 0; i < RND; ++i) {
            ra0[SSR_UIDX(tid + i * NT)] = ev[i];
            rb0[SSR_UIDX(tid + i * NT)] = tv[i];
          }
          if (tid == 0) {
            ra0[N / 2] = eq;
            rb0[N / 2] = tq;
          }
        }
      } else if constexpr (!BLUESTEIN) {
        if (a_nz && b_nz) ssr_epilogue_direct<T, LOGN, MODE, PPT, IN64>(p, acc, tid, L.re, L.im, ra0, ra1, rb0, rb1, b_ok, true, true);
        else ssr_epilogue_direct<T, LOGN, MODE, PPT, IN64>(p, acc, tid, L.re, L.im, ra0, ra1, rb0, rb1, b_ok, a_nz, b_nz);
        if (u + 1 < u1) ssr_stft_prefetched_flags<PPT>(R, tid, L.nz, par ^ 1);
      } else {
        for (int k = tid; k < F; k += NT) {
          const int kn = (k == 0) ? 0 : n_fft - k;
          const cx<T> zk = {L.re[ssr_pad(k)], L.im[ssr_pad(k)]};
          const cx<T> zn = {L.re[ssr_pad(kn)], L.im[ssr_pad(kn)]};
          ssr_emit_bin<T, MODE, IN64>(p, acc, (unsigned)k, zk, zn, ra0, ra1, rb0, rb1, b_ok, a_nz, b_nz);
        }
      }
      if (want_lsd) SSR_WAVE_SUM_STORE(tid, NT, acc[0], L.sc1);
      if constexpr (SUMS)
        for (int q = 0; q < 6; ++q) R.sums[q] += acc[1 + q];
    });
  }

  if (part == nullptr) return;
  // ---- chunk tail: last frame's LSD, then block-sum the SISpec accumulators.
  if constexpr (SUMS) {
    SSR_PHASE(blk, regs, for (int q = 0; q < 6; ++q) SSR_WAVE_SUM_ADD(tid, NT, R.sums[q], L.wacc + q * 16));
  }
  SSR_PHASE(blk, regs, if (tid == 0) {
    double lsd = L.res[0];
    if (want_lsd && u1 > u0) {
      double s = 0.0;
      for (int w = 0; w < NW; ++w) s += L.sc1[w];
      lsd += sqrt(s / (double)F);
    }
    part[0] = lsd;
    for (int q = 0; q < 6; ++q) {
      double s = 0.0;
      for (int w = 0; w < NW; ++w) s += L.wacc[q * 16 + w];
      part[1 + q] = s;
    }
    part[7] = 0.0;
  });
}
