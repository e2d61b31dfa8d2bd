// Host-side constant tables for an STFT plan (window, twiddles, Bluestein chirps), computed in
// long double and rounded once to the working type.  Pure host C++ (used by the HIP library and by the
// emulation harness).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "ssr_block.h"

static const long double SSR_PI_L = 3.14159265358979323846264338327950288L;

inline bool ssr_is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
inline int ssr_ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

// engine choice for an n_fft:
//  * direct 2^LOGN FFT when n_fft is a power of two in [256, 4096];
//  * otherwise Bluestein with M = 2^LOGM >= max(256, 2*n_fft - 1), M <= 8192;
//  * radix-3 x Bluestein when n_fft = 3 q and plain Bluestein would need M = 8192 while the three length-q
//    sub-transforms fit M = 2048 (n_fft = 2229 = 3 * 743, i.e. AudioMetrics(48000)): 6 x FFT-2048 in a 35 KB LDS
//    buffer (2 workgroups per CU) instead of 2 x FFT-8192 in 139 KB (1 workgroup per CU).
struct SsrEngine { bool ok; bool bluestein; int logn; int radix; int q; int m; };   // m: transform length when it is not 2^logn (1536), else 0
inline SsrEngine ssr_pick_engine(int n_fft) {
  SsrEngine e{false, false, 0, 1, 0, 0};
  if (n_fft < 2) return e;
  e.q = n_fft;
  if (ssr_is_pow2(n_fft) && n_fft >= 256 && n_fft <= 4096) { e.ok = true; e.logn = ssr_ilog2(n_fft); return e; }
  int m = 256;
  while (m < 2 * n_fft - 1) m <<= 1;
#ifndef SSR_NO_RADIX3
  if (n_fft % 3 == 0 && m >= 8192) {
    const int q = n_fft / 3;
    int mq = 256;
    while (mq < 2 * q - 1) mq <<= 1;
    if (mq <= 2048) { e.ok = true; e.bluestein = true; e.logn = ssr_ilog2(mq); e.radix = 3; e.q = q; return e; }
  }
#endif
  if (m > 8192) return e;
  e.ok = true; e.bluestein = true; e.logn = ssr_ilog2(m);
  return e;
}

// The wave engine's variant for float32 pairs (ssr_stft_rn_wave.h): n_fft = R q over M = 2048 on R autonomous waves.
//  * R = 3 where the block engine already splits by three (2229);
//  * R = 1 where plain Bluestein runs at M = 2048 (513 <= n_fft <= 1024, not a power of two: 743);
//  * R = 2 for even n_fft whose plain Bluestein length is 4096 (1026 <= n_fft <= 2048: 1114, 1486) - its tables differ from
//    the block engine's (chirp of length q = n_fft / 2, filter spectrum of 2048 points) and are built separately.
inline SsrEngine ssr_pick_wave_engine(int n_fft) {
  const SsrEngine e = ssr_pick_engine(n_fft);
  SsrEngine none{false, false, 0, 1, 0, 0};
  if (!e.ok || !e.bluestein) return none;
  SsrEngine w = none;
  if (e.radix == 3) w = (e.logn == 11 && e.q <= 768) ? e : none;
  else if (e.logn == 11) w = e;                                            // R = 1, q = n_fft
  else if (e.logn == 12 && n_fft % 2 == 0 && n_fft / 2 <= 1024) w = SsrEngine{true, true, 11, 2, n_fft / 2, 0};
#ifndef SSR_NO_M1536
  // The chirp-z of a sub-sequence of q <= 768 samples only needs M >= 2 q - 1 = 1535: M = 1536 = 24 x 64 (24 points per lane, an
  // in-register radix-24 first pass; ssr_fft24.h) does a quarter less arithmetic than M = 2048.  Every AudioMetrics(rate) size
  // qualifies (q = 743 or 557).
  if (w.ok && w.q <= 768) w.m = 1536;
#endif
  return w;
}

constexpr int SSR_WAVE_N = 2048, SSR_WAVE_TWP = 7 * 32 + 12 * 64;   // wave engines: transform length, lane-ordered twiddle copies
constexpr int SSR_WAVE24_N = 1536, SSR_WAVE24_TWP = 2 * 9 * 64;     // the 24-points-per-lane variant (ssr_fft24.h)

template <typename T> struct SsrTables {
  SsrEngine eng;
  int n_fft;
  std::vector<T> window;                        // [n_fft]  periodic Hann (inverse STFT / OLA)
  std::vector<T> window_h;                      // [n_fft]  0.5 * Hann: the 1/2 of the two-for-one separation, pre-applied (exact)
  std::vector<cx<T>> tw, wchirp, bfilt, chirp;  // [N] | [n_fft] | [N] | [n_fft] (chirp also carries the 1/2)
};

inline void ssr_host_fft_ld(std::vector<long double>& re, std::vector<long double>& im) {
  const int n = (int)re.size();
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  for (int len = 2; len <= n; len <<= 1) {
    for (int i = 0; i < n; i += len) {
      for (int k = 0; k < len / 2; ++k) {
        const long double ang = -2.0L * SSR_PI_L * k / len;
        const long double wr = cosl(ang), wi = sinl(ang);
        const int a = i + k, b = i + k + len / 2;
        const long double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
        re[b] = re[a] - xr; im[b] = im[a] - xi;
        re[a] += xr; im[a] += xi;
      }
    }
  }
}

// plain O(n^2) DFT in long double for lengths that are not a power of two (1536: once per plan)
inline void ssr_host_dft_ld(std::vector<long double>& re, std::vector<long double>& im) {
  const int n = (int)re.size();
  std::vector<long double> c(n), s(n), orr(n), oi(n);
  for (int i = 0; i < n; ++i) { const long double a = -2.0L * SSR_PI_L * i / n; c[i] = cosl(a); s[i] = sinl(a); }
  for (int k = 0; k < n; ++k) {
    long double ar = 0.0L, ai = 0.0L;
    for (int m = 0; m < n; ++m) {
      const int j = (int)(((int64_t)k * m) % n);
      ar += re[m] * c[j] - im[m] * s[j];
      ai += re[m] * s[j] + im[m] * c[j];
    }
    orr[k] = ar; oi[k] = ai;
  }
  re = orr; im = oi;
}

template <typename T> bool ssr_build_tables_for(int n_fft, SsrEngine eng, SsrTables<T>& t);
template <typename T> bool ssr_build_tables(int n_fft, SsrTables<T>& t) { return ssr_build_tables_for<T>(n_fft, ssr_pick_engine(n_fft), t); }

template <typename T> bool ssr_build_tables_for(int n_fft, SsrEngine eng, SsrTables<T>& t) {
  t.eng = eng;
  t.n_fft = n_fft;
  if (!t.eng.ok) return false;
  const int N = t.eng.m ? t.eng.m : (1 << t.eng.logn);
  t.window.resize(n_fft);
  std::vector<long double> w(n_fft);
  for (int m = 0; m < n_fft; ++m) {
    w[m] = 0.5L - 0.5L * cosl(2.0L * SSR_PI_L * m / n_fft);  // periodic Hann (fftbins=True)
    t.window[m] = (T)w[m];
  }
  t.window_h.resize(n_fft);
  for (int m = 0; m < n_fft; ++m) t.window_h[m] = (T)0.5 * t.window[m];
  t.tw.resize(N);
  for (int i = 0; i < N; ++i) {
    const long double ang = -2.0L * SSR_PI_L * i / N;
    t.tw[i] = {(T)cosl(ang), (T)sinl(ang)};
  }
  if (N == SSR_WAVE24_N) {
    // 24 points per lane (ssr_fft24.h): butterfly b < 3 of lane l is j = l + 64 b = 24 t_low + q; lane-ordered copies
    //   [N + 64 (3 b + k) + l]        = tw[8 q 2^k mod N]        (pass 1: W_192^(q 2^k)),
    //   [N + 576 + 64 (3 b + k) + l]  = tw[j 2^k]                (pass 2: w^j, w^2j, w^4j)
    t.tw.resize(N + SSR_WAVE24_TWP);
    for (int b = 0; b < 3; ++b)
      for (int k = 0; k < 3; ++k)
        for (int l = 0; l < 64; ++l) {
          const int j = l + 64 * b, q = j % 24;
          t.tw[N + 64 * (3 * b + k) + l] = t.tw[(8 * q << k) % N];
          t.tw[N + 576 + 64 * (3 * b + k) + l] = t.tw[(j << k) % N];
        }
  }
  if (N == SSR_WAVE_N) {
    // Lane-ordered copies of the twiddles the wave engines read (ssr_stft_wave.h: SSR_W_LOAD_TW1 / TW2), behind the table:
    //   [N + 32 (q - 1) + l]          = tw[8 l q],          q = 1..7, l < 32   (pass 1: the lane's seven)
    //   [N + 224 + 64 (3 b + k) + l]  = tw[(l + 64 b) 2^k], b < 4, k < 3, l < 64 (pass 2: w^j, w^2j, w^4j of butterfly b)
    // A wave's load of one of them is 512 B / 1 KB contiguous instead of 32 / 64 lines 128..512 B apart.
    t.tw.resize(N + SSR_WAVE_TWP);
    for (int q = 1; q < 8; ++q)
      for (int l = 0; l < 32; ++l) t.tw[N + 32 * (q - 1) + l] = t.tw[8 * l * q];
    for (int b = 0; b < 4; ++b)
      for (int k = 0; k < 3; ++k)
        for (int l = 0; l < 64; ++l) t.tw[N + 224 + 64 * (3 * b + k) + l] = t.tw[(l + 64 * b) << k];
  }
  if (t.eng.bluestein) {
    // inner (Bluestein) length q: n_fft itself, or n_fft / R under a radix-R outer step
    const int q = t.eng.q, R = t.eng.radix;
    std::vector<long double> cr(q), ci(q);
    for (int k = 0; k < q; ++k) {
      const int64_t kk = ((int64_t)k * k) % (2 * (int64_t)q);  // exact phase reduction
      const long double ang = -SSR_PI_L * (long double)kk / q;
      cr[k] = cosl(ang); ci[k] = sinl(ang);
    }
    // chirp[r*q + k]  = 0.5 * exp(-i*pi*k^2/q) * exp(-2*pi*i * r*k / n_fft)   (post-multiplier incl. the outer twiddle)
    // wchirp[r*q + m] = window[R*m + r] * exp(-i*pi*m^2/q)                     (pre-multiplier on the decimated frame)
    t.chirp.resize((size_t)R * q); t.wchirp.resize((size_t)R * q);
    for (int r = 0; r < R; ++r)
      for (int k = 0; k < q; ++k) {
        const int64_t rk = ((int64_t)r * k) % n_fft;
        const long double a2 = -2.0L * SSR_PI_L * (long double)rk / n_fft;
        const long double tr = cosl(a2), ti = sinl(a2);
        t.chirp[(size_t)r * q + k] = {(T)(0.5L * (cr[k] * tr - ci[k] * ti)), (T)(0.5L * (cr[k] * ti + ci[k] * tr))};
        const long double wv = w[(size_t)R * k + r];
        t.wchirp[(size_t)r * q + k] = {(T)(wv * cr[k]), (T)(wv * ci[k])};
      }
    std::vector<long double> br(N, 0.0L), bi(N, 0.0L);
    for (int m = 0; m < q; ++m) {
      br[m] = cr[m]; bi[m] = -ci[m];
      if (m) { br[N - m] = cr[m]; bi[N - m] = -ci[m]; }
    }
    if (ssr_is_pow2(N)) ssr_host_fft_ld(br, bi); else ssr_host_dft_ld(br, bi);
    t.bfilt.resize(N);
    for (int i = 0; i < N; ++i) t.bfilt[i] = {(T)(br[i] / N), (T)(bi[i] / N)};
  }
  return true;
}
