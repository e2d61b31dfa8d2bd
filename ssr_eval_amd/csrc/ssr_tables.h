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

// engine choice for an n_fft: direct 2^LOGN FFT when n_fft is a power of two in [256, 4096];
// otherwise Bluestein with M = 2^LOGM >= max(256, 2*n_fft - 1), M <= 8192.
struct SsrEngine { bool ok; bool bluestein; int logn; };
inline SsrEngine ssr_pick_engine(int n_fft) {
  SsrEngine e{false, false, 0};
  if (n_fft < 2) return e;
  if (ssr_is_pow2(n_fft) && n_fft >= 256 && n_fft <= 4096) { e.ok = true; e.logn = ssr_ilog2(n_fft); return e; }
  int m = 256;
  while (m < 2 * n_fft - 1) m <<= 1;
  if (m > 8192) return e;
  e.ok = true; e.bluestein = true; e.logn = ssr_ilog2(m);
  return e;
}

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

template <typename T> bool ssr_build_tables(int n_fft, SsrTables<T>& t) {
  t.eng = ssr_pick_engine(n_fft);
  t.n_fft = n_fft;
  if (!t.eng.ok) return false;
  const int N = 1 << t.eng.logn;
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
  if (t.eng.bluestein) {
    std::vector<long double> cr(n_fft), ci(n_fft);
    for (int k = 0; k < n_fft; ++k) {
      const int64_t kk = ((int64_t)k * k) % (2 * (int64_t)n_fft);  // exact phase reduction
      const long double ang = -SSR_PI_L * (long double)kk / n_fft;
      cr[k] = cosl(ang); ci[k] = sinl(ang);
    }
    t.chirp.resize(n_fft); t.wchirp.resize(n_fft);
    for (int k = 0; k < n_fft; ++k) {
      t.chirp[k] = {(T)0.5 * (T)cr[k], (T)0.5 * (T)ci[k]};
      t.wchirp[k] = {(T)(w[k] * cr[k]), (T)(w[k] * ci[k])};
    }
    std::vector<long double> br(N, 0.0L), bi(N, 0.0L);
    for (int m = 0; m < n_fft; ++m) {
      br[m] = cr[m]; bi[m] = -ci[m];
      if (m) { br[N - m] = cr[m]; bi[N - m] = -ci[m]; }
    }
    ssr_host_fft_ld(br, bi);
    t.bfilt.resize(N);
    for (int i = 0; i < N; ++i) t.bfilt[i] = {(T)(br[i] / N), (T)(bi[i] / N)};
  }
  return true;
}
