/* Oracle (TEST INFRASTRUCTURE): torchlibrosa's STFT -> hard low-pass -> ISTFT in the reference's arithmetic class, with the one
 * thing the published code leaves to the BLAS kernel - the float32 accumulation order of the dense DFT dot products - FIXED so
 * that a GPU kernel can be compared with it bit for bit:
 *
 *   a dot product over K terms is evaluated as chains of `kb` fused multiply-adds (ascending k, starting from 0.0f), the chains'
 *   results added to a float32 total in order ("K-blocked sgemm with FMA": what oneDNN / MKL kernels do, with kb their block).
 *
 * Everything else follows the published modules step by step (torchlibrosa 0.0.7-0.0.9 stft.py, as wrapped by
 * ssr_eval/dsp.py:21-39,76-81,107-119 and driven by ssr_eval/lowpass.py:17-28):
 *   STFT.forward     reflect pad n_fft/2, real = conv1d(x, W_re, stride hop), imag = conv1d(x, W_im)       float32
 *   spectrogram_phase mag = clamp(re*re + im*im, 1e-8)^0.5, cos = re / mag, sin = im / mag               float32, no contraction
 *   lowpass.py:24-25  mag[cut:] = 0;   ISTFT input = (mag * cos, mag * sin)
 *   ISTFT.forward    Hermitian mirror, s = conv_real(full_re) - conv_imag(full_im) (1x1 convs over n_fft channels; all-zero
 *                    channels add exact zeros and are skipped), F.fold (col2im: for a given output sample the frames are added
 *                    in DESCENDING frame order - the loop runs over the kernel offset), divided by the folded hann^2 (same order,
 *                    float32) clamped at 1e-11, trimmed to [n_fft/2, n_fft/2 + length).
 * The weight matrices are inputs (oracle/stft.py::tl_weights, or the tables libssrhip builds: ssr_tl_weights).
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -shared -fPIC (oracle/tl_chain.py).  fmaf() must be the correctly rounded one
 * (-mfma inlines vfmadd; glibc's software fmaf is also exact).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* out[c] = sum_k a[k] * w[k * ldw + c], c < ncols, chains of kb terms.  acc / tot: scratch [ncols]. */
static void chain_rows(const float* a, int K, const float* w, int64_t ldw, int ncols, int kb, float* acc, float* tot) {
  for (int c = 0; c < ncols; ++c) tot[c] = 0.0f;
  for (int k0 = 0; k0 < K; k0 += kb) {
    const int k1 = k0 + kb < K ? k0 + kb : K;
    for (int c = 0; c < ncols; ++c) acc[c] = 0.0f;
    for (int k = k0; k < k1; ++k) {
      const float av = a[k];
      const float* wr = w + (int64_t)k * ldw;
      for (int c = 0; c < ncols; ++c) acc[c] = fmaf(av, wr[c], acc[c]);
    }
    for (int c = 0; c < ncols; ++c) tot[c] = tot[c] + acc[c];
  }
}

/* Forward transform of one padded signal.  xp: [n + n_fft] reflect-padded; wre_t / wim_t: [n_fft][ldw] = weights TRANSPOSED
 * (row = sample j, column = bin k); re / im: [T][nb] for bins < nb. */
void tl_chain_stft(const float* xp, int T, int n_fft, int hop, const float* wre_t, const float* wim_t, int64_t ldw, int nb,
                   int kb, float* re, float* im) {
  float* acc = (float*)malloc(sizeof(float) * nb * 2);
  for (int t = 0; t < T; ++t) {
    chain_rows(xp + (int64_t)t * hop, n_fft, wre_t, ldw, nb, kb, acc, re + (int64_t)t * nb);
    chain_rows(xp + (int64_t)t * hop, n_fft, wim_t, ldw, nb, kb, acc, im + (int64_t)t * nb);
  }
  free(acc);
}

/* spectrogram_phase + the cut + mag * cos / mag * sin (dsp.py:76-81, lowpass.py:24-25, dsp.py:112-116).  cut >= nb: no cut. */
void tl_chain_magphase_cut(float* re, float* im, int64_t rows, int nb, int cut, float eps) {
  for (int64_t r = 0; r < rows; ++r)
    for (int k = 0; k < nb; ++k) {
      float* pr = re + r * nb + k;
      float* pi = im + r * nb + k;
      if (k >= cut) { *pr = 0.0f; *pi = 0.0f; continue; }
      const float a = *pr * *pr, b = *pi * *pi;
      float s = a + b;
      if (s < eps) s = eps;
      const float mag = sqrtf(s);
      const float c = *pr / mag, sn = *pi / mag;
      *pr = mag * c;
      *pi = mag * sn;
    }
}

/* Inverse transform + fold + window-sum division + trim.  re / im: [T][nb] (bins >= nbz are taken as zero);
 * ire_t / iim_t: [n_fft][n_fft] TRANSPOSED inverse weights (row = bin k of the FULL spectrum, column = output sample);
 * w2: window^2 float32 [n_fft]; out: [length]; start: first kept sample of the overlap-added signal (ISTFT._trim_edges: n_fft / 2
 * when the forward transform was centred, 0 otherwise); samples past its end are written as 0. */
void tl_chain_istft(const float* re, const float* im, int T, int nb, int nbz, int n_fft, int hop, const float* ire_t,
                    const float* iim_t, const float* w2, int kb, int length, int start, float* out) {
  const int half = n_fft / 2;
  if (nbz > nb) nbz = nb;
  int mmax = nbz - 1 < half - 1 ? nbz - 1 : half - 1;       /* mirrored bins 1 .. mmax */
  if (mmax < 0) mmax = 0;
  const int K = nbz + mmax;                                  /* non-zero channels of the full spectrum, ascending channel order */
  float* fre = (float*)malloc(sizeof(float) * (K + 1));
  float* fim = (float*)malloc(sizeof(float) * (K + 1));
  float* wre = (float*)malloc(sizeof(float) * (size_t)(K + 1) * n_fft);   /* the K rows of the weight tables, compacted */
  float* wim = (float*)malloc(sizeof(float) * (size_t)(K + 1) * n_fft);
  for (int j = 0; j < K; ++j) {
    const int ch = j < nbz ? j : n_fft - mmax + (j - nbz);
    memcpy(wre + (size_t)j * n_fft, ire_t + (size_t)ch * n_fft, sizeof(float) * n_fft);
    memcpy(wim + (size_t)j * n_fft, iim_t + (size_t)ch * n_fft, sizeof(float) * n_fft);
  }
  float* s = (float*)malloc(sizeof(float) * (size_t)T * n_fft);
  float* acc = (float*)malloc(sizeof(float) * n_fft);
  float* sr = (float*)malloc(sizeof(float) * n_fft);
  float* si = (float*)malloc(sizeof(float) * n_fft);
  for (int t = 0; t < T; ++t) {
    const float* r = re + (int64_t)t * nb;
    const float* i = im + (int64_t)t * nb;
    for (int j = 0; j < K; ++j) {
      if (j < nbz) { fre[j] = r[j]; fim[j] = i[j]; }
      else { const int src = mmax - (j - nbz); fre[j] = r[src]; fim[j] = -i[src]; }
    }
    chain_rows(fre, K, wre, n_fft, n_fft, kb, acc, sr);
    chain_rows(fim, K, wim, n_fft, n_fft, kb, acc, si);
    for (int m = 0; m < n_fft; ++m) s[(size_t)t * n_fft + m] = sr[m] - si[m];
  }
  for (int p = 0; p < length; ++p) {
    const int q = p + start;
    if (q >= (T - 1) * hop + n_fft) { out[p] = 0.0f; continue; }
    int t_hi = q / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    float y = 0.0f, ws = 0.0f;
    for (int t = t_hi; t >= 0 && q - t * hop < n_fft; --t) {
      y = y + s[(size_t)t * n_fft + (q - t * hop)];
      ws = ws + w2[q - t * hop];
    }
    if (ws < 1e-11f) ws = 1e-11f;
    out[p] = y / ws;
  }
  free(fre); free(fim); free(wre); free(wim); free(s); free(acc); free(sr); free(si);
}
