/* Oracle (TEST INFRASTRUCTURE): torchlibrosa's STFT -> hard low-pass -> ISTFT in the reference's arithmetic, with the one
 * thing the published code leaves to the BLAS kernel - the float32 accumulation order of the dense DFT dot products - written
 * out.  The order restated here is the one torch-CPU (2.10, oneDNN 3.7.1, AVX-512 host, >= 2 threads) runs for
 * torchlibrosa's two convolutions, established by BIT-FOR-BIT comparison with F.conv1d in this container
 * (tests/test_oracle.py::test_tl_chain_is_torch_conv1d_bit_for_bit; profiles/r05_lowpass_class_members.json):
 *
 *   STFT.forward  (F.conv1d, stride hop, kernel n_fft; signals of >= 55 frames):  ONE chain of n_fft fused multiply-adds per
 *                 output, ascending sample index, starting from 0.0f                                             (kb = 0)
 *   ISTFT.forward (the two 1x1 F.conv1d over the n_fft channels of the Hermitian-mirrored spectrum): chains of fused
 *                 multiply-adds over BLOCKS OF 256 CHANNELS of the full spectrum (channel index / 256), ascending channel,
 *                 each from 0.0f; the blocks' results added to a float32 total in ascending block order       (kbf = 256)
 *                 - an all-zero channel adds exact zeros and is skipped.
 * (Single-threaded torch blocks the inverse differently - 384 or 448 channels, by shape -, MKL sgemm 384 at 8 threads, OpenBLAS
 * 512: other members of the same class, up to 0.8 % apart in LSD of the low-passed signal.)
 *
 * Everything else follows the published modules step by step (torchlibrosa 0.0.7-0.0.9 stft.py, as wrapped by
 * ssr_eval/dsp.py:21-39,76-81,107-119 and driven by ssr_eval/lowpass.py:17-28):
 *   STFT.forward     reflect pad n_fft/2, real = conv1d(x, W_re, stride hop), imag = conv1d(x, W_im)       float32
 *   spectrogram_phase mag = clamp(re*re + im*im, 1e-8)^0.5, cos = re / mag, sin = im / mag               float32, no contraction
 *   lowpass.py:24-25  mag[cut:] = 0;   ISTFT input = (mag * cos, mag * sin)
 *   ISTFT.forward    Hermitian mirror, s = conv_real(full_re) - conv_imag(full_im), F.fold (col2im: for a given output sample
 *                    the frames are added in DESCENDING frame order - the loop runs over the kernel offset), divided by the
 *                    folded hann^2 (same order, float32) clamped at 1e-11, trimmed to [n_fft/2, n_fft/2 + length).
 * The weight matrices are inputs (oracle/stft.py::tl_weights = torchlibrosa's own numpy construction).
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -shared -fPIC (oracle/tl_chain.py).  fmaf() must be the correctly rounded one
 * (-mfma inlines vfmadd; glibc's software fmaf is also exact).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* out[c] = sum_k a[k] * w[k * ldw + c], c < ncols, chains of kb terms (kb <= 0: one chain).  acc / tot: scratch [ncols]. */
static void chain_rows(const float* a, int K, const float* w, int64_t ldw, int ncols, int kb, float* acc, float* tot) {
  if (kb <= 0) kb = K > 0 ? K : 1;
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
 * ire_t / iim_t: [n_fft][n_fft] TRANSPOSED inverse weights (row = channel k of the FULL spectrum, column = output sample);
 * kbf: channels per chain block of the full spectrum (<= 0: one chain over all channels);
 * w2: window^2 float32 [n_fft]; out: [length]; start: first kept sample of the overlap-added signal (ISTFT._trim_edges: n_fft / 2
 * when the forward transform was centred, 0 otherwise); samples past its end are written as 0. */
void tl_chain_istft(const float* re, const float* im, int T, int nb, int nbz, int n_fft, int hop, const float* ire_t,
                    const float* iim_t, const float* w2, int kbf, int length, int start, float* out) {
  const int half = n_fft / 2;
  if (nbz > nb) nbz = nb;
  if (kbf <= 0) kbf = n_fft;
  int mmax = nbz - 1 < half - 1 ? nbz - 1 : half - 1;       /* mirrored bins 1 .. mmax = channels n_fft - mmax .. n_fft - 1 */
  if (mmax < 0) mmax = 0;
  const int K = nbz + mmax;                                  /* non-zero channels of the full spectrum, ascending channel order */
  float* s = (float*)malloc(sizeof(float) * (size_t)T * n_fft);
  float* a1 = (float*)malloc(sizeof(float) * n_fft);
  float* a2 = (float*)malloc(sizeof(float) * n_fft);
  float* t1 = (float*)malloc(sizeof(float) * n_fft);
  float* t2 = (float*)malloc(sizeof(float) * n_fft);
  for (int t = 0; t < T; ++t) {
    const float* r = re + (int64_t)t * nb;
    const float* i = im + (int64_t)t * nb;
    for (int m = 0; m < n_fft; ++m) { t1[m] = 0.0f; t2[m] = 0.0f; a1[m] = 0.0f; a2[m] = 0.0f; }
    int prev_block = -1;
    for (int j = 0; j < K; ++j) {
      const int src = j < nbz ? j : mmax - (j - nbz);        /* bin the channel's value comes from */
      const int ch = j < nbz ? j : n_fft - src;              /* channel of the full spectrum */
      const float fr = r[src], fi = j < nbz ? i[src] : -i[src];
      if (ch / kbf != prev_block && prev_block >= 0)         /* a chain ends: total += chain, restart from 0 */
        for (int m = 0; m < n_fft; ++m) { t1[m] = t1[m] + a1[m]; t2[m] = t2[m] + a2[m]; a1[m] = 0.0f; a2[m] = 0.0f; }
      prev_block = ch / kbf;
      const float* wr = ire_t + (size_t)ch * n_fft;
      const float* wi = iim_t + (size_t)ch * n_fft;
      for (int m = 0; m < n_fft; ++m) { a1[m] = fmaf(fr, wr[m], a1[m]); a2[m] = fmaf(fi, wi[m], a2[m]); }
    }
    for (int m = 0; m < n_fft; ++m) s[(size_t)t * n_fft + m] = (t1[m] + a1[m]) - (t2[m] + a2[m]);
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
  free(s); free(a1); free(a2); free(t1); free(t2);
}
