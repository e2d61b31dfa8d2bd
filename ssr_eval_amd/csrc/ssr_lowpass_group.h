// Kernel body K6, fused: STFT-domain hard low-pass / inverse STFT of 2048-point plans WITH the overlap-add inside the kernel -
// no frame / segment workspace in HBM at all.  Semantics: ssr_lowpass.h (ssr_eval/lowpass.py:17-28, ssr_eval/dsp.py:76-119).
//
// Round 2 wrote one segment per frame pair (2.2 GB per 1024 x 4 s at hop 441) for k_ola_paired to read back (2.3 GB) - 3.95x the
// algorithmic traffic of 2 n 4 bytes per signal.  A per-wave accumulation ring did not pay (LDS: occupancy; global memory: the
// rings thrash L2, profiles/r02_notes.md).  Here the unit of work is a ROUND: the four waves of a workgroup transform four
// consecutive frame pairs (8 frames) at the same time - each wave autonomous on its own 17 KB exchange array, exactly as in
// ssr_lowpass_wave.h - and once all four are done the four arrays are free, so TOGETHER they serve as the accumulation buffer
// of the round's 8 hop + (n_fft - hop) output positions:
//
//   transforms (no barrier inside) | barrier | buffer <- carried tail, zeros | frames added, one wave colour at a time |
//   finished positions x 1 / overlap-added squared window -> out (coalesced float32) | the n_fft - hop unfinished ones -> tail
//
//   * a wave adds its own two frames one after the other (ds_add_f64: nothing comes back, no wait; the LDS operations of one
//     wave execute in order); waves whose frame pairs are >= NCW apart never overlap (NCW = 3 at hop 441 and 512), so NCW
//     barrier-separated steps add everything - in an order that depends only on the absolute frame indices: the output bits do
//     not depend on the launch geometry;
//   * the tail (positions the NEXT round's frames still reach) lives in 12.9 KB behind the arrays: 4 x 16.9 KB + 12.9 KB = 80.5 KB
//     per workgroup -> TWO workgroups of four waves per CU (the same two waves per SIMD as before), which are in different
//     phases of their rounds most of the time: one transforms while the other adds up and stores (eight waves in ONE workgroup
//     per CU were measured slower than the unfused path: every wave waits out every barrier);
//   * a chunk that starts inside a signal first re-runs the frame pairs that reach into its first round (a warm-up round whose
//     output is discarded) instead of exchanging boundary fringes through memory; for signals of a few seconds a chunk is the
//     whole signal and nothing is recomputed;
//   * sums are formed in float64 from the unrounded windowed frames and rounded to float32 once (the paired-segment path rounded
//     each frame pair to float32 first); the normalisation multiplies by the tabulated reciprocal of the window sum.
// HBM traffic = the signal in + the signal out.
#pragma once
#include "ssr_lowpass_wave.h"

constexpr int SSR_LG_WAVES = 4, SSR_LG_NT = 64 * SSR_LG_WAVES, SSR_LG_FR = 2 * SSR_LG_WAVES;   // frames per round

struct SsrLowpassGroupParams {
  SsrLowpassParams<double> lp;   // in / in_off / len / cut / frame_off (ISTFT rows) / n_fft / hop / window / tw / spec_re / spec_im
  const int64_t* out_off;        // [n_items]
  float* out;
  const double* window64;        // [n_fft] Hann (edge positions: ssr_ola_wss)
  const double* wss_tab;         // [hop] overlap-added squared window where every overlapping frame exists
  const double* wss_rcp_tab;     // [hop] its reciprocal
  int rounds_per_chunk, n_chunks;
};

SSR_HD constexpr size_t ssr_lowpass_group_lds_bytes(int hop) {
  return sizeof(double) * ((size_t)SSR_LG_WAVES * SSR_W_PN + (size_t)(SSR_W_N - hop));
}
// hop range of the fused path: the round's buffer must fit the workgroup's arrays, and ONE warm-up round must cover every
// frame that reaches into the next round (ceil(n_fft / hop) / 2 <= waves)
SSR_HD constexpr bool ssr_lowpass_group_ok(int hop) {
  return hop >= 1 && hop <= SSR_W_N / 2 && ((SSR_W_N + hop - 1) / hop) / 2 <= SSR_LG_WAVES &&
         (SSR_LG_FR - 1) * hop + SSR_W_N <= SSR_LG_WAVES * SSR_W_PN;
}

// grid = n_items * n_chunks workgroups of SSR_LG_NT threads
template <bool ANALYSIS, typename BLK>
SSR_BODY void ssr_lowpass_group_body(const SsrLowpassGroupParams& gp, BLK& blk, int chunk, int item, char* lds_base) {
  typedef double T;
  constexpr int N = SSR_W_N, F = N / 2 + 1, NW = SSR_LG_WAVES, NT = SSR_LG_NT, FR = SSR_LG_FR;
  constexpr bool SPLIT = true;
  using Regs = SsrLowpassWaveRegs<T>;
  const SsrLowpassParams<T>& p = gp.lp;
  T* arrays = reinterpret_cast<T*>(lds_base);                       // [NW][SSR_W_PN]: one exchange array per wave
  double* acc = reinterpret_cast<double*>(lds_base);                // the round's accumulation buffer ALIASES the arrays
  double* tail = acc + (size_t)NW * SSR_W_PN;                       // [N - hop]
  const int n = p.len[item], hop = p.hop;
  const int64_t out0 = gp.out_off[item];
  SSR_REGS(Regs, regs, blk);
  if (n <= N / 2) {           // precondition (include/ssr_hip.h): len > n_fft / 2; such an item is skipped, its output zeroed
    SSR_PHASE(blk, regs, if (chunk == 0) for (int s = tid; s < n; s += NT) gp.out[out0 + s] = 0.0f);
    return;
  }
  const int n_frames = ssr_num_frames_dev(n, N, hop);
  const int n_units = (n_frames + 1) / 2, n_rounds = (n_units + NW - 1) / NW;
  const int r_begin = chunk * gp.rounds_per_chunk;
  if (r_begin >= n_rounds) return;
  const int r_end = (r_begin + gp.rounds_per_chunk < n_rounds) ? r_begin + gp.rounds_per_chunk : n_rounds;
  const int TL = N - hop, NB = FR * hop + TL, NC = (N + hop - 1) / hop;
  const int warm_first = NW - NC / 2;          // warm-up round: the frame pairs from this wave on reach into the next round
  int NCW = 1;                                 // wave colours: frame pairs NCW apart do not overlap ((2 NCW - 1) hop >= n_fft)
  while (NCW < NW && (2 * NCW - 1) * hop < N) ++NCW;
  const int64_t row0 = p.frame_off ? p.frame_off[item] : 0;
  const T inv_n = (T)1 / (T)N;
  const SsrView<float> vs(ANALYSIS ? p.in + p.in_off[item] : p.spec_re, ANALYSIS ? n : 0);
  const SsrView<float> vre(ANALYSIS ? nullptr : p.spec_re + row0 * F, ANALYSIS ? 0 : (int64_t)n_frames * F);
  const SsrView<float> vim(ANALYSIS ? nullptr : p.spec_im + row0 * F, ANALYSIS ? 0 : (int64_t)n_frames * F);
  const SsrView<T> vw(p.window, N);
  const SsrView<cx<T>> vt(p.tw, N + SSR_W_TWP);
  const int cut = ANALYSIS ? p.cut[item] : F;

  SSR_PHASE(blk, regs, for (int i = tid; i < TL; i += NT) tail[i] = 0.0);
  BLK blk0 = blk;
#define SSR_LG_L (SsrWaveBuf<T>{arrays + ssr_wave_of(tid) * SSR_W_PN, arrays + ssr_wave_of(tid) * SSR_W_PN})
#define SSR_LG_LOAD_WIN ; { const int lane_ = tid & 63; SSR_UNROLL for (int r_ = 0; r_ < SSR_W_P / 2; ++r_) R.wl[r_] = vw.at(SSR_UIDX(lane_ + 64 * r_)); }
#define VT vt
  for (int r = (r_begin > 0 ? r_begin - 1 : 0); r < r_end; ++r) {
    const bool warm = r < r_begin;
    blk = blk0; ssr_launder(blk);
    // ---- the round's eight frame pairs, one per wave, no barrier (frames beyond the signal re-read the last one: their
    // results are never added)
    if constexpr (ANALYSIS) {
      SSR_WPHASE(blk, regs, {
        const int lane = tid & 63, g = NW * r + ssr_wave_of(tid);
        const bool b_valid = 2 * g + 1 < n_frames;
        SSR_UNROLL for (int i = 0; i < SSR_W_P / 2; ++i) R.wl[i] = vw.at(SSR_UIDX(lane + 64 * i));
        ssr_lowpass_wave_prefetch<T>(R, lane, vs, g, hop, n, n_frames);
        SSR_UNROLL for (int i = 0; i < SSR_W_P; ++i) {
          const T w = (i < SSR_W_P / 2) ? R.wl[i] : (T)1 - R.wl[i - SSR_W_P / 2];
          R.v[i] = {(T)R.pa[i] * w, b_valid ? (T)R.pb[i] * w : (T)0};
        }
        ssr_dft32(R.v);
      });
      SSR_W_FFT_TAIL(blk, blk0, regs, SSR_LG_L, );
      SSR_WPHASE(blk, regs, {
        const int lane = tid & 63;
        cx<T> z[SSR_W_P];
        SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q = 0; q < 8; ++q) {
          const int k = lane + 64 * b + 256 * q;
          const bool zero = (k >= cut) && (k <= N - cut);
          z[b + 4 * q] = {zero ? (T)0 : R.v[8 * b + q].y, zero ? (T)0 : R.v[8 * b + q].x};
        }
        SSR_UNROLL for (int i = 0; i < SSR_W_P; ++i) R.v[i] = z[i];
        ssr_dft32(R.v);
      });
    } else {
      SSR_WPHASE(blk, regs, {
        const int lane = tid & 63, g = NW * r + ssr_wave_of(tid);
        const int ta = (2 * g < n_frames) ? 2 * g : n_frames - 1, tbr = 2 * g + 1;
        const bool b_valid = tbr < n_frames;
        SSR_UNROLL for (int i = 0; i < SSR_W_P; ++i) {
          const int k = lane + 64 * i;
          const int kk = (k <= N / 2) ? k : N - k;
          const T sgn = (k <= N / 2) ? (T)1 : (T)-1;
          const bool edge = (kk == 0) || (kk == N / 2);
          const int64_t oa = (int64_t)ta * F, ob = (int64_t)(b_valid ? tbr : ta) * F;
          const T ar = (T)vre.at(SSR_UIDX(kk), oa), ai = edge ? (T)0 : sgn * (T)vim.at(SSR_UIDX(kk), oa);
          const T br = b_valid ? (T)vre.at(SSR_UIDX(kk), ob) : (T)0;
          const T bi = (b_valid && !edge) ? sgn * (T)vim.at(SSR_UIDX(kk), ob) : (T)0;
          R.v[i] = {ai + br, ar - bi};
        }
        ssr_dft32(R.v);
      });
    }
    SSR_W_FFT_TAIL(blk, blk0, regs, SSR_LG_L, SSR_LG_LOAD_WIN);
    // registers: swap(N * IFFT): frame 2g = .y, frame 2g+1 = .x at sample m = lane + 64 (b + 4 q).  Synthesis window, 1 / N.
    blk = blk0; ssr_launder(blk);
    SSR_WPHASE(blk, regs, {
      SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q = 0; q < 8; ++q) {
        const int i2 = b + 4 * q, i = 8 * b + q;
        const T w = ((i2 < SSR_W_P / 2) ? R.wl[i2] : (T)1 - R.wl[i2 - SSR_W_P / 2]) * inv_n;
        R.v[i] = {R.v[i].x * w, R.v[i].y * w};
      }
    });
    SSR_PHASE(blk, regs, {});                                       // every wave is done with its exchange array
    // ---- the arrays become the round's buffer: position pp <-> padded position FR r hop + pp
    SSR_PHASE(blk, regs, for (int pp = tid; pp < NB; pp += NT) acc[pp] = (pp < TL) ? tail[pp] : 0.0);
    for (int c = 0; c < NCW; ++c) {
      SSR_PHASE(blk, regs, {
        const int lane = tid & 63, wv = ssr_wave_of(tid), g = NW * r + wv, ta = 2 * g, tb = ta + 1;
        const bool act = !(warm && wv < warm_first) && wv % NCW == c;
        if (act && ta < n_frames) {
          double* a = acc + (ta - FR * r) * hop + lane;
          SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q = 0; q < 8; ++q) SSR_LDS_ACCUM(a + 64 * (b + 4 * q), R.v[8 * b + q].y);
        }
        if (act && tb < n_frames) {
          double* a = acc + (tb - FR * r) * hop + lane;
          SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q = 0; q < 8; ++q) SSR_LDS_ACCUM(a + 64 * (b + 4 * q), R.v[8 * b + q].x);
        }
      });
    }
    // ---- positions no later frame reaches are final (after the signal's last round: all of them); the rest is the tail
    SSR_PHASE(blk, regs, {
      const int p_end = (r == n_rounds - 1) ? NB : FR * hop;
      if (!warm)
        for (int pp = tid; pp < p_end; pp += NT) {
          const int pos = FR * r * hop + pp, s = pos - N / 2;
          if (s >= 0 && s < n) {
            const int t0 = pos / hop;
            const double y = (t0 >= (N - 1) / hop && t0 <= n_frames - 1) ? acc[pp] * gp.wss_rcp_tab[pos - t0 * hop]
                                                                        : acc[pp] / ssr_ola_wss(gp.window64, N, hop, n_frames, pos);
            gp.out[out0 + s] = (float)y;
          }
        }
      for (int i = tid; i < TL; i += NT) tail[i] = acc[FR * hop + i];
    });
  }
#undef VT
#undef SSR_LG_LOAD_WIN
#undef SSR_LG_L
}
