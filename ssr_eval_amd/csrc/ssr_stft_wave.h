// Kernel body K1-K4, wave-autonomous engine: the 2048-point pair STFT + fused LSD / SISpec epilogue with ONE WAVE PER
// FRAME and no workgroup barrier anywhere.
//
// Why: the 256-thread engine of ssr_stft.h runs a frame on four waves in lock step - eight s_barrier per frame (14 % of
// the kernel, measured), three LDS exchanges, and four waves that all stall together.  Here a frame belongs to one wave:
//   * 64 lanes x 32 points.  The first pass is a radix-32 DFT done entirely IN the lane's registers (4 x 8 with
//     compile-time twiddles), so the transform is 32 x 8 x 8 = three passes and only TWO exchanges through LDS;
//   * exchanges are ordered by the hardware (the DS operations of one wave execute in order), so a "phase boundary" is a
//     compiler-only wave-scope fence: no s_barrier, no s_waitcnt between a wave's writes and its reads;
//   * what a lane needs every frame is loop-invariant and stays in registers: its 16 window values
//     (w[m + N/2] = 1/2 - w[m]) and the 7 twiddles of the second pass (w^(8 (l mod 32) q) does not depend on the
//     butterfly); the first pass needs none.  Only the third pass loads 3 table twiddles per butterfly;
//   * the per-frame LSD reduction is a wave shuffle, the silent-frame vote a ballot.
// Occupancy is bounded by LDS, not by registers: one wave owns one 2048-point buffer (33 KB float64: 4 waves per CU, one
// per SIMD, up to 512 VGPRs each) - or, with the re / im parts exchanged one after the other through a single array
// (SPLIT = true), 17 KB: 8 waves per CU at <= 256 VGPRs.
//
// Arithmetic (window, two-for-one packing, separation, magnitudes, metric terms) is the one of ssr_stft.h; the parity
// tests run both engines against the oracle and against each other.  Reference semantics: ssr_eval/metrics.py:26-30,
// 109-121 (see ssr_stft.h).
#pragma once
#include "ssr_stft.h"

// (SSR_WPHASE / SSR_WAVE_SYNC: ssr_block.h)

// Developer build (-DSSR_PHASE_CLOCKS): shader-clock stamps at the phase boundaries of k_stft_wave's frame loop, summed per
// launch (tu_stft.inc prints them).  SSR_CLK(i) is empty everywhere else.
#define SSR_CLK(i)
#if defined(SSR_PHASE_CLOCKS) && !defined(SSR_HOST_EMU)
static __device__ unsigned long long ssr_dbg_clk[8];
#define SSR_CLK_NOW(i) do { SSR_SCHED_BARRIER(); clk_[i] = __builtin_readcyclecounter(); SSR_SCHED_BARRIER(); } while (0)
#endif

// k_stft_wave's third pass in PAIRED butterfly order (below): 1 = the product, 0 = the ascending order + half exchange of
// rounds 2-5 (A/B builds)
#ifndef SSR_WAVE_PAIRED
#define SSR_WAVE_PAIRED 1
#endif
// (developer builds: -DSSR_WAVE_EPI_SB_OFF lets the compiler place the epilogue's sample / window requests freely)
#ifdef SSR_WAVE_EPI_SB_OFF
#define SSR_WAVE_EPI_SB() do {} while (0)
#else
#define SSR_WAVE_EPI_SB() SSR_SCHED_BARRIER()
#endif
#ifdef SSR_WAVE_NOPF
#define SSR_WAVE_PF(...) do {} while (0)
#else
#define SSR_WAVE_PF(...) do { __VA_ARGS__; } while (0)
#endif
#ifndef SSR_WAVE_PF1_SUMS
#define SSR_WAVE_PF1_SUMS 1
#endif
#ifndef SSR_WAVE_PF1
#define SSR_WAVE_PF1 0
#endif
constexpr int SSR_W_N = 2048, SSR_W_L = 64, SSR_W_P = 32;     // points, lanes, points per lane
constexpr int SSR_W_TWP = 7 * 32 + 12 * 64;                   // lane-ordered twiddle copies behind the table (= SSR_WAVE_TWP)
SSR_DEV int ssr_wpad(int i) { return i + (i >> 5); }            // lane stride 32 -> 33 doubles: conflict-free ds_*_b64
constexpr int SSR_W_PN = SSR_W_N + (SSR_W_N >> 5) + 1;
constexpr int SSR_W_IMOFF = 1024 + 32;    // SPLIT: the upper-half imaginary parts of the final exchange sit behind the real parts
constexpr int SSR_W_ROWB = 4 * 1028;      // ROWS_VIA_LDS: bytes between the two staged magnitude rows (1025 bins, 16-byte aligned)
#ifndef SSR_WAVE_ROWS_VIA_LDS
#define SSR_WAVE_ROWS_VIA_LDS 0           // developer build: 1 (measured: a null - see ROWS_VIA_LDS below)
#endif

// exp(-2 pi i m / 32), m = 0..21 (the exponents n2 * k1 of the in-lane 4 x 8 decomposition)
template <typename T> SSR_DEV cx<T> ssr_w32(int m) {
  constexpr double c[22] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708, 0.70710678118654752440,
                            0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785, 0.0,
                            -0.19509032201612826785, -0.38268343236508977173, -0.55557023301960222474, -0.70710678118654752440,
                            -0.83146961230254523708, -0.92387953251128675613, -0.98078528040323044913, -1.0,
                            -0.98078528040323044913, -0.92387953251128675613, -0.83146961230254523708, -0.70710678118654752440,
                            -0.55557023301960222474};
  constexpr double s[22] = {0.0, -0.19509032201612826785, -0.38268343236508977173, -0.55557023301960222474, -0.70710678118654752440,
                            -0.83146961230254523708, -0.92387953251128675613, -0.98078528040323044913, -1.0,
                            -0.98078528040323044913, -0.92387953251128675613, -0.83146961230254523708, -0.70710678118654752440,
                            -0.55557023301960222474, -0.38268343236508977173, -0.19509032201612826785, 0.0,
                            0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474, 0.70710678118654752440,
                            0.83146961230254523708};
  return {(T)c[m], (T)s[m]};
}

// v[r], r = 8 n1 + n2  (n1 < 4, n2 < 8)   ->   v[8 k1 + k2] = DFT32(v)[k1 + 4 k2]      (in place, registers only)
// NZ: only v[0 .. NZ) can be non-zero (a zero-padded Bluestein input: NZ = 12 or 16 rows of 64 samples); the first
// radix-4 stage then skips the additions of zeros - the same values, 96 fewer instructions at NZ = 12.
template <typename T, int NZ = 32> SSR_DEV void ssr_dft32(cx<T>* v) {
  const T h = (T)0.70710678118654752440;
  static_assert(NZ == 32 || (NZ > 8 && NZ <= 16), "full, or two non-zero row groups");
  SSR_UNROLL for (int n2 = 0; n2 < 8; ++n2) {
    cx<T> t[4] = {v[n2], v[8 + n2], v[16 + n2], v[24 + n2]};
    if (NZ == 32) {
      ssr_bfly4(t);
    } else if (8 + n2 < NZ) {                       // t[2] = t[3] = 0
      const cx<T> r = cmul_negi(t[1]);
      t[2] = csub(t[0], t[1]); t[3] = csub(t[0], r);
      t[1] = cadd(t[0], r);    t[0] = cadd(t[0], v[8 + n2]);
    } else {                                        // only t[0]
      t[1] = t[0]; t[2] = t[0]; t[3] = t[0];
    }
    SSR_UNROLL for (int k1 = 0; k1 < 4; ++k1) {
      const int m = n2 * k1;                       // compile-time after unrolling: the special angles cost no multiply
      cx<T> x = t[k1];
      if (m == 0) {
      } else if (m == 8) {
        x = cmul_negi(x);
      } else if (m == 16) {
        x = {-x.x, -x.y};
      } else if (m == 4) {
        x = {h * (x.x + x.y), h * (x.y - x.x)};
      } else if (m == 12) {
        x = {h * (x.y - x.x), -h * (x.x + x.y)};
      } else if (m == 20) {
        x = {-h * (x.x + x.y), h * (x.x - x.y)};
      } else {
        x = cmul(x, ssr_w32<T>(m));
      }
      v[8 * k1 + n2] = x;
    }
  }
  SSR_UNROLL for (int k1 = 0; k1 < 4; ++k1) ssr_bfly8(v + 8 * k1);
}
// frequency held by register rho after ssr_dft32
SSR_DEV constexpr int ssr_dft32_freq(int rho) { return (rho >> 3) + 4 * (rho & 7); }

struct SsrTrue { static constexpr bool value = true; };
struct SsrFalse { static constexpr bool value = false; };

template <typename T, bool SUMS> struct SsrWaveRegs {
  cx<T> v[SSR_W_P];          // the lane's 32 points
  T tx[SSR_W_P];             // SPLIT exchange: real parts read back while the imaginary parts are still to be written
  float pa[SSR_W_P], pb[SSR_W_P];   // samples of the NEXT frame, requested one frame ahead
  T wl[SSR_W_P / 2];         // 0.5 * hann[tid + 64 r], r < 16 of the next frame: requested at the TOP of the epilogue, ahead of
                             // the magnitude stores (behind them in the in-order memory counter every frame would wait for
                             // the stores' acknowledgements)
  cx<T> tw1[7];              // w^(8 (tid mod 32) q), q = 1..7: requested while the first exchange is in flight
  cx<T> tw2[12];             // w^j, w^2j, w^4j of the four third-pass butterflies: requested while the second one is
  double lsd_total;          // lane 0: sum over the chunk's frames of the per-frame LSD
};

template <typename T, bool SPLIT> struct SsrWaveLds {
  // scratch first (doubles), then the exchange array(s)
  static constexpr size_t bytes() { return sizeof(double) * 4 + sizeof(int) * 8 + sizeof(T) * (SPLIT ? 1 : 2) * SSR_W_PN; }
  double* sc1; int* nz; T* re; T* im;
  SSR_MEMBER explicit SsrWaveLds(char* base) {
    sc1 = reinterpret_cast<double*>(base);
    nz = reinterpret_cast<int*>(sc1 + 4);           // [2 signals][2 flag sets]
    re = reinterpret_cast<T*>(nz + 8);
    im = SPLIT ? re : re + SSR_W_PN;
  }
  SSR_MEMBER bool nonzero(int which, int par) const {
    int f = nz[which * 2 + par];
#ifndef SSR_HOST_EMU
    f = __builtin_amdgcn_readfirstlane(f);
#endif
    return f != 0;
  }
};

// a wave's exchange array(s) inside a workgroup of several autonomous waves (ssr_stft_rn_wave.h, ssr_lowpass_group.h)
template <typename T> struct SsrWaveBuf { T* re; T* im; };

// LDS of one k_stft_wave wave.  The six SISpec / log-SISpec running sums of a lane live behind the exchange array (6 x 64
// float64 = 3 KB: 16,968 + 3,072 B still makes eight waves per CU) and are updated with ds_add_f64 - as registers they
// were twelve more than the variant had (11 spilled to scratch: 0.9 GB of scratch traffic per launch).
// (offset of the accumulators: the float32 engine's arrays end on a 4-byte boundary, ds_add_f64 needs 8)
template <typename T, bool SPLIT> constexpr size_t ssr_stft_wave_sums_offset() { return (SsrWaveLds<T, SPLIT>::bytes() + 7) & ~(size_t)7; }
template <typename T, bool SPLIT, bool SUMS> constexpr size_t ssr_stft_wave_lds_bytes() {
  return SUMS ? ssr_stft_wave_sums_offset<T, SPLIT>() + 6 * 64 * sizeof(double) : SsrWaveLds<T, SPLIT>::bytes();
}

// LDS slots as (per-lane base) + (compile-time offset): the offsets fold into the DS instructions' immediate fields.
//  after pass 0 : register rho = frequency q = ssr_dft32_freq(rho) of sub-transform `tid` -> slot 32 tid + q, padded 33 tid + q
//  radix-8 load : register 8 b + q = input q of butterfly j = tid + 64 b -> slot j + 256 q, padded tid + tid/32 + 66 b + 264 q
//  after pass 1 : output q of butterfly j -> slot (j - k) 8 + k + 32 q, k = j mod 32,
//                 padded 264 (tid/32) + tid mod 32 + 528 b + 33 q
//  partner read : Z[2048 - k], k = tid + 64 b + 256 q, sits at upper-half slot 1024 - k,
//                 padded 1056 - tid - ceil(tid/32) - 66 b - 264 q   (lane 0, b = q = 0 reads a slot it does not use)
struct SsrWaveBase { int st0, ld8, st1, pr; };
SSR_DEV SsrWaveBase ssr_wave_bases(int tid) {
  return {33 * tid, tid + (tid >> 5), 264 * (tid >> 5) + (tid & 31), 1056 - tid - ((tid + 31) >> 5)};
}
SSR_DEV constexpr int ssr_w_off_st0(int i) { return ssr_dft32_freq(i); }
SSR_DEV constexpr int ssr_w_off_ld8(int i) { return 66 * (i >> 3) + 264 * (i & 7); }
SSR_DEV constexpr int ssr_w_off_st1(int i) { return 528 * (i >> 3) + 33 * (i & 7); }

// One exchange through LDS: R.v[i] goes to slot WB + WO(i), then R.v[i] is reloaded from slot RB + RO(i).
// SPLIT: real parts first, imaginary parts second, through the same array.
// EXTRA: statements issued right after the (first) write phase - the table loads of the NEXT pass: the registers of the
// values just written are free at that point, and the loads' latency overlaps the exchange.
// (the read side as (per-lane bases RBASES(lane), index RIDX(bases, i)): SSR_W_EXCHANGE reads slot RB + RO(i); the paired
// second exchange of k_stft_wave reads each butterfly's inputs from its own per-lane base)
#define SSR_W_EXCHANGE_G(blk, regs, L, WB, WO, RBASES, RIDX, EXTRA)                                       \
  if constexpr (!SPLIT) {                                                                                 \
    SSR_WPHASE(blk, regs, { const int w_ = ssr_wave_bases(tid & 63).WB;                                        \
      SSR_UNROLL for (int i = 0; i < SSR_W_P; ++i) { (L).re[w_ + WO(i)] = R.v[i].x; (L).im[w_ + WO(i)] = R.v[i].y; } }); \
    SSR_WPHASE(blk, regs, { SSR_SCHED_BARRIER(); EXTRA; const auto r_ = RBASES(tid & 63);                    \
      SSR_UNROLL for (int i = 0; i < SSR_W_P; ++i) R.v[i] = {(L).re[RIDX(r_, i)], (L).im[RIDX(r_, i)]}; });   \
  } else {                                                                                                \
    SSR_WPHASE(blk, regs, { const int w_ = ssr_wave_bases(tid & 63).WB;                                        \
      SSR_UNROLL for (int i = 0; i < SSR_W_P; ++i) (L).re[w_ + WO(i)] = R.v[i].x; });                     \
    SSR_WPHASE(blk, regs, { SSR_SCHED_BARRIER(); EXTRA; const auto r_ = RBASES(tid & 63);                    \
      SSR_UNROLL for (int i = 0; i < SSR_W_P; ++i) R.tx[i] = (L).re[RIDX(r_, i)]; });                     \
    SSR_WPHASE(blk, regs, { const int w_ = ssr_wave_bases(tid & 63).WB;                                        \
      SSR_UNROLL for (int i = 0; i < SSR_W_P; ++i) (L).re[w_ + WO(i)] = R.v[i].y; });                     \
    SSR_WPHASE(blk, regs, { const auto r_ = RBASES(tid & 63);                                                \
      SSR_UNROLL for (int i = 0; i < SSR_W_P; ++i) R.v[i] = {R.tx[i], (L).re[RIDX(r_, i)]}; });           \
  }
#define SSR_W_RIDX_LD8(b_, i) ((b_).ld8 + ssr_w_off_ld8(i))
#define SSR_W_EXCHANGE(blk, regs, L, WB, WO, RB, RO, EXTRA) SSR_W_EXCHANGE_G(blk, regs, L, WB, WO, ssr_wave_bases, SSR_W_RIDX_##RB##_##RO, EXTRA)
#define SSR_W_RIDX_ld8_ssr_w_off_ld8(b_, i) SSR_W_RIDX_LD8(b_, i)

// ---- PAIRED output order (k_stft_wave, round 6): the third pass's butterfly j emits Z[j + 256 q] and a bin's partner
// Z[2048 - k] = Z[(256 - j) + 256 (7 - q)] comes out of butterfly 256 - j.  A lane that runs BOTH butterflies of such a pair holds
// every (k, 2048 - k) couple it needs in its own registers: the upper halves no longer cross lanes through LDS (one exchange
// of 32 stores + 32 loads and one dependent LDS round trip per frame pair less; the same arithmetic on the same operands:
// magnitudes bit-identical to the ascending order).  Lane l runs j = l, 64 + l, 192 - l, 256 - l (registers 8 b + q, b = 0..3);
// j = 0 and j = 128 pair with themselves and both sit in lane 0, whose four butterflies are 0, 64, 192, 128.
//   partner of register (b, q), q < 4:  lane > 0: (3 - b, 7 - q);   lane 0: b = 0 -> (0, 8 - q) (q = 0: itself, q = 4: Nyquist,
//   itself), b = 3 -> (3, 7 - q), b = 1 <-> 2 as everywhere.
struct SsrWavePBase { int ld8, p2, p3; };
SSR_DEV int ssr_wave_pj3(int tid) { return tid == 0 ? 128 : 256 - tid; }
SSR_DEV SsrWavePBase ssr_wave_pbases(int tid) {
  return {tid + (tid >> 5), ssr_wpad(192 - tid), ssr_wpad(ssr_wave_pj3(tid))};
}
SSR_DEV int ssr_w_rd_paired(const SsrWavePBase& B, int i) {
  const int b = i >> 3, q = i & 7;
  return (b == 0 ? B.ld8 : b == 1 ? B.ld8 + 66 : b == 2 ? B.p2 : B.p3) + 264 * q;
}
#define SSR_W_RIDX_PAIRED(b_, i) ssr_w_rd_paired(b_, i)
// the lane-ordered copies [N + 224 + 64 (3 b + m) + l] = w^((l + 64 b) 2^m): butterfly 192 - l is copy b = 2 at lane 64 - l
// (lane 0: copy 3, lane 0), butterfly 256 - l copy b = 3 at lane 64 - l (lane 0, j = 128: copy 2, lane 0)
#define SSR_W_LOAD_TW2_PAIRED { const int t_ = tid & 63; const unsigned l_ = SSR_UIDX(t_);                       \
    const unsigned o2_ = SSR_UIDX(t_ == 0 ? 192 : 64 - t_); const unsigned o3_ = SSR_UIDX(t_ == 0 ? -192 : 64 - t_); \
    SSR_UNROLL for (int i = 0; i < 6; ++i) R.tw2[i] = VT.at(l_ + (SSR_W_N + 224 + 64 * i));                       \
    SSR_UNROLL for (int i = 6; i < 9; ++i) R.tw2[i] = VT.at(o2_ + (SSR_W_N + 224 + 64 * i));                      \
    SSR_UNROLL for (int i = 9; i < 12; ++i) R.tw2[i] = VT.at(o3_ + (SSR_W_N + 224 + 64 * i)); }

// The rest of the transform after the in-register radix-32 pass (ssr_dft32 applied to R.v): exchange, radix-8 pass with the
// lane's seven twiddles, exchange, radix-8 pass with table twiddles.  On exit register 8 b + q holds Z[tid + 64 b + 256 q].
// BLK0: the untouched block descriptor (a fresh opaque lane index per stage: addresses are formed where they are used).
// The lane index is `tid & 63` throughout: a workgroup of several autonomous waves (ssr_stft_rn_wave.h) passes an L whose
// arrays are the calling wave's own.
// EXTRA2: further table loads to issue with the last pass's twiddles (the low-pass kernel's synthesis window).
// (VT: a view of the plan's twiddle table INCLUDING the lane-ordered copies behind it, SSR_W_N + SSR_W_TWP entries -
// ssr_tables.h: every load below is one contiguous run of 32 / 64 table entries)
#define SSR_W_LOAD_TW1 { const unsigned l_ = SSR_UIDX(tid & 31); \
                         SSR_UNROLL for (int q = 1; q < 8; ++q) R.tw1[q - 1] = VT.at(l_ + (SSR_W_N + 32 * (q - 1))); }
#define SSR_W_LOAD_TW2 { const unsigned l_ = SSR_UIDX(tid & 63); \
                         SSR_UNROLL for (int i = 0; i < 12; ++i) R.tw2[i] = VT.at(l_ + (SSR_W_N + 224 + 64 * i)); }
#define SSR_W_FFT_TAIL(blk, BLK0, regs, L, EXTRA2)                                                                        \
  SSR_W_FFT_TAIL_X(blk, BLK0, regs, L, SSR_W_EXCHANGE(blk, regs, L, st1, ssr_w_off_st1, ld8, ssr_w_off_ld8, SSR_W_LOAD_TW2 EXTRA2))
#define SSR_W_FFT_TAIL_PAIRED(blk, BLK0, regs, L)                                                                         \
  SSR_W_FFT_TAIL_X(blk, BLK0, regs, L, SSR_W_EXCHANGE_G(blk, regs, L, st1, ssr_w_off_st1, ssr_wave_pbases, SSR_W_RIDX_PAIRED, SSR_W_LOAD_TW2_PAIRED))
#define SSR_W_FFT_TAIL_X(blk, BLK0, regs, L, EXCH2)                                                                       \
  blk = BLK0; ssr_launder(blk);                                                                                     \
  SSR_W_EXCHANGE(blk, regs, L, st0, ssr_w_off_st0, ld8, ssr_w_off_ld8, SSR_W_LOAD_TW1);                             \
  /* pass 1: four radix-8 butterflies, twiddles w^(8 (j mod 32) q) - the same seven for every butterfly of the lane */ \
  SSR_WPHASE(blk, regs, {                                                                                           \
    SSR_UNROLL for (int b = 0; b < 4; ++b) {                                                                        \
      SSR_UNROLL for (int q = 1; q < 8; ++q) R.v[8 * b + q] = cmul(R.v[8 * b + q], R.tw1[q - 1]);                   \
      ssr_bfly8(R.v + 8 * b);                                                                                       \
    }                                                                                                               \
    SSR_CLK(2);                                                                                                     \
  });                                                                                                               \
  blk = BLK0; ssr_launder(blk);                                                                                     \
  EXCH2;                                                                                                            \
  /* pass 2: four radix-8 butterflies, twiddles w^(j q), j = tid + 64 b: three table values + four products each */  \
  SSR_WPHASE(blk, regs, {                                                                                           \
    SSR_UNROLL for (int b = 0; b < 4; ++b) {                                                                        \
      cx<T>* x = R.v + 8 * b;                                                                                       \
      const cx<T> w1 = R.tw2[3 * b], w2 = R.tw2[3 * b + 1], w4 = R.tw2[3 * b + 2];                                  \
      x[1] = cmul(x[1], w1);                                                                                        \
      x[2] = cmul(x[2], w2);                                                                                        \
      x[4] = cmul(x[4], w4);                                                                                        \
      const cx<T> w3 = cmul(w1, w2);                                                                                \
      x[3] = cmul(x[3], w3);                                                                                        \
      x[5] = cmul(x[5], cmul(w1, w4));                                                                              \
      x[6] = cmul(x[6], cmul(w2, w4));                                                                              \
      x[7] = cmul(x[7], cmul(w3, w4));                                                                              \
      ssr_bfly8(x);                                                                                                 \
    }                                                                                                               \
    SSR_CLK(3);                                                                                                     \
  });

// Request unit u's samples into the prefetch registers (branch-free, always valid addresses, reflection only at the ends).
// WHICH: 1 = the first signal's, 2 = the second's, 3 = both.  A wave can have 63 vector-memory instructions in flight (vmcnt is
// six bits); the 64th waits at issue for the oldest to return, so the body requests the two signals at different times.
template <typename T, int WHICH = 3, typename REGS>
SSR_DEV void ssr_wave_prefetch(const SsrStftParams<T>& p, REGS& R, int tid, const SsrView<float>& va, const SsrView<float>& vb,
                               int u, int n, int n_frames) {
  const int t_c = (u < n_frames) ? u : n_frames - 1;
  const int base = t_c * p.hop - SSR_W_N / 2;
  if (base >= 0 && base + SSR_W_N <= n) {            // wave-uniform: the frame lies fully inside the signal
    SSR_UNROLL for (int r = 0; r < SSR_W_P; ++r) {
      if (WHICH & 1) R.pa[r] = va.at(SSR_UIDX(tid + 64 * r), base);
      if (WHICH & 2) R.pb[r] = vb.at(SSR_UIDX(tid + 64 * r), base);
    }
  } else {
    SSR_UNROLL for (int r = 0; r < SSR_W_P; ++r) {
      const unsigned m = SSR_UIDX(ssr_reflect(base + tid + 64 * r, n));
      if (WHICH & 1) R.pa[r] = va.at(m);
      if (WHICH & 2) R.pb[r] = vb.at(m);
    }
  }
}

// Silent-frame votes of the unit whose samples sit in the prefetch registers, into flag set `par` (the body calls it at
// the top of the frame that consumes them; cf. ssr_stft_prefetched_flags).
template <typename REGS> SSR_DEV void ssr_wave_flags(REGS& R, int tid, int* nz, int par) {
  SSR_UNROLL for (int r = 0; r < SSR_W_P; ++r) { ssr_touch(R.pa[r]); ssr_touch(R.pb[r]); }
  unsigned ora = 0u, orb = 0u;
  SSR_UNROLL for (int r = 1; r < SSR_W_P; ++r) { ora |= ssr_mag_bits(R.pa[r]); orb |= ssr_mag_bits(R.pb[r]); }
  ora |= (tid == 0) ? 0u : ssr_mag_bits(R.pa[0]);    // sample m = 0 carries window weight exactly 0
  orb |= (tid == 0) ? 0u : ssr_mag_bits(R.pb[0]);
  SSR_WAVE_ANY_STORE(tid, ora != 0u, nz + par);
  SSR_WAVE_ANY_STORE(tid, orb != 0u, nz + 2 + par);
}

// grid = n_items * n_chunks workgroups of ONE wave; PAIR mode, direct 2048-point engine, float32 signals.
// MAG: 1 / 0 = magnitude rows are / are not written (p.out_kind == SSR_OUT_MAG known at compile time), -1 = decided at run time.
template <typename T, bool SUMS, bool SPLIT, int MAG = -1, typename BLK>
SSR_BODY void ssr_stft_wave_body(const SsrStftParams<T>& p, BLK& blk, int chunk, int item, char* lds_base) {
  constexpr int N = SSR_W_N, F = N / 2 + 1;
  constexpr bool PAIRED = SSR_WAVE_PAIRED != 0;
  // Developer variant (round 6, -DSSR_WAVE_ROWS_VIA_LDS=1; NOT the product): the two magnitude rows of a frame leave as 2 x 4 aligned
  // 16-byte stores per lane instead of 2 x 16 dword stores - laid out in bin order in the wave's exchange array (free between the last
  // pass and the next frame's first exchange; the DS operations of a wave execute in order) and read back four consecutive bins per
  // lane.  The premise: the address unit is busy 60-67 % of this kernel and its address FIFO fills 16x as often as in the variant that
  // stores no rows (profiles/r06_stft_wave_vmem_path.txt).  Measured: 136 -> 112 vector-memory instructions per frame pair, the same
  // time (2.34 ms alternating on one box, three rounds; same magnitudes, 84 tests) - the 0.5 ms the rows cost is not instruction count.
  constexpr bool ROWS_VIA_LDS = PAIRED && SSR_WAVE_ROWS_VIA_LDS != 0;
  using Regs = SsrWaveRegs<T, SUMS>;
  SsrWaveLds<T, SPLIT> L(lds_base);
  const int n = p.len[item], hop = p.hop;
  const int n_frames = ssr_num_frames_dev(n, N, hop);
  // Units of this chunk: u0, u0 + S, u0 + 2 S, ... (< u1).  S = 1: consecutive frames.  S > 1: the S chunks of a group
  // interleave over the group's span of S * units_per_chunk frames - they run at the same time on the same XCD (k_stft_wave's
  // block mapping), so the 75 % overlap of neighbouring frames is found in that XCD's L2 instead of being fetched again by the
  // same wave 13 us later, when it has long been evicted (2048 waves x 16 KB of live signal = the whole L2).
  const int S = p.interleave > 1 ? p.interleave : 1;
  const int span0 = (chunk / S) * S * p.units_per_chunk;
  const int u0 = span0 + chunk % S;
  const int u1 = (span0 + S * p.units_per_chunk < n_frames) ? span0 + S * p.units_per_chunk : n_frames;
  const float* sa = p.a + p.a_off[item];
  const float* sb = p.b + p.b_off[item];
  const int64_t row0 = p.frame_off[item];
  double* part = p.part ? p.part + ((int64_t)item * p.n_chunks + chunk) * SSR_NPART : nullptr;
  // the variant without running sums is only launched when no SISpec bit is set: say so, and their code disappears
  const int mask = SUMS ? p.metric_mask : (p.metric_mask & SSR_M_LSD);
  const bool want_lsd = mask & SSR_M_LSD;
  const SsrView<float> va(sa, n), vb(sb, n);
  const SsrView<T> vw(p.window, N);
  const SsrView<cx<T>> vt(p.tw, N + SSR_W_TWP);

  double* lsum = reinterpret_cast<double*>(lds_base + ssr_stft_wave_sums_offset<T, SPLIT>());   // [6][64], SUMS only
  SSR_REGS(Regs, regs, blk);
  SSR_WPHASE(blk, regs, {
    R.lsd_total = 0.0;
    if constexpr (SUMS) for (int q = 0; q < 6; ++q) lsum[64 * q + tid] = 0.0;
    if (tid == 0) L.sc1[0] = 0.0;
    if (u0 < u1) {
      ssr_wave_prefetch<T>(p, R, tid, va, vb, u0, n, n_frames);
      SSR_UNROLL for (int r = 0; r < SSR_W_P / 2; ++r) R.wl[r] = vw.at(SSR_UIDX(tid + 64 * r));
    }
    SSR_VMEM_DRAIN();
  });

  BLK blk0 = blk;
#ifdef SSR_CLK_NOW
#undef SSR_CLK
#define SSR_CLK(i) SSR_CLK_NOW(i)
  unsigned long long clk_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, clk_sum[5] = {0, 0, 0, 0, 0}, clk_top[2] = {0, 0}, clk_frames = 0;
#endif
  for (int u = u0, it = 0; u < u1; u += S, ++it) {
    // The lane's table values (window, twiddles) and addresses are loop-invariant, and 128 + 64 registers of data and
    // prefetched samples leave no room to keep them: with the lane index opaque the optimiser cannot hoist them out of the
    // loop (it would, and then shuffle some 250 values through the accumulator registers every frame).  Table values are
    // re-requested from L1 / L2 a phase before they are needed; LDS slots are four per-lane bases plus immediates.
    blk = blk0; ssr_launder(blk);
    // ---- pass 0: window, radix-32 DFT in registers (Stockham pass with stride 1: no twiddle).  Lane 0 also closes the
    // previous frame's LSD.
    SSR_WPHASE(blk, regs, {
      SSR_CLK(0);
#ifdef SSR_WAVE_NOPF   /* developer experiment: no frame-ahead requests - the unit's samples and window are loaded here (occupancy study) */
      ssr_wave_prefetch<T>(p, R, tid, va, vb, u, n, n_frames);
      SSR_UNROLL for (int r = 0; r < SSR_W_P / 2; ++r) R.wl[r] = vw.at(SSR_UIDX(tid + 64 * r));
#endif
      ssr_wave_flags(R, tid, L.nz, it & 1);            // silent-frame votes of this unit, read by its epilogue
      SSR_CLK(6);
      SSR_UNROLL for (int r = 0; r < SSR_W_P; ++r) {
        const T w = (r < SSR_W_P / 2) ? R.wl[r] : (T)0.5 - R.wl[r - SSR_W_P / 2];       // w[m + N/2] = 1/2 - w[m]
        R.v[r] = {(T)R.pa[r] * w, (T)R.pb[r] * w};
      }
      SSR_CLK(7);
      ssr_dft32(R.v);
      if (want_lsd && it > 0 && tid == 0) R.lsd_total += sqrt(L.sc1[0] / (double)F);
      SSR_CLK(1);
    });
#define VT vt
    if constexpr (PAIRED) { SSR_W_FFT_TAIL_PAIRED(blk, blk0, regs, L); } else { SSR_W_FFT_TAIL(blk, blk0, regs, L, ); }
#undef VT
    // PAIRED: register 8 b + q holds Z[j_b + 256 q], j = tid, 64 + tid, 192 - tid, 256 - tid (lane 0: 0, 64, 192, 128) - every
    // partner Z[2048 - k] of the lane's bins is in the lane's own registers (see SSR_W_FFT_TAIL_PAIRED).
    // Ascending order: register 8 b + q holds Z[k], k = tid + 64 b + 256 q.  The bins k <= 1024 are this lane's to emit; each
    // needs Z[2048 - k], which lives in the upper half (q >= 4) of lane 64 - tid: the upper halves go through LDS once.
    if constexpr (PAIRED) {
    } else if constexpr (!SPLIT) {
      SSR_WPHASE(blk, regs, { const int w_ = ssr_wave_bases(tid).ld8;
        SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q = 4; q < 8; ++q) {
          L.re[w_ + 66 * b + 264 * (q - 4)] = R.v[8 * b + q].x; L.im[w_ + 66 * b + 264 * (q - 4)] = R.v[8 * b + q].y; } });
    } else {
      SSR_WPHASE(blk, regs, { const int w_ = ssr_wave_bases(tid).ld8;
        SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q = 4; q < 8; ++q) {
          L.re[w_ + 66 * b + 264 * (q - 4)] = R.v[8 * b + q].x;
          L.re[SSR_W_IMOFF + w_ + 66 * b + 264 * (q - 4)] = R.v[8 * b + q].y; } });
    }

    // ---- epilogue: four groups of four bins (partner values in, magnitudes out per group: few registers live at once)
    blk = blk0; ssr_launder(blk);
    const int64_t OP = p.out_pitch ? p.out_pitch : F;      // floats between output rows
    float* ra0 = p.out_a ? p.out_a + (row0 + u) * OP : nullptr;   // wave-uniform row pointers
    float* rb0 = p.out_b ? p.out_b + (row0 + u) * OP : nullptr;
    SSR_WPHASE(blk, regs, {
      // UNCONDITIONAL (the last frame of a chunk re-requests a clamped, valid frame that nobody consumes): under a condition
      // the previous contents of the 64 + 32 registers would stay live through all three passes for the path not taken
      // The first signal and the window now (32 + 16 requests), the second signal half-way through the bins: 64 + 16 at once
      // stopped the wave at the 64th request until the oldest ones had returned, i.e. exposed the latency it is here to hide.
      // PAIRED: all 128 data registers are live here, and a quarter of them is released per butterfly group - the first
      // signal is requested after group 0, the second after group 1, the window (the shortest way: L1 / L2) after group 2
      if constexpr (!PAIRED) {
        ssr_wave_prefetch<T, 1>(p, R, tid, va, vb, u + S, n, n_frames);
        SSR_UNROLL for (int r = 0; r < SSR_W_P / 2; ++r) R.wl[r] = vw.at(SSR_UIDX(tid + 64 * r));
      }
      SSR_SCHED_BARRIER();
      SSR_CLK(4);
      double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const int par = it & 1;
      const bool a_nz = L.nonzero(0, par), b_nz = L.nonzero(1, par);
      const bool both = a_nz && b_nz;                             // wave-uniform: the common case carries no selects
      const int im_off = SPLIT ? SSR_W_IMOFF : 0;
      const T* lre = L.re;
      const T* lim = SPLIT ? L.re : L.im;
      const bool store = MAG < 0 ? p.out_kind == SSR_OUT_MAG : MAG != 0;
      // the two magnitude rows as buffer views: scalar row base + one lane offset + an immediate per bin (a plain pointer
      // costs a 64-bit vector add per store)
      const SsrRwView<float> wa(ra0, store ? F : 0), wb(rb0, (store && rb0 != nullptr) ? F : 0);   // out_b == null: the target rows
      // are not written (ssr_pair_metrics_multi: they exist already) - the stores fall to the buffer range check
      const int lane4 = 4 * tid;
      char* mrows = reinterpret_cast<char*>(L.re);                 // ROWS_VIA_LDS: [est row | target row], SSR_W_ROWB bytes apart
      // byte offset of bin j_b + 256 q in a magnitude row: lane part + immediate
      const int lane4_2 = PAIRED ? 4 * (192 - tid) : 0, lane4_3 = PAIRED ? 4 * ssr_wave_pj3(tid) : 0;
      auto bin_off = [&](int b, int q) -> int {
        if constexpr (!PAIRED) return lane4 + 4 * (64 * b + 256 * q);
        else return (b == 0 ? lane4 : b == 1 ? lane4 + 256 : b == 2 ? lane4_2 : lane4_3) + 1024 * q;
      };
      const int pr = PAIRED ? 0 : ssr_wave_bases(tid).pr;
      // Lane 0's butterflies 0 and 128 pair with THEMSELVES (q <-> 8 - q, bin 0 and the Nyquist bin with themselves; q <-> 7 - q):
      // its upper halves are rotated once into the registers the general rule (3 - b, 7 - q) reads.  Register 4 - the Nyquist bin
      // Z[1024], which pairs with itself - is taken right after group 0 (32 data registers are free by then), and only then
      // receives its rotated value.
      cx<T> rot4 = {(T)0, (T)0};
      if constexpr (PAIRED) {
        if (tid == 0) {
          rot4 = R.v[28];
          const cx<T> t29 = R.v[29], t30 = R.v[30], t31 = R.v[31];
          R.v[31] = R.v[0]; R.v[30] = R.v[7]; R.v[29] = R.v[6]; R.v[28] = R.v[5];
          R.v[7] = t31; R.v[6] = t30; R.v[5] = t29;
        }
      }
#ifdef SSR_WAVE_G                                                   /* developer builds: bins in flight */
      constexpr int G = SSR_WAVE_G;
#else
      constexpr int G = (SUMS || PAIRED) ? 2 : 4;                  // bins in flight (the variant with running sums is tighter; PAIRED: the
                                                                  // partners are in registers - there is no LDS latency to cover)
#endif
      // The sixteen bins, two at a time.  FAST (a compile-time fact inside each copy of the loop): both frames hold signal and
      // the mask is the variant's full set - no zero forcing, no per-bin test of the mask, and the float32 arithmetic of a
      // bin pair runs as packed instructions (ssr_pair_bins2_fast).  The wave-uniform choice is made ONCE per frame, outside
      // the loop: taken per bin it split the epilogue into 48 basic blocks with two scalar branches each.
      constexpr int PF1 = SUMS ? SSR_WAVE_PF1_SUMS : SSR_WAVE_PF1;    // PAIRED: the butterfly group after which the first signal is requested (then the second, then the window)
      auto bins = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        SSR_UNROLL for (int b = 0; b < 4; ++b) SSR_UNROLL for (int q0 = 0; q0 < 4; q0 += G) {
          cx<T> zn[G];
          if constexpr (PAIRED) {
            SSR_UNROLL for (int q = 0; q < G; ++q) zn[q] = R.v[8 * (3 - b) + 7 - (q0 + q)];   // Z[2048 - k] out of butterfly 256 - j
          } else {
            SSR_UNROLL for (int q = 0; q < G; ++q)                    // Z[2048 - k] sits at upper-half slot 1024 - k
              zn[q] = {lre[pr - 66 * b - 264 * (q0 + q)], lim[im_off + pr - 66 * b - 264 * (q0 + q)]};
            if (b == 0 && q0 == 0 && tid == 0) zn[0] = R.v[0];       // bin 0 pairs with itself
          }
          SSR_UNROLL for (int q = 0; q < G; q += 2) {
            const cx<T> zk0 = R.v[8 * b + q0 + q], zk1 = R.v[8 * b + q0 + q + 1];
            f2 e, t;
            if constexpr (FAST) {
              ssr_pair_bins2_fast<T, SUMS>(acc, zk0, zn[q], zk1, zn[q + 1], e, t);
            } else {
              float e0, t0, e1, t1;
              ssr_pair_bin<T, 0, true>(mask, acc, zk0, zn[q], a_nz, b_nz, e0, t0);
              ssr_pair_bin<T, 0, true>(mask, acc, zk1, zn[q + 1], a_nz, b_nz, e1, t1);
              e = f2_make(e0, e1); t = f2_make(t0, t1);
            }
            if (store) {
              if constexpr (ROWS_VIA_LDS) {                          // the two rows in bin order in the exchange array (free until the next exchange)
                *reinterpret_cast<float*>(mrows + bin_off(b, q0 + q)) = e.x;
                *reinterpret_cast<float*>(mrows + bin_off(b, q0 + q + 1)) = e.y;
                *reinterpret_cast<float*>(mrows + SSR_W_ROWB + bin_off(b, q0 + q)) = t.x;
                *reinterpret_cast<float*>(mrows + SSR_W_ROWB + bin_off(b, q0 + q + 1)) = t.y;
              } else {
#ifdef SSR_WAVE_ROWS_NT                                              /* developer build: the rows as streaming (nt) stores */
                wa.st_raw_nt(bin_off(b, q0 + q), e.x);
                wa.st_raw_nt(bin_off(b, q0 + q + 1), e.y);
                wb.st_raw_nt(bin_off(b, q0 + q), t.x);
                wb.st_raw_nt(bin_off(b, q0 + q + 1), t.y);
#else
                wa.st_raw(bin_off(b, q0 + q), e.x);
                wa.st_raw(bin_off(b, q0 + q + 1), e.y);
                wb.st_raw(bin_off(b, q0 + q), t.x);
                wb.st_raw(bin_off(b, q0 + q + 1), t.y);
#endif
              }
            }
          }
          if (PAIRED && b == 0 && q0 + G == 4) {
            if (tid == 0) {                                           // the Nyquist bin, then register 4's rotated value
              const cx<T> zq = R.v[4];
              float e, t;
              ssr_pair_bin<T, 0, true>(mask, acc, zq, zq, a_nz, b_nz, e, t);
              if (store) { wa.st_raw(4 * (N / 2), e); wb.st_raw(4 * (N / 2), t); }
              R.v[4] = rot4;
            }
          }
          if ((PAIRED ? b == PF1 : false) && q0 + G == 4) {
            SSR_WAVE_EPI_SB();
            SSR_WAVE_PF(ssr_wave_prefetch<T, 1>(p, R, tid, va, vb, u + S, n, n_frames));
            SSR_WAVE_EPI_SB();
          }
          if (b == (PAIRED ? PF1 + 1 : 1) && q0 + G == 4) {
            SSR_WAVE_EPI_SB();
            SSR_WAVE_PF(ssr_wave_prefetch<T, 2>(p, R, tid, va, vb, u + S, n, n_frames));
            SSR_WAVE_EPI_SB();
          }
          if ((PAIRED ? b == PF1 + 2 : false) && q0 + G == 4) {
            SSR_WAVE_EPI_SB();
            SSR_WAVE_PF(SSR_UNROLL for (int r = 0; r < SSR_W_P / 2; ++r) R.wl[r] = vw.at(SSR_UIDX(tid + 64 * r)));
            SSR_WAVE_EPI_SB();
          }
        }
        if (PAIRED && PF1 + 2 > 3) {                              // (the window behind the last group)
          SSR_WAVE_EPI_SB();
          SSR_WAVE_PF(SSR_UNROLL for (int r = 0; r < SSR_W_P / 2; ++r) R.wl[r] = vw.at(SSR_UIDX(tid + 64 * r)));
          SSR_WAVE_EPI_SB();
        }
      };
      constexpr int FULL = SUMS ? (SSR_M_LSD | SSR_M_LOG_SISPEC | SSR_M_SISPEC) : SSR_M_LSD;
      if (both && (mask & 7) == FULL) bins(SsrTrue{});
      else bins(SsrFalse{});
      if (!PAIRED && tid == 0) {                                  // the Nyquist bin: Z[1024] pairs with itself
        const cx<T> zq = {lre[0], lim[im_off]};
        float e, t;
        ssr_pair_bin<T, 0, true>(mask, acc, zq, zq, a_nz, b_nz, e, t);
        if (store) { wa.st_raw(4 * (N / 2), e); wb.st_raw(4 * (N / 2), t); }
      }
      if (want_lsd) SSR_WAVE_SUM_STORE(tid, 64, acc[0], L.sc1);
      if constexpr (SUMS)
        for (int q = 0; q < 6; ++q) SSR_LDS_ACCUM(lsum + 64 * q + tid, acc[1 + q]);
      SSR_CLK(5);
    });
    if constexpr (ROWS_VIA_LDS) {
      SSR_WPHASE(blk, regs, {
        const bool store = MAG < 0 ? p.out_kind == SSR_OUT_MAG : MAG != 0;
        if (store) {                                                // (wave-uniform)
          const SsrRwView<float> wa(ra0, F), wb(rb0, rb0 != nullptr ? F : 0);
          const char* mrows = reinterpret_cast<const char*>(L.re);
          SSR_UNROLL for (int j = 0; j < 4; ++j) {                  // bins 4 tid + 256 j .. + 3 (bin 1024 went out with lane 0's own store)
            const int off = 16 * tid + 1024 * j;
            const float* ea = reinterpret_cast<const float*>(mrows + off);
            const float* ta = reinterpret_cast<const float*>(mrows + SSR_W_ROWB + off);
            const float e0 = ea[0], e1 = ea[1], e2 = ea[2], e3 = ea[3], t0 = ta[0], t1 = ta[1], t2 = ta[2], t3 = ta[3];
            wa.st_raw4(off, e0, e1, e2, e3);
            wb.st_raw4(off, t0, t1, t2, t3);
          }
        }
      });
    }
#ifdef SSR_CLK_NOW
    for (int q = 0; q < 5; ++q) clk_sum[q] += clk_[q + 1] - clk_[q];
    clk_top[0] += clk_[6] - clk_[0]; clk_top[1] += clk_[7] - clk_[6];
    ++clk_frames;
#endif
  }
#ifdef SSR_CLK_NOW
#undef SSR_CLK
#define SSR_CLK(i)
  if (blk.tid == 0) {
    for (int q = 0; q < 5; ++q) atomicAdd(&ssr_dbg_clk[q], clk_sum[q]);
    atomicAdd(&ssr_dbg_clk[5], clk_frames);
    atomicAdd(&ssr_dbg_clk[6], clk_top[0]); atomicAdd(&ssr_dbg_clk[7], clk_top[1]);
  }
#endif

  if (part == nullptr) return;
  // ---- chunk tail: last frame's LSD and the wave-sums of the SISpec accumulators
  if constexpr (SUMS) {
    for (int q = 0; q < 6; ++q) {
      SSR_WPHASE(blk, regs, SSR_WAVE_SUM_STORE(tid, 64, lsum[64 * q + tid], L.sc1 + 1));
      SSR_WPHASE(blk, regs, if (tid == 0) part[1 + q] = L.sc1[1]);
    }
  } else {
    SSR_WPHASE(blk, regs, if (tid == 0) for (int q = 0; q < 6; ++q) part[1 + q] = 0.0);
  }
  SSR_WPHASE(blk, regs, if (tid == 0) {
    double lsd = R.lsd_total;
    if (want_lsd && u1 > u0) lsd += sqrt(L.sc1[0] / (double)F);
    part[0] = lsd;
    part[7] = 0.0;
  });
}
