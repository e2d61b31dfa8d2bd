// Kernel bodies K3-K5 on materialised [T, F] float32 spectrograms: SSIM (7x7 uniform window),
// LSD / SISpec / log-SISpec reductions, and the per-item finalisation of the partial records.
//
// SSIM semantics (ssr_eval/metrics.py:123-132 -> skimage.metrics.structural_similarity(win_size=7),
// float images, no data_range): data_range = 2 -> C1 = 4e-4, C2 = 3.6e-3; uniform 7x7 window; sample
// covariance (49/48); only windows that lie fully inside the image survive the 3-pixel border crop, so
// exactly the (T-6) x (F-6) valid windows are evaluated; float64 moments; mean in float64.
#pragma once
#include "ssr_stft.h"

#define SSR_SSIM_NT 64   // ONE wave per workgroup: the row loop's barriers cost nothing and waves run independently
#define SSR_SSIM_MAXCPT 6  // contiguous output columns per thread (template parameter CPT = 1..6)
#define SSR_SSIM_WIN 7

// outputs per strip for a given CPT; a strip reads SSR_SSIM_NT*CPT + 6 input columns
SSR_HD constexpr int ssr_ssim_strip_out(int cpt) { return SSR_SSIM_NT * cpt; }
SSR_HD int ssr_ssim_pick_cpt(int n_bins) {
  // contiguous outputs per thread: the candidate in {4, 5, 6} that wastes the fewest lanes in the last strip
  const int outs = n_bins - (SSR_SSIM_WIN - 1);
  if (outs <= SSR_SSIM_NT * 4) {
    int c = (outs + SSR_SSIM_NT - 1) / SSR_SSIM_NT;
    return c < 1 ? 1 : c;
  }
  int best = 4, best_waste = 1 << 30;
  for (int c = 4; c <= SSR_SSIM_MAXCPT; ++c) {
    const int w = SSR_SSIM_NT * c;
    const int waste = ((outs + w - 1) / w) * w - outs;
    if (waste < best_waste) { best_waste = waste; best = c; }
  }
  return best;
}

struct SsrSsimParams {
  const float* x;            // est spectrograms  [rows, F]
  const float* y;            // target spectrograms
  const int64_t* frame_off;  // [n_items] first row of item i
  const int32_t* n_rows;     // [n_items] T_i
  int F;
  int rows_per_tile, n_row_tiles, n_strips;
  double* part;              // [n_items, n_row_tiles * n_strips] sum of S over the tile
  int pitch;                 // floats between rows (0: F).  A multiple of 4 on a 16-byte-aligned base selects CONTIG.
  // virtual items (ssr_pair_metrics_multi: K estimates per target): item v = k * vi_n + i reads estimate plane k (x + k * x_plane)
  // against the ONE target image of item i; vi_n = 0: plain items
  int vi_n;
  int64_t x_plane;
};
SSR_DEV int ssr_ssim_pitch(const SsrSsimParams& p) { return p.pitch ? p.pitch : p.F; }

// Only four window sums are needed: S depends on vx and vy through vx + vy alone, so x^2 and y^2 are
// accumulated together:  q0 = sum x, q1 = sum y, q2 = sum (x^2 + y^2), q3 = sum x*y.
template <int CPT> struct SsrSsimRegs {
  double cs[CPT + 1][4];  // running 7-row sums for the thread's input columns (strided: tid + NT*i; CONTIG: ssr_ssim_col)
  double s;               // sum of S over the thread's outputs
  float px[2][4][CPT + 1];   // row values in flight, two row steps deep: [step parity][entering x, y, leaving x, y][column slot]
                             // (CONTIG requests the entering rows only: the leaving ones come from the two rings)
  float ring[SSR_SSIM_WIN][CPT + 1];   // CONTIG: the x pixels of the last seven rows, slot = row step mod 7 - the leaving x row is
                                       // not fetched again.  The y pixels of those rows sit in an LDS ring (SsrSsimLds::yring):
                                       // a second REGISTER ring is 172 VGPRs -> two waves per SIMD, or spills at three (round 2:
                                       // all measured slower than fetching y's leaving row again).
  double pxy[CPT], pb[CPT];   // CONTIG: the SSIM expression of the thread's outputs while the window sums arrive one quantity at a
  float den[CPT];             // time (pxy: sx, then sx sy; pb: sx^2 + sy^2; den: b1 b2 - ssr_ssim_stage1..3)
};

// Column-sum index in LDS.  The horizontal pass reads with a lane stride of CPT doubles; for even CPT that is a
// 4-way (CPT = 4) bank conflict on ds_read_b64 (measured: 60 % of the kernel's LDS cycles), so one spare slot
// is inserted every CPT columns to make the lane stride CPT + 1 (odd).  Odd CPT needs nothing.
template <int CPT> SSR_DEV int ssr_ssim_slot(int c) { return (CPT % 2 == 0) ? c + c / CPT : c; }

// CONTIG (round 3): the four column sums go through ONE array, a quantity at a time (write q, read q, write q + 1 ... - the DS
// operations of the single wave execute in order), which makes room for a seven-row ring of the y pixels (rows of RW floats:
// the thread's own four columns as one 16-byte slot at 4 tid, the strip's extra columns at 256 + tid) at twelve waves per CU:
// 7,392 + 2,648 + 704 B per wave against 11,296 B with four column arrays and no ring.
template <int CPT, bool CONTIG = false> struct SsrSsimLds {
  static constexpr int PW = SSR_SSIM_NT * CPT + 8 + ((CPT % 2 == 0) ? (SSR_SSIM_NT * CPT + 8) / CPT + 1 : 0);
  static constexpr int NQ = CONTIG ? 1 : 4;
  static constexpr int RW = SSR_SSIM_NT * CPT + 8;
  static constexpr size_t ring_bytes() { return CONTIG ? sizeof(float) * SSR_SSIM_WIN * RW : 0; }
  static constexpr size_t bytes() { return ring_bytes() + sizeof(double) * (NQ * PW + SSR_SSIM_NT + 16 + 8); }
  float* yring; double* col; double* sc0; double* sc1; double* res;
  SSR_MEMBER explicit SsrSsimLds(char* base) {
    yring = reinterpret_cast<float*>(base);
    col = reinterpret_cast<double*>(base + ring_bytes());
    sc0 = col + NQ * PW;
    sc1 = sc0 + SSR_SSIM_NT;
    res = sc1 + 16;
  }
};

// 16-byte slot of an LDS row (p 16-byte aligned)
SSR_DEV void ssr_ld4(const float* p, float* o) {
#ifdef SSR_HOST_EMU
  memcpy(o, p, 16);
#else
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = *reinterpret_cast<const f4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
#endif
}
SSR_DEV void ssr_st4(float* p, const float* v) {
#ifdef SSR_HOST_EMU
  memcpy(p, v, 16);
#else
  typedef float f4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<f4*>(p) = f4{v[0], v[1], v[2], v[3]};
#endif
}

// One [T, F] float32 image of the batch as the row loads see it.  Device: a raw buffer resource (scalar base +
// scalar row offset + 32-bit lane offset in ONE instruction - the compiler otherwise keeps twenty per-lane 64-bit
// pointers and advances each with a vector add every row); host emulation: a plain pointer.
struct SsrImage {
#ifdef SSR_HOST_EMU
  const float* base;
  SSR_MEMBER SsrImage(const float* p, int64_t) : base(p) {}
  SSR_MEMBER float at(int64_t row_elems, unsigned col) const { return base[row_elems + col]; }
  SSR_MEMBER void at4(int64_t row_elems, unsigned col, float* o) const { memcpy(o, base + row_elems + col, 16); }
#else
  __amdgpu_buffer_rsrc_t rsrc;
  SSR_MEMBER SsrImage(const float* p, int64_t n_elems)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(n_elems * 4), 0x00020000)) {}   // raw dwords
  SSR_MEMBER float at(int64_t row_elems, unsigned col) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(col * 4u), (int)(row_elems * 4), 0));
  }
  // four consecutive pixels, (row_elems + col) a multiple of 4 on a 16-byte-aligned image: one aligned 16-byte load
  SSR_MEMBER void at4(int64_t row_elems, unsigned col, float* o) const {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(col * 4u), (int)(row_elems * 4), 0));
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
#endif
};

// Which input column of the strip a thread's register slot i holds.  Strided (the general kernel): tid + NT i - every load
// instruction reads 64 consecutive pixels.  CONTIG (CPT = 4, rows 16-byte aligned): slots 0..3 are the thread's own four
// consecutive columns 4 tid .. 4 tid + 3 (ONE 16-byte load per image row; the wave reads 1 KB contiguous) and slot 4 is one of
// the strip's six extra columns, 256 + tid.  The thread's outputs are columns 4 tid .. 4 tid + 3 either way; under CONTIG
// their window sums take the first four column sums from the thread's own registers and only six from LDS.
template <int CPT, bool CONTIG> SSR_DEV int ssr_ssim_col(int tid, int i) {
  if constexpr (CONTIG) return (i < CPT) ? CPT * tid + i : SSR_SSIM_NT * CPT + tid;
  else return tid + SSR_SSIM_NT * i;
}

// Row step, split in two so that the loads of the NEXT step are in flight while the current one is consumed:
// ssr_ssim_row_load issues the loads of the row entering the 7-row window (row_add) and of the row leaving
// it (row_sub; re-reads row_add when there is none) with clamped, always valid column indices;
// ssr_ssim_row_apply folds the loaded values into the thread's running column sums.
template <int CPT, bool CONTIG, int SET>
SSR_DEV void ssr_ssim_row_load(const SsrSsimParams& p, SsrSsimRegs<CPT>& R, int tid, const SsrImage& x, const SsrImage& y,
                               int row_add, int row_sub, int c_in0, int ncol_in) {
  constexpr int VC = CPT + 1;
  const bool sub = row_sub >= 0;
  const int pitch = ssr_ssim_pitch(p);
  const int64_t ea = (int64_t)row_add * pitch + c_in0;                     // block-uniform element offsets of the rows
  const int64_t es = (int64_t)(sub ? row_sub : row_add) * pitch + c_in0;
  if constexpr (CONTIG) {
    static_assert(!CONTIG || CPT == 4 || CPT == 8, "four or eight consecutive columns per thread");
    // (a quad past the strip's last column - narrow last strip - is clamped to the last aligned quad: never used, see apply)
    const int last4 = (ncol_in - 1) & ~3;
    // (the leaving rows come from the rings - x: registers, y: LDS; R.px[SET][2..3] are not loaded)
#pragma unroll
    for (int h = 0; h < CPT / 4; ++h) {
      const int c0 = CPT * tid + 4 * h;
      const unsigned c4 = (unsigned)((c0 < last4) ? c0 : last4);
      float q[2][4];
      x.at4(ea, c4, q[0]); y.at4(ea, c4, q[1]);
#pragma unroll
      for (int i = 0; i < 4; ++i) { R.px[SET][0][4 * h + i] = q[0][i]; R.px[SET][1][4 * h + i] = q[1][i]; }
    }
    int ce = SSR_SSIM_NT * CPT + tid;
    if (ce >= ncol_in) ce = ncol_in - 1;
    R.px[SET][0][CPT] = x.at(ea, (unsigned)ce); R.px[SET][1][CPT] = y.at(ea, (unsigned)ce);
    return;
  }
#pragma unroll
  for (int i = 0; i < VC; ++i) {
    int c = tid + SSR_SSIM_NT * i;
    if (c >= ncol_in) c = ncol_in - 1;
    const unsigned uc = (unsigned)c;
    R.px[SET][0][i] = x.at(ea, uc);
    R.px[SET][1][i] = y.at(ea, uc);
    R.px[SET][2][i] = x.at(es, uc);
    R.px[SET][3][i] = y.at(es, uc);
  }
}

template <int CPT, bool CONTIG, int SET, int SLOT>
SSR_DEV void ssr_ssim_row_apply(SsrSsimRegs<CPT>& R, int tid, bool sub, int ncol_in) {
  constexpr int VC = CPT + 1;
#pragma unroll
  for (int i = 0; i < VC; ++i) {
    if (ssr_ssim_col<CPT, CONTIG>(tid, i) < ncol_in) {
      const double a = (double)R.px[SET][0][i], b = (double)R.px[SET][1][i];
      const double c = sub ? (double)R.px[SET][2][i] : 0.0, d = sub ? (double)R.px[SET][3][i] : 0.0;
      // every product goes into its running sum with one fused multiply-add (10 operations per column instead of 13)
      R.cs[i][0] = (R.cs[i][0] + a) - c;
      R.cs[i][1] = (R.cs[i][1] + b) - d;
      R.cs[i][2] = fma(-d, d, fma(-c, c, fma(b, b, fma(a, a, R.cs[i][2]))));
      R.cs[i][3] = fma(-c, d, fma(a, b, R.cs[i][3]));
    }
  }
}

// CONTIG row step: the entering row (R.px[SET][0..1], requested two steps ago) joins the running column sums and the row seven
// steps older leaves them - its x pixels from the register ring, its y pixels from the LDS ring (`yrow`: the ring row of slot
// step mod 7) - and takes their places in both rings.  No condition on the ring traffic: a thread past the strip's last column
// moves clamped, never-used values, and threads without an extra column share one spare slot.  No condition on "is there a leaving
// row yet" either: both rings start as zeros, and subtracting a zero row changes no bit of the sums.
template <int CPT, int SET, int SLOT>
SSR_DEV void ssr_ssim_row_apply_contig(SsrSsimRegs<CPT>& R, float* yrow, int tid) {
  constexpr int VC = CPT + 1;
  float* own = yrow + CPT * tid;
  float* ext = yrow + SSR_SSIM_NT * CPT + (tid < 7 ? tid : 7);
  float d[VC];
#pragma unroll
  for (int h = 0; h < CPT / 4; ++h) ssr_ld4(own + 4 * h, d + 4 * h);
  d[CPT] = *ext;
#pragma unroll
  for (int h = 0; h < CPT / 4; ++h) ssr_st4(own + 4 * h, R.px[SET][1] + 4 * h);
  *ext = R.px[SET][1][CPT];
  // (columns past the strip's end: the loads were clamped to valid pixels, their sums are formed and never used)
#pragma unroll
  for (int i = 0; i < VC; ++i) {
    const double a = (double)R.px[SET][0][i], b = (double)R.px[SET][1][i];
    const double c = (double)R.ring[SLOT][i], dd = (double)d[i];
    R.cs[i][0] = (R.cs[i][0] + a) - c;
    R.cs[i][1] = (R.cs[i][1] + b) - dd;
    R.cs[i][2] = fma(-dd, dd, fma(-c, c, fma(b, b, fma(a, a, R.cs[i][2]))));
    R.cs[i][3] = fma(-c, dd, fma(a, b, R.cs[i][3]));
    R.ring[SLOT][i] = R.px[SET][0][i];
  }
}

// S for one window from the four RAW 49-pixel sums.  Every factor of the skimage expression is scaled by
// n^2 = 49^2 (numerator and denominator alike), which removes the divisions by n:
//   A1 = 2 ux uy + C1           -> 2 sx sy + C1 n^2
//   A2 = 2 cov (uxy - ux uy) + C2 -> 2 cov (n sxy - sx sy) + C2 n^2
//   B1 = ux^2 + uy^2 + C1       -> sx^2 + sy^2 + C1 n^2
//   B2 = cov (uxx - ux^2 + uyy - uy^2) + C2 -> cov (n sq - (sx^2 + sy^2)) + C2 n^2
// Evaluated in three stages, so that a caller which forms the window sums one quantity at a time can fold each into the expression
// as soon as it exists (two live values per output instead of four).  float64 exactly where the cancellation lives - the products
// sx sy, sx^2 + sy^2 and the two differences n sxy - sx sy, n sq - (sx^2 + sy^2) - and float32 for everything after it: the four
// factors are sums of positive terms of O(1) relative accuracy 6e-8, the ratio is formed with the hardware reciprocal
// (v_rcp_f32, 1 ulp).  ~3e-7 per pixel, unbiased, against a 1e-5 bar on the MEAN of ~4e5 pixels (measured on the test vectors:
// < 2e-7 on the mean).  Round 3 moved the factors a1, a2, b1, b2 themselves to float32 (they were float64 up to the two
// products a1 a2 and b1 b2): 5 float64 operations per output instead of 10 - the kernel's time follows its FP64 operation count.
struct SsrSsimK {
  static constexpr double n = 49.0;
  static constexpr float cov = (float)(49.0 / 48.0);
  static constexpr float C1n = (float)((0.01 * 2.0) * (0.01 * 2.0) * 49.0 * 49.0), C2n = (float)((0.03 * 2.0) * (0.03 * 2.0) * 49.0 * 49.0);
};
SSR_DEV void ssr_ssim_stage1(double sx, double sy, double& pxy, double& pp) { pxy = sx * sy; pp = fma(sx, sx, sy * sy); }
SSR_DEV float ssr_ssim_stage2f(double pp, double sq) {                       // b1 * b2
  const double t2 = fma(SsrSsimK::n, sq, -pp);
  const float b1 = (float)pp + SsrSsimK::C1n;
  const float b2 = fmaf(SsrSsimK::cov, (float)t2, SsrSsimK::C2n);
  return b1 * b2;
}
SSR_DEV float ssr_ssim_stage3f(double pxy, double sxy, float b12) {
  const double t1 = fma(SsrSsimK::n, sxy, -pxy);
  const float a1 = fmaf(2.0f, (float)pxy, SsrSsimK::C1n);
  const float a2 = fmaf(2.0f * SsrSsimK::cov, (float)t1, SsrSsimK::C2n);
#ifndef SSR_HOST_EMU
  return (a1 * a2) * __builtin_amdgcn_rcpf(b12);                                      // one reciprocal for both ratios
#else
  return (a1 * a2) / b12;
#endif
}
SSR_DEV double ssr_ssim_value(double sx, double sy, double sq, double sxy) {
  double pxy, pp;
  ssr_ssim_stage1(sx, sy, pxy, pp);
  return (double)ssr_ssim_stage3f(pxy, sxy, ssr_ssim_stage2f(pp, sq));
}

// grid = (n_row_tiles * n_strips, n_items); block = SSR_SSIM_NT.
// Vertical pass: thread owns STRIDED input columns (coalesced loads, running 7-row sums in registers) and
// publishes the four column sums through LDS.  Horizontal pass: thread owns CPT CONTIGUOUS outputs, reads
// CPT+6 column sums per quantity once and slides the 7-wide window across them.
// A store of the column-sum hand-off, as a volatile LDS access: the compiler does not pair volatile stores into ds_write2_b64, so each
// sum leaves as its own ds_write_b64.  (tools/ubench/fp64_mix: an exchange written with ds_write_b64 costs a third of the same bytes
// written with ds_write2_b64; here, where the LDS pipe is ~65 % busy next to the VALU's ~69 %: 0.924 -> 0.908 ms per 1024 pairs, three
// alternations on one box, the same values - profiles/r05_notes.md section 8.)
#ifndef SSR_HOST_EMU
#define SSR_SSIM_ST(arr, idx, val) (*(volatile __attribute__((address_space(3))) double*)(&(arr)[idx]) = (val))
#else
#define SSR_SSIM_ST(arr, idx, val) ((arr)[idx] = (val))
#endif
template <int CPT, bool CONTIG, typename BLK>
SSR_BODY void ssr_ssim_body(const SsrSsimParams& p, BLK& blk, int tile, int item, char* lds_base) {
  constexpr int NT = SSR_SSIM_NT, PW = SsrSsimLds<CPT, CONTIG>::PW, VC = CPT + 1, W = SSR_SSIM_WIN;
  using Regs = SsrSsimRegs<CPT>;
  using Lds = SsrSsimLds<CPT, CONTIG>;
  Lds L(lds_base);
  const int row_tile = tile / p.n_strips, strip = tile % p.n_strips;
  const int item_v = item;                           // index of the partial record
  const float* xbase = p.x;
  if (p.vi_n > 0) { xbase += (int64_t)(item / p.vi_n) * p.x_plane; item = item % p.vi_n; }
  const int T = p.n_rows[item];
  const int out_rows = T - (W - 1);
  const int r0 = row_tile * p.rows_per_tile;
  const int r1 = (r0 + p.rows_per_tile < out_rows) ? r0 + p.rows_per_tile : out_rows;
  const int c_in0 = strip * ssr_ssim_strip_out(CPT);
  const int ncol_in = (p.F - c_in0 < NT * CPT + (W - 1)) ? p.F - c_in0 : NT * CPT + (W - 1);
  const int ncol_out = ncol_in - (W - 1);
  double* part = p.part + (int64_t)item_v * p.n_row_tiles * p.n_strips + tile;
  const int pitch = ssr_ssim_pitch(p);
  const SsrImage x(xbase + p.frame_off[item] * pitch, (int64_t)T * pitch);
  const SsrImage y(p.y + p.frame_off[item] * pitch, (int64_t)T * pitch);

  SSR_REGS(Regs, regs, blk);
  if (r0 >= r1 || ncol_out <= 0) {
    SSR_PHASE(blk, regs, if (tid == 0) *part = 0.0);
    return;
  }
  // Row steps s = 0 .. n_steps-1: step s adds row r0 + s to the running 7-row column sums and (s >= 7) subtracts row
  // r0 + s - 7; from s = 6 on the sums are those of output row r0 + s - 6.  The rows of step s + 2 are requested as soon as
  // step s has consumed its own (two steps in flight: one was not enough to cover the load latency - each wave sat idle
  // ~80 % of its life, and neither fewer LDS bytes nor fewer instructions moved the kernel).
  const int n_steps = (W - 1) + (r1 - r0);
#if defined(SSR_DEV_KNOBS) && defined(SSR_ABL_SSIM)   // timing-only ablations (wrong results): 1 no ssim_value, 2 no LDS, 4 no row loads
#define SSR_SABL(bit) ((SSR_ABL_SSIM) & (bit))
#else
#define SSR_SABL(bit) 0
#endif
#define SSR_SSIM_STEP(s_, SET, SLOT)                                                                                         \
  SSR_PHASE(blk, regs, {                                                                                                    \
    ssr_ssim_row_apply<CPT, false, SET, SLOT>(R, tid, (s_) >= W, ncol_in);                                                   \
    if ((s_) + 2 < n_steps && !SSR_SABL(4))                                                                                 \
      ssr_ssim_row_load<CPT, false, SET>(p, R, tid, x, y, r0 + (s_) + 2, ((s_) + 2 >= W) ? r0 + (s_) + 2 - W : -1, c_in0, ncol_in); \
    if ((s_) >= W - 1 && !SSR_SABL(2)) {                                                                                    \
      for (int i = 0; i < VC; ++i) {                                                                                        \
        const int c = ssr_ssim_col<CPT, false>(tid, i);                                                                     \
        if (c < ncol_in)                                                                                                    \
          for (int q = 0; q < 4; ++q) L.col[q * PW + ssr_ssim_slot<CPT>(c)] = R.cs[i][q];                                    \
      }                                                                                                                     \
    }                                                                                                                       \
  });                                                                                                                       \
  if ((s_) >= W - 1) {                                                                                                      \
    SSR_PHASE(blk, regs, {                                                                                                  \
      const int j0 = tid * CPT;                                                                                             \
      if (j0 < ncol_out) {                                                                                                  \
        double w[4][CPT];                            /* window sums, one quantity at a time (register pressure) */          \
        for (int q = 0; q < 4; ++q) {                                                                                       \
          double v[CPT + W - 1];                                                                                            \
          for (int d = 0; d < CPT + W - 1; ++d) {                                                                           \
            int c = j0 + d;                                                                                                 \
            if (c >= ncol_in) c = ncol_in - 1;      /* only feeds outputs that are masked below */                          \
            if (SSR_SABL(2)) v[d] = R.cs[d % VC][q];                                                                        \
            else v[d] = L.col[q * PW + ssr_ssim_slot<CPT>(c)];                                                              \
          }                                                                                                                 \
          double sw = v[0];                                                                                                 \
          for (int d = 1; d < W; ++d) sw += v[d];                                                                           \
          w[q][0] = sw;                                                                                                     \
          for (int i = 1; i < CPT; ++i) {                                                                                   \
            sw += v[i + W - 1] - v[i - 1];                                                                                  \
            w[q][i] = sw;                                                                                                   \
          }                                                                                                                 \
        }                                                                                                                   \
        for (int i = 0; i < CPT; ++i)                                                                                       \
          if (j0 + i < ncol_out) R.s += SSR_SABL(1) ? w[0][i] + w[1][i] + w[2][i] + w[3][i] : ssr_ssim_value(w[0][i], w[1][i], w[2][i], w[3][i]); \
      }                                                                                                                     \
    });                                                                                                                     \
  }
  SSR_PHASE(blk, regs, {
    for (int i = 0; i < VC; ++i)
      for (int q = 0; q < 4; ++q) R.cs[i][q] = 0.0;
    R.s = 0.0;
    if constexpr (CONTIG) {                                     // both seven-row rings start as zero rows
      const float z4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int k = 0; k < W; ++k) {
        for (int i = 0; i < VC; ++i) R.ring[k][i] = 0.0f;
        for (int h = 0; h < CPT / 4; ++h) ssr_st4(L.yring + k * Lds::RW + CPT * tid + 4 * h, z4);
        L.yring[k * Lds::RW + NT * CPT + (tid & 7)] = 0.0f;
      }
    }
    ssr_ssim_row_load<CPT, CONTIG, 0>(p, R, tid, x, y, r0, -1, c_in0, ncol_in);
    if (n_steps > 1) ssr_ssim_row_load<CPT, CONTIG, 1>(p, R, tid, x, y, r0 + 1, -1, c_in0, ncol_in);
  });
  if constexpr (CONTIG) {
    // One wave per workgroup and every hand-off inside it: wave-scope phases (no s_waitcnt lgkmcnt(0) at the boundaries).  The
    // column sums cross the lanes one quantity at a time through the single L.col array.
#define SSR_SSIM_STEP_C(s_, SET, SLOT)                                                                                       \
    SSR_WPHASE(blk, regs, {                                                                                                  \
      ssr_ssim_row_apply_contig<CPT, SET, SLOT>(R, L.yring + (SLOT) * Lds::RW, tid);                                              \
      if ((s_) + 2 < n_steps && !SSR_SABL(4))                                                                                \
        ssr_ssim_row_load<CPT, true, SET>(p, R, tid, x, y, r0 + (s_) + 2, -1, c_in0, ncol_in);                               \
    });                                                                                                                      \
    if ((s_) >= W - 1) {                                                                                                     \
      SSR_UNROLL for (int q = 0; q < 4; ++q) {                                                                               \
        SSR_WPHASE(blk, regs, {             /* every lane publishes: columns past the strip's end carry finite, unused sums */ \
          /* (a lane's columns are read by its LEFT neighbour only, and only the first six of them) */                      \
          SSR_UNROLL for (int i = 0; i < (CPT < 6 ? CPT : 6); ++i) SSR_SSIM_ST(L.col, ssr_ssim_slot<CPT>(CPT * tid + i), R.cs[i][q]); \
          L.col[ssr_ssim_slot<CPT>(NT * CPT + (tid < 7 ? tid : 7))] = R.cs[CPT][q];                                          \
        });                                                                                                                  \
        SSR_WPHASE(blk, regs, {                                                                                              \
          double v[CPT + W - 1];                                                                                             \
          SSR_UNROLL for (int d = 0; d < CPT + W - 1; ++d)                                                                   \
            v[d] = (d < CPT) ? R.cs[d < CPT ? d : 0][q]    /* the thread's own columns: already in its registers */         \
                             : L.col[ssr_ssim_slot<CPT>(CPT * tid + d)];                                                     \
          /* 7-wide windows in groups of four outputs over ten column sums: the shared core v3..v6, then v1 + v2 and       \
             v7 + v8 (11 additions per group; the eight-column variant shares v5 + v6 and v7 + v8 between its two groups:   \
             20) - the same association for every output whichever variant computes it */                                    \
          static_assert((CPT == 4 || CPT == 8) && W == 7, "window tree of the CONTIG variant");                             \
          double wq[CPT];                                                                                                    \
          {                                                                                                                  \
            const double p56 = v[5] + v[6], p78 = v[7] + v[8];                                                               \
            const double core = (v[3] + v[4]) + p56;                                                                         \
            const double wa = core + (v[1] + v[2]), wb = core + p78;                                                         \
            wq[0] = wa + v[0]; wq[1] = wa + v[7]; wq[2] = wb + v[2]; wq[3] = wb + v[9];                                      \
            if constexpr (CPT == 8) {                                                                                        \
              const double core2 = p78 + (v[9] + v[10]);                                                                     \
              const double wa2 = core2 + p56, wb2 = core2 + (v[11] + v[12]);                                                 \
              wq[4] = wa2 + v[4]; wq[5] = wa2 + v[11]; wq[6] = wb2 + v[6]; wq[7] = wb2 + v[13];                              \
            }                                                                                                                \
          }                                                                                                                  \
          /* fold the quantity into the SSIM expression at once (ssr_ssim_value's stages): two live values per output.       \
             The thread's four float32 values of a row are added in float32 (values in [-1, 1]: 1e-7 per row, unbiased,    \
             against the 1e-5 bar on the mean) and join the float64 sum once. */                                             \
          const int j0 = tid * CPT;                                                                                          \
          float row_s = 0.0f;                                                                                                \
          SSR_UNROLL for (int i = 0; i < CPT; ++i) {                                                                         \
            if (q == 0) R.pxy[i] = wq[i];                                                                                    \
            if (q == 1) { const double sx = R.pxy[i]; ssr_ssim_stage1(sx, wq[i], R.pxy[i], R.pb[i]); }                       \
            if (q == 2) R.den[i] = ssr_ssim_stage2f(R.pb[i], wq[i]);                                                         \
            if (q == 3) {                                                                                                    \
              const float sv = ssr_ssim_stage3f(R.pxy[i], wq[i], R.den[i]);                                                  \
              row_s += (j0 + i < ncol_out) ? sv : 0.0f;                                                                      \
            }                                                                                                                \
          }                                                                                                                  \
          if (q == 3) R.s += (double)row_s;                                                                                  \
        });                                                                                                                  \
      }                                                                                                                      \
    }
    // ring slot = step mod 7, prefetch set = step mod 2: fourteen steps per trip, all indices static
    for (int s0 = 0; s0 < n_steps; s0 += 14) {
#define SSR_SSIM_STEP_K(k) if (s0 + (k) < n_steps) { SSR_SSIM_STEP_C(s0 + (k), (k) % 2, (k) % 7) }
      SSR_SSIM_STEP_K(0) SSR_SSIM_STEP_K(1) SSR_SSIM_STEP_K(2) SSR_SSIM_STEP_K(3) SSR_SSIM_STEP_K(4) SSR_SSIM_STEP_K(5) SSR_SSIM_STEP_K(6)
      SSR_SSIM_STEP_K(7) SSR_SSIM_STEP_K(8) SSR_SSIM_STEP_K(9) SSR_SSIM_STEP_K(10) SSR_SSIM_STEP_K(11) SSR_SSIM_STEP_K(12) SSR_SSIM_STEP_K(13)
#undef SSR_SSIM_STEP_K
    }
#undef SSR_SSIM_STEP_C
  } else {
    for (int s0 = 0; s0 < n_steps; s0 += 2) {
      SSR_SSIM_STEP(s0, 0, 0)
      if (s0 + 1 < n_steps) { SSR_SSIM_STEP(s0 + 1, 1, 0) }
    }
  }
#undef SSR_SSIM_STEP
#define SSR_GET_S(q) R.s
  SSR_BLOCK_SUM(blk, regs, NT, 1, L.sc0, L.sc1, L.res, SSR_GET_S);
#undef SSR_GET_S
  SSR_PHASE(blk, regs, if (tid == 0) *part = L.res[0]);
}

// ---------------------------------------------------------------------------------------------------
// LSD / SISpec / log-SISpec on given spectrogram pairs (backs AudioMetrics.lsd / .sispec called
// directly on tensors).  Produces the same partial record as the fused STFT epilogue.
struct SsrSpecRedParams {
  const float* x; const float* y;
  const int64_t* frame_off; const int32_t* n_rows;
  int F, metric_mask, rows_per_chunk, n_chunks;
  double* part;  // [n_items, n_chunks, SSR_NPART]
};
struct SsrSpecRedRegs { double acc[7]; double lsd_sum; };
struct SsrSpecRedLds {
  static constexpr int NT = 256;
  static constexpr size_t bytes() { return sizeof(double) * (NT + 16 + 8); }
  double* sc0; double* sc1; double* res;
  SSR_MEMBER explicit SsrSpecRedLds(char* base) {
    sc0 = reinterpret_cast<double*>(base); sc1 = sc0 + NT; res = sc1 + 16;
  }
};

template <typename BLK>
SSR_BODY void ssr_specred_body(const SsrSpecRedParams& p, BLK& blk, int chunk, int item, char* lds_base) {
  constexpr int NT = SsrSpecRedLds::NT;
  SsrSpecRedLds L(lds_base);
  const int T = p.n_rows[item];
  const int t0 = chunk * p.rows_per_chunk;
  const int t1 = (t0 + p.rows_per_chunk < T) ? t0 + p.rows_per_chunk : T;
  const float* x = p.x + p.frame_off[item] * p.F;
  const float* y = p.y + p.frame_off[item] * p.F;
  double* part = p.part + ((int64_t)item * p.n_chunks + chunk) * SSR_NPART;
  const bool want_lsd = p.metric_mask & SSR_M_LSD;
  SSR_REGS(SsrSpecRedRegs, regs, blk);
  SSR_PHASE(blk, regs, for (int q = 0; q < 7; ++q) R.acc[q] = 0.0; R.lsd_sum = 0.0);
  for (int t = t0; t < t1; ++t) {
    SSR_PHASE(blk, regs, {
      R.acc[0] = 0.0;
      for (int k = tid; k < p.F; k += NT)
        ssr_accumulate_metrics(x[(int64_t)t * p.F + k], y[(int64_t)t * p.F + k], p.metric_mask, R.acc);
      L.sc0[tid] = R.acc[0];
    });
    if (want_lsd) {
      SSR_PHASE(blk, regs, if (tid < 16) {
        double s = 0.0;
        for (int i = tid; i < NT; i += 16) s += L.sc0[i];
        L.sc1[tid] = s;
      });
      SSR_PHASE(blk, regs, if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < 16; ++i) s += L.sc1[i];
        R.lsd_sum += sqrt(s / (double)p.F);
      });
    }
  }
#define SSR_GET_ACC(q) R.acc[(q) + 1]
  SSR_BLOCK_SUM(blk, regs, NT, 6, L.sc0, L.sc1, L.res, SSR_GET_ACC);
#undef SSR_GET_ACC
  SSR_PHASE(blk, regs, if (tid == 0) {
    part[0] = R.lsd_sum;
    for (int q = 0; q < 6; ++q) part[1 + q] = L.res[q];
    part[7] = 0.0;
  });
}

// ---------------------------------------------------------------------------------------------------
// Finalisation: one thread per item.  out[item*4 + {0,1,2,3}] = lsd, log_sispec, sispec, ssim
// (the key order of ssr_eval/metrics.py:98-103).  Entries whose metric bit is clear are left as NaN.
struct SsrFinalizeParams {
  const double* part; int n_chunks;        // [n_items, n_chunks, SSR_NPART] (may be null)
  const double* ssim_part; int n_tiles;    // [n_items, n_tiles] (may be null)
  const int32_t* n_rows;                   // T_i
  int F, metric_mask, n_items;
  double* out;                             // [n_items, 4]
  // virtual items (ssr_pair_metrics_multi): record v = k * vi_n + i belongs to item i, key out_key0 + k of out_keys keys per item:
  // out[(i * out_keys + out_key0 + k) * 4]; vi_n = 0: plain
  int vi_n, out_keys, out_key0;
};

SSR_DEV double ssr_sispec_from_sums(double sdd, double stt, double sdt) {
  // ssr_eval/metrics.py:114-121 with energy_unify (utils.py:79-82): scaled = (Set * t) / (Stt + EPS), from the sums on
  // the difference d = e - t:  Set = Stt + Sdt,  1 - alpha = (EPS - Sdt) / (Stt + EPS) (no cancellation),
  // noise = e - alpha t = d + (1 - alpha) t  ->  ||noise||^2 = Sdd + 2 (1 - alpha) Sdt + (1 - alpha)^2 Stt.
  const double EPS = 1e-12;
  const double den = stt + EPS;
  const double alpha = (stt + sdt) / den;
  const double beta = (EPS - sdt) / den;                 // 1 - alpha
  const double tt = alpha * alpha * stt;
  double nn = sdd + 2.0 * beta * sdt + beta * beta * stt;
  if (nn < 0.0) nn = 0.0;
  return 10.0 * log10(tt / (nn + EPS) + EPS);
}

SSR_DEV void ssr_finalize_item(const SsrFinalizeParams& p, int item) {
  const double nan_ = NAN;
  const int real = p.vi_n > 0 ? item % p.vi_n : item;
  double* o = p.vi_n > 0 ? p.out + ((int64_t)real * p.out_keys + p.out_key0 + item / p.vi_n) * 4 : p.out + (int64_t)item * 4;
  const int T = p.n_rows[real];
  o[0] = o[1] = o[2] = o[3] = nan_;
  if (p.part) {
    double s[SSR_NPART];
    for (int q = 0; q < SSR_NPART; ++q) s[q] = 0.0;
    for (int c = 0; c < p.n_chunks; ++c)
      for (int q = 0; q < SSR_NPART; ++q) s[q] += p.part[((int64_t)item * p.n_chunks + c) * SSR_NPART + q];
    if (p.metric_mask & SSR_M_LSD) o[0] = s[0] / (double)T;
    if (p.metric_mask & SSR_M_LOG_SISPEC) o[1] = ssr_sispec_from_sums(s[4], s[5], s[6]);
    if (p.metric_mask & SSR_M_SISPEC) o[2] = ssr_sispec_from_sums(s[1], s[2], s[3]);
  }
  if (p.ssim_part && (p.metric_mask & SSR_M_SSIM)) {
    double s = 0.0;
    for (int c = 0; c < p.n_tiles; ++c) s += p.ssim_part[(int64_t)item * p.n_tiles + c];
    // images smaller than the 7x7 window have no valid output (skimage raises for them): NaN, never a division by <= 0
    o[3] = (T >= SSR_SSIM_WIN && p.F >= SSR_SSIM_WIN) ? s / ((double)(T - 6) * (double)(p.F - 6)) : nan_;
  }
}
