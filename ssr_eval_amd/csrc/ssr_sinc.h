// Kernel body N2: band-limited (windowed-sinc) resampler with the arithmetic of resampy.resample(filter="kaiser_best"),
// which is what librosa.load(file, sr) / librosa.resample(res_type="kaiser_best") run when a file's rate differs from the
// requested one (ssr_eval/eval.py:242, ssr_eval/metrics.py:22-23).  resampy is a third-party package absent from the
// reference tree and from the image: this follows its published algorithm (resampy 0.2.x interpn.resample_f; J.O. Smith's
// bandlimited interpolation) - parity with the real package is unpinned.
//
// For output sample t (time register tr[t] = t / ratio, accumulated by the host exactly as resampy's loop does):
//   n = int(tr), frac = scale * (tr - n), scale = min(1, ratio)
//   left wing : offset = int(frac * num_table), eta = frac * num_table - offset,
//               y += (win[offset + i*step] + eta * delta[offset + i*step]) * x[n - i],  i = 0 .. min(n + 1, (nwin - offset) / step) - 1
//   right wing: frac = scale - frac, same offset / eta, y += (...) * x[n + k + 1],      k = 0 .. min(n_in - n - 1, (nwin - offset) / step) - 1
// with float64 weights and products and the running sum y rounded to float32 after every tap (resampy accumulates
// into the float32 output array).  No fused multiply-add: every product and sum is rounded separately, so the result is
// bit-identical to the NumPy restatement (oracle/resampy.py) given the same tables.
//
// Work mapping (round 3; round 2 ran one thread per consecutive output with ~128 dependent table gathers from global memory
// and every input sample fetched ~128 times through L1 / L2):
//   * a workgroup owns a CONTIGUOUS run of outputs and stages the input window that run touches in LDS once (zero outside the
//     signal; the taps beyond the signal's ends are skipped as in resampy, the zeros are never read);
//   * for a rational rate pair sr_new / sr_orig = a / b the filter phase repeats every a outputs, so a wave takes ONE phase r
//     and its 64 lanes take 64 CONSECUTIVE PERIODS (outputs a j + r): all lanes of a load read the same table entry (the
//     accumulated time register drifts by < 1e-10, i.e. at most the neighbouring entry) - one cache line per load instead of 64 -
//     and their input samples sit b apart in LDS.  b odd (44.1 -> 48 kHz: 147) is conflict-free; an even b (48 -> 44.1 kHz and
//     16 -> 44.1 kHz: 160 = 5 * 32 puts all 64 lanes on ONE bank - measured 2.5x slower) gets the window stored with one pad
//     word per 32 (PAD: stride 165);
//   * when 64 periods of input do not fit the LDS window (44.1 -> 16 kHz: b = 441) a wave takes 2 or 4 phases x 32 or 16
//     periods (`pw`: 2 or 4 table lines per load instead of 1);
//   * `period` = 1 (no rational structure given, or even 16 periods do not fit) falls back to consecutive outputs per lane:
//     the table reads scatter again, the input still comes from LDS.
#pragma once
#include "ssr_block.h"
#include <type_traits>

constexpr int SSR_SINC_NT = 256;
constexpr int SSR_SINC_AHEAD = 4;       // taps whose operands are requested ahead of their use (ssr_sinc_one's uniform-offset loop)

struct SsrSincParams {
  const float* in;
  const int64_t* in_off;    // [n_items]
  const int32_t* in_len;    // [n_items]
  const int64_t* out_off;   // [n_items]
  const int32_t* out_len;   // [n_items]  int(n_in * ratio)
  const double* time_reg;   // [>= max_out_len] time register of output t (independent of the item)
  const double* win;        // [nwin]  interp_win (already scaled by ratio when ratio < 1)
  const double* delta;      // [nwin]  interp_delta: win[i + 1] - win[i], last entry 0
  int nwin, num_table, index_step;
  double scale;             // min(1, ratio)
  float* out;
  // block geometry (host: ssr_sinc_geometry)
  int period;               // a (outputs per filter-phase period), or 1
  int pw;                   // phases per wave (1, 2, 4): 64 / pw lanes = consecutive periods of one phase
  int m;                    // a block = (64 / pw) m periods = (64 / pw) m period outputs
  int max_room;             // most taps a wing can have: nwin / index_step (+1)
  int lds_floats;           // floats of LDS the launch provides for the input window
  // device launches only (ssr_resample_sinc builds it per call, ssr_sinc_table_body): the two tables re-ordered by filter phase,
  // tab[offset * tab_r + i] = {win, delta}[offset + i * index_step] (zeros past the table's end), tab_rows = index_step + 1 rows
  const double* tab;
  int tab_r, tab_rows;
};

struct alignas(16) SsrSincPair { double w, d; };
struct alignas(64) SsrSincQuad { SsrSincPair e[4]; };

// Round 5: the phase-major copy of the interpolation tables.  The taps of ONE output walk the tables with a stride of index_step
// entries (4 KB for kaiser_best): 2 x 129 different cache lines per output, fetched by scalar loads whose latency five waves per
// SIMD could not hide (one tap per ~500 cycles and wave; the kernel ran at the rate that model predicts).  Re-ordered by phase the
// taps of an output are consecutive 16-byte {win, delta} pairs, read four taps ahead with vector loads.
SSR_HD int ssr_sinc_tab_r(int max_room) { return (max_room + 3) / 4 * 4 + 4; }     // row length: whole quads + one quad of read-ahead
SSR_DEV void ssr_sinc_table_body(const SsrSincParams& p, int64_t e) {
  if (e >= (int64_t)p.tab_rows * p.tab_r) return;
  const int offset = (int)(e / p.tab_r), i = (int)(e - (int64_t)offset * p.tab_r);
  const int64_t idx = offset + (int64_t)i * p.index_step;
  double* t = const_cast<double*>(p.tab) + 2 * e;
  t[0] = idx < p.nwin ? p.win[idx] : 0.0;
  t[1] = idx < p.nwin ? p.delta[idx] : 0.0;
}

// One output, input through `xs` (xs[pad(s - lo)] = x[s]; lo = 0, no pad and xs = x: straight from global memory).
template <bool PAD, bool STAGED>
SSR_DEV float ssr_sinc_one(const SsrSincParams& p, const SsrView<double>& vwin, const SsrView<double>& vdelta, const float* xs, int lo,
                           int n_in, int64_t t) {
  const double tr = p.time_reg[t];
  const int n = (int)tr;
  double frac = ssr_fmul_rn(p.scale, ssr_fadd_rn(tr, -(double)n));
  float y = 0.0f;
  for (int wing = 0; wing < 2; ++wing) {
    const double index_frac = ssr_fmul_rn(frac, (double)p.num_table);
    const int offset = (int)index_frac;
    const double eta = ssr_fadd_rn(index_frac, -(double)offset);
    const int room = (p.nwin - offset) / p.index_step;
    const int avail = wing == 0 ? n + 1 : n_in - n - 1;
    const int cnt = room < avail ? room : avail;
    const int x0 = (wing == 0 ? n : n + 1) - lo, dx = wing == 0 ? -1 : 1;
#ifndef SSR_HOST_EMU
    // The taps' table entries come from the phase-major table (consecutive 16-byte pairs along the taps), four taps ahead of their use.
    // One phase per wave (the usual geometry): the lanes' offsets are equal and the entries wave-uniform - ONE scalar load
    // (s_load_dwordx16) per four taps.  Where they are not equal - the accumulated time register straddles a table boundary
    // (exactly integral index_frac in exact arithmetic: every fifth phase of 44.1 -> 48 kHz splits its lanes between two neighbouring
    // entries), or the wave holds several phases (pw > 1) - each lane reads ITS row with vector loads.  (Until round 5 the strided
    // scalar loads waited ~one L2 round trip per tap, and the unequal case gathered win[] / delta[] 4 KB apart per tap: a fifth of
    // the outputs took most of the kernel's time.  The uniform entries as broadcast VECTOR loads were no faster than that: a
    // broadcast load still returns 1 KB per wave through the 64 B/clk vector-memory return path.)
    // The loop runs to a wave-uniform tap count with no per-lane condition on the signal's ends: a tap beyond them reads one of the
    // zeros staged around the signal, and y + w * 0 is y (resampy skips that tap; only the sign of an exact zero could differ).
    if (STAGED) {
      const int o1 = __builtin_amdgcn_readfirstlane(offset);
      const bool uniform = __builtin_amdgcn_ballot_w64(offset != o1) == 0ull;
      // (read as doubles: type-based alias analysis then knows that the float stores of the outputs cannot touch them, which is
      // what lets the compiler use scalar loads; an aggregate copy of a quad became 16-byte vector loads)
      const double* ta = p.tab + 2 * (int64_t)o1 * p.tab_r;               // wave-uniform
      const double* tl = p.tab + 2 * (int64_t)offset * p.tab_r;           // the lane's own row
      const int room_u = uniform ? (p.nwin - o1) / p.index_step : p.max_room - 1;     // (max_room - 1 = the wing of offset 0: the longest)
      // taps i .. i + 3 use entries c.e[0..3] and input samples a0..a3; the entries AND the samples of the next four taps are
      // requested before these four are computed (with a few waves per SIMD nothing else hides a load's latency)
      auto xat = [&](int i) { const int xi = x0 + dx * i; return xs[PAD ? xi + (xi >> 5) : xi]; };
      auto wing_loop = [&](auto own_tag) {
        constexpr bool OWN = decltype(own_tag)::value;
        // OWN rows: a lane's wing can be shorter than the loop - (nwin - offset) / index_step taps - and the entry behind its last one is
        // a real entry of the row unless it lies past the table's end: the lane's weight is forced to zero there, y + 0 * x is y.
        const double* t = OWN ? tl : ta;
        auto tap = [&](const SsrSincPair& c, float xv32, int i) {
          double weight = ssr_fadd_rn(c.w, ssr_fmul_rn(eta, c.d));
          if (OWN) weight = (i < room) ? weight : 0.0;
          y = (float)ssr_fadd_rn((double)y, ssr_fmul_rn(weight, (double)xv32));
        };
        auto quad = [&](int q) {
          SsrSincQuad a;
          SSR_UNROLL for (int k = 0; k < 4; ++k) { a.e[k].w = t[8 * q + 2 * k]; a.e[k].d = t[8 * q + 2 * k + 1]; }
          return a;
        };
        SsrSincQuad c = quad(0);
        float a0 = xat(0), a1 = xat(1), a2 = xat(2), a3 = xat(3);        // (staged zeros surround the signal: always readable)
        int i = 0;
        for (; i + 4 <= room_u; i += 4) {
          const SsrSincQuad n = quad(i / 4 + 1);
          const float b0 = xat(i + 4), b1 = xat(i + 5), b2 = xat(i + 6), b3 = xat(i + 7);
          tap(c.e[0], a0, i); tap(c.e[1], a1, i + 1); tap(c.e[2], a2, i + 2); tap(c.e[3], a3, i + 3);
          c = n;
          a0 = b0; a1 = b1; a2 = b2; a3 = b3;
        }
        if (i < room_u) tap(c.e[0], a0, i);
        if (i + 1 < room_u) tap(c.e[1], a1, i + 1);
        if (i + 2 < room_u) tap(c.e[2], a2, i + 2);
      };
      if (uniform) wing_loop(std::false_type{});
      else wing_loop(std::true_type{});
    } else
#endif
    for (int i = 0; i < cnt; ++i) {
      const unsigned idx = (unsigned)(offset + i * p.index_step);
      const double weight = ssr_fadd_rn(vwin.at(idx), ssr_fmul_rn(eta, vdelta.at(idx)));
      const int xi = x0 + dx * i;
      const double xv = (double)xs[PAD ? xi + (xi >> 5) : xi];
      y = (float)ssr_fadd_rn((double)y, ssr_fmul_rn(weight, xv));
    }
    frac = ssr_fadd_rn(p.scale, -frac);           // "invert P"
  }
  return y;
}

#ifdef SSR_HOST_EMU
// Scalar statement of what a lane of the device's phase-major loops computes for one output (tests/emu): the lane's own row of `tab`,
// the loop run to the longest wing (max_room - 1 taps) with a zero weight past the lane's own wing, input samples outside the signal
// read as the staged zeros.  Must equal ssr_sinc_one (which skips those taps) up to the sign of an exact zero.
static inline float ssr_sinc_one_tab_host(const SsrSincParams& p, const float* x, int n_in, int64_t t) {
  const double tr = p.time_reg[t];
  const int n = (int)tr;
  double frac = ssr_fmul_rn(p.scale, ssr_fadd_rn(tr, -(double)n));
  float y = 0.0f;
  for (int wing = 0; wing < 2; ++wing) {
    const double index_frac = ssr_fmul_rn(frac, (double)p.num_table);
    const int offset = (int)index_frac;
    const double eta = ssr_fadd_rn(index_frac, -(double)offset);
    const int room = (p.nwin - offset) / p.index_step;
    const double* row = p.tab + 2 * (int64_t)offset * p.tab_r;
    for (int i = 0; i < p.max_room - 1; ++i) {
      double weight = ssr_fadd_rn(row[2 * i], ssr_fmul_rn(eta, row[2 * i + 1]));
      if (i >= room) weight = 0.0;
      const int64_t xi = wing == 0 ? (int64_t)n - i : (int64_t)n + 1 + i;
      const double xv = (xi >= 0 && xi < n_in) ? (double)x[xi] : 0.0;
      y = (float)ssr_fadd_rn((double)y, ssr_fmul_rn(weight, xv));
    }
    frac = ssr_fadd_rn(p.scale, -frac);
  }
  return y;
}
#endif

// grid = n_items * blocks_per_item workgroups of SSR_SINC_NT threads; dynamic LDS: lds_floats floats
template <bool PAD, typename BLK>
SSR_BODY void ssr_sinc_block_body(const SsrSincParams& p, BLK& blk, int item, int block, char* lds_base) {
  constexpr int NT = SSR_SINC_NT, NWAVES = NT / 64;
  float* xs = reinterpret_cast<float*>(lds_base);
  const int n_in = p.in_len[item], n_out = p.out_len[item];
  const int P = p.period, LPP = 64 / p.pw, JB = LPP * p.m;      // lanes per phase; periods per block
  const int64_t t0 = (int64_t)block * JB * P;                   // the block's outputs: [t0, t1)
  if (t0 >= n_out || n_in <= 0) return;
  const int64_t t1 = (t0 + (int64_t)JB * P < n_out) ? t0 + (int64_t)JB * P : (int64_t)n_out;
  const float* x = p.in + p.in_off[item];
  const SsrView<double> vwin(p.win, p.nwin), vdelta(p.delta, p.nwin);
  // input window of the block: every index a tap of one of its outputs can touch
  // (+ SSR_SINC_AHEAD samples either side: the uniform-offset loop requests the samples of the next four taps before it knows whether
  // the wing has that many)
  const int lo = (int)p.time_reg[t0] - (p.max_room - 1) - SSR_SINC_AHEAD;
  const int hi = (int)p.time_reg[t1 - 1] + p.max_room + SSR_SINC_AHEAD;
  const int W = hi - lo + 1;
  const bool staged = (PAD ? W + (W >> 5) + 1 : W) <= p.lds_floats;   // (the host's geometry guarantees it; a safety net otherwise)
  SSR_REGS(int, regs, blk);
  SSR_PHASE(blk, regs, {
    if (staged)
      for (int i = tid; i < W; i += NT) {
        const int s = lo + i;
        xs[PAD ? i + (i >> 5) : i] = (s >= 0 && s < n_in) ? x[s] : 0.0f;
      }
  });
  SSR_PHASE(blk, regs, {
    const int wave = ssr_wave_of(tid), lane = tid & 63;
    const int sub = lane / LPP, jl = lane - sub * LPP;
    const int n_rg = (P + p.pw - 1) / p.pw;                       // groups of pw phases
    for (int w = wave; w < n_rg * p.m; w += NWAVES) {             // work item = (phase group rg, period group g): wave-uniform
      const int rg = w / p.m, g = w - rg * p.m;
      const int r = rg * p.pw + sub;
      const int64_t t = t0 + (int64_t)(LPP * g + jl) * P + r;
      if (r < P && t < t1)
        p.out[p.out_off[item] + t] = staged ? ssr_sinc_one<PAD, true>(p, vwin, vdelta, xs, lo, n_in, t)
                                            : ssr_sinc_one<false, false>(p, vwin, vdelta, x, 0, n_in, t);
    }
  });
}

// Block geometry for a rate pair (host side; also used by the emulation harness).  period_hint: a of the reduced ratio
// sr_new / sr_orig = a / b, or <= 1 when unknown.  lds_cap_floats: the largest input window to stage.
struct SsrSincGeometry { int period, pw, m, max_room, lds_floats, outputs_per_block, pad; };
SSR_HD SsrSincGeometry ssr_sinc_geometry(int period_hint, double ratio, int nwin, int index_step, int lds_cap_floats) {
  SsrSincGeometry g;
  g.max_room = nwin / index_step + 1;
  const int halo = 2 * g.max_room + 4 + 2 * SSR_SINC_AHEAD;
  int P = period_hint > 1 ? period_hint : 1, pw = 1;
  const double b = (double)P / ratio;                            // input samples per period
  const double cap = (double)lds_cap_floats * 32.0 / 33.0 - 2.0; // (room for the pad words)
  if (P > 1) {
    while (pw < 4 && (64 / pw) * b + halo > cap) pw *= 2;
    if ((64 / pw) * b + halo > cap) { P = 1; pw = 1; }
  }
  const double span = (P > 1 ? (64 / pw) * b : 64.0 / ratio);     // input samples spanned by one period group of a wave
  int m = (int)((cap - halo) / span);
  const int per_m = (64 / pw) * P;
  const int m_target = (8192 + per_m - 1) / per_m;              // ~8 k outputs per block
  if (m > m_target) m = m_target;
  if (m < 1) m = 1;
  g.period = P; g.pw = pw; g.m = m;
  g.outputs_per_block = per_m * m;
  // pad the LDS window when the lanes' input stride is even (a multiple of 32 would put every lane on one bank)
  const long bi = (long)(b + 0.5);
  g.pad = (P > 1 && (bi % 2) == 0) ? 1 : 0;
  const double w = (double)g.outputs_per_block / ratio + halo + 2;
  g.lds_floats = (int)(g.pad ? w * 33.0 / 32.0 + 2 : w) + 1;
  return g;
}
