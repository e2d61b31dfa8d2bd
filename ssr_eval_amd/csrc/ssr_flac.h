// N2 ingest: a FLAC decoder for the files the reference reads through librosa.load / soundfile (ssr_eval/eval.py:158-169 lists
// ".wav" and ".flac" - the VCTK test set is FLAC; ssr_eval/eval.py:242, ssr_eval/metrics.py:21-24).  Host code (the reference
// decodes on the host too); the decoded 16-bit frames go straight into the page-locked staging arena and cross PCIe as int16.
//
// Written from the format specification (RFC 9639): STREAMINFO, frame header (fixed / variable block size, every block-size and
// sample-rate code, UTF-8 coded frame / sample number, CRC-8), subframes CONSTANT / VERBATIM / FIXED (order 0-4) / LPC (order 1-32)
// with wasted bits, Rice residuals (4- and 5-bit parameters, partition orders, escaped partitions), the three stereo
// decorrelation modes, 4-32 bits per sample, frame CRC-16.  Self-check: every file carries the MD5 of its decoded PCM in
// STREAMINFO; the decoder recomputes it (RFC 1321, below) and refuses a file whose signature does not match - a bit-exact pin
// that needs no second decoder.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

namespace ssr_flac {

// ---- MD5 (RFC 1321) ---------------------------------------------------------------------------------------------------------
struct Md5 {
  uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
  uint64_t n_bytes = 0;
  uint8_t buf[64];
  size_t fill = 0;

  static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
        0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
        0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
        0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
        0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
        0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t m[16];
    for (int i = 0; i < 16; ++i) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      if (i < 16) { f = (B & C) | (~B & D); g = i; }
      else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
      else { f = C ^ (B | ~D); g = (7 * i) & 15; }
      const uint32_t t = D;
      D = C;
      C = B;
      B = B + rol(A + f + K[i] + m[g], S[i]);
      A = t;
    }
    a += A; b += B; c += C; d += D;
  }
  void update(const uint8_t* p, size_t n) {
    n_bytes += n;
    if (fill) {
      const size_t take = n < 64 - fill ? n : 64 - fill;
      memcpy(buf + fill, p, take);
      fill += take; p += take; n -= take;
      if (fill == 64) { block(buf); fill = 0; }
    }
    for (; n >= 64; p += 64, n -= 64) block(p);
    if (n) { memcpy(buf, p, n); fill = n; }
  }
  void finish(uint8_t out[16]) {
    const uint64_t bits = n_bytes * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (8 * i));
    update(len, 8);
    const uint32_t v[4] = {a, b, c, d};
    for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(v[i / 4] >> (8 * (i % 4)));
  }
};

// ---- stream description -----------------------------------------------------------------------------------------------------
struct Info {
  int sample_rate = 0, channels = 0, bits = 0, min_block = 0, max_block = 0;
  int64_t total_samples = 0;      // per channel; 0 = unknown
  uint8_t md5[16] = {};
  bool has_md5 = false;           // an all-zero signature means "not computed by the encoder"
  size_t audio_offset = 0;        // first frame
};

struct Reader {
  const uint8_t* p;
  size_t n, pos = 0;
  uint64_t acc = 0;
  int nacc = 0;                   // valid bits in acc (right-aligned)
  bool bad = false;

  Reader(const uint8_t* data, size_t size, size_t at) : p(data), n(size), pos(at) {}
  inline void refill() {
    while (nacc <= 56 && pos < n) { acc = (acc << 8) | p[pos++]; nacc += 8; }
  }
  inline uint32_t bits(int k) {   // k in [0, 32]
    if (k == 0) return 0;
    if (nacc < k) { refill(); if (nacc < k) { bad = true; nacc = 0; return 0; } }
    nacc -= k;
    return (uint32_t)((acc >> nacc) & ((k == 32) ? 0xffffffffu : ((1u << k) - 1u)));
  }
  inline int32_t sbits(int k) {   // k in [1, 32], two's complement
    const uint32_t v = bits(k);
    return k == 32 ? (int32_t)v : (int32_t)((v ^ (1u << (k - 1))) - (1u << (k - 1)));
  }
  inline int64_t sbits_wide(int k) {   // k up to 33 (side channel of 32-bit audio)
    if (k <= 32) return sbits(k);
    const int64_t hi = sbits(k - 32);
    return (hi << 32) | bits(32);
  }
  inline uint32_t unary() {       // zeros before the next 1
    uint32_t q = 0;
    for (;;) {
      if (nacc == 0) { refill(); if (nacc == 0) { bad = true; return q; } }
      const uint64_t window = acc & ((nacc == 64) ? ~0ull : ((1ull << nacc) - 1ull));
      if (window == 0) { q += (uint32_t)nacc; nacc = 0; continue; }
      const int lead = __builtin_clzll(window) - (64 - nacc);
      q += (uint32_t)lead;
      nacc -= lead + 1;
      return q;
    }
  }
  inline void align() { nacc -= nacc % 8; }
  inline size_t byte_pos() const { return pos - (size_t)(nacc / 8); }      // only meaningful when aligned
};

inline uint8_t crc8(const uint8_t* p, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= p[i];
    for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
  }
  return c;
}
struct Crc16Table {
  uint16_t t[256];
  Crc16Table() {
    for (int i = 0; i < 256; ++i) {
      uint16_t c = (uint16_t)(i << 8);
      for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
      t[i] = c;
    }
  }
};
inline uint16_t crc16(const uint8_t* p, size_t n) {
  static const Crc16Table tab;
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ tab.t[(c >> 8) ^ p[i]]);
  return c;
}

inline bool parse_header(const uint8_t* d, size_t n, Info& info, std::string& err) {
  size_t pos = 0;
  if (n >= 10 && d[0] == 'I' && d[1] == 'D' && d[2] == '3') {             // an ID3v2 tag in front of the stream
    const size_t sz = ((size_t)(d[6] & 0x7f) << 21) | ((size_t)(d[7] & 0x7f) << 14) | ((size_t)(d[8] & 0x7f) << 7) | (size_t)(d[9] & 0x7f);
    pos = 10 + sz;
  }
  if (pos + 4 > n || memcmp(d + pos, "fLaC", 4) != 0) { err = "not a FLAC stream (no fLaC marker)"; return false; }
  pos += 4;
  bool have_info = false;
  for (;;) {
    if (pos + 4 > n) { err = "truncated metadata"; return false; }
    const bool last = (d[pos] & 0x80) != 0;
    const int type = d[pos] & 0x7f;
    const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
    pos += 4;
    if (pos + len > n) { err = "truncated metadata block"; return false; }
    if (type == 0) {
      if (len < 34) { err = "short STREAMINFO"; return false; }
      const uint8_t* s = d + pos;
      info.min_block = (s[0] << 8) | s[1];
      info.max_block = (s[2] << 8) | s[3];
      info.sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      info.channels = ((s[12] >> 1) & 7) + 1;
      info.bits = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      info.total_samples = ((int64_t)(s[13] & 0x0f) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
      memcpy(info.md5, s + 18, 16);
      info.has_md5 = false;
      for (int i = 0; i < 16; ++i) info.has_md5 = info.has_md5 || info.md5[i] != 0;
      have_info = true;
    }
    pos += len;
    if (last) break;
  }
  if (!have_info) { err = "no STREAMINFO block"; return false; }
  if (info.sample_rate <= 0 || info.bits < 4 || info.bits > 32) { err = "invalid STREAMINFO"; return false; }
  info.audio_offset = pos;
  return true;
}

// One subframe -> out[0 .. bs).  bps: bits per sample of this subframe (frame bits + 1 for a side channel).
inline bool decode_subframe(Reader& r, int bs, int bps, int64_t* out, std::string& err) {
  if (r.bits(1) != 0) { err = "subframe padding bit set"; return false; }
  const int type = (int)r.bits(6);
  int wasted = 0;
  if (r.bits(1)) wasted = (int)r.unary() + 1;
  bps -= wasted;
  if (bps < 1) { err = "wasted bits exceed the sample size"; return false; }
  int order = 0;
  bool lpc = false;
  if (type == 0) {
    const int64_t v = r.sbits_wide(bps);
    for (int i = 0; i < bs; ++i) out[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < bs; ++i) out[i] = r.sbits_wide(bps);
  } else if (type >= 8 && type <= 12) {
    order = type - 8;
  } else if (type >= 32) {
    order = type - 31;
    lpc = true;
  } else {
    err = "reserved subframe type";
    return false;
  }
  if (type >= 8) {
    if (order > bs) { err = "predictor order exceeds the block size"; return false; }
    for (int i = 0; i < order; ++i) out[i] = r.sbits_wide(bps);
    int32_t coef[32];
    int shift = 0;
    if (lpc) {
      const int prec = (int)r.bits(4) + 1;
      if (prec == 16) { err = "invalid LPC precision"; return false; }
      shift = r.sbits(5);
      if (shift < 0) { err = "negative LPC shift"; return false; }
      for (int i = 0; i < order; ++i) coef[i] = r.sbits(prec);
    }
    // residual
    const int method = (int)r.bits(2);
    if (method > 1) { err = "reserved residual coding method"; return false; }
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int po = (int)r.bits(4);
    const int parts = 1 << po;
    if ((bs >> po) << po != bs && po > 0) { err = "block size not divisible by the partition count"; return false; }
    if ((bs >> po) < order && po > 0) { err = "partition shorter than the predictor order"; return false; }
    int i = order;
    for (int part = 0; part < parts; ++part) {
      const int count = (bs >> po) - (part == 0 ? order : 0);
      const int k = (int)r.bits(pbits);
      if (k == esc) {
        const int raw = (int)r.bits(5);
        for (int j = 0; j < count; ++j) out[i++] = raw ? r.sbits(raw) : 0;
      } else {
        for (int j = 0; j < count; ++j) {
          const uint64_t q = r.unary();
          const uint64_t u = (q << k) | r.bits(k);
          out[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1u);
        }
      }
      if (r.bad) { err = "truncated residual"; return false; }
    }
    // prediction
    if (!lpc) {
      switch (order) {
        case 0: break;
        case 1: for (int t = 1; t < bs; ++t) out[t] += out[t - 1]; break;
        case 2: for (int t = 2; t < bs; ++t) out[t] += 2 * out[t - 1] - out[t - 2]; break;
        case 3: for (int t = 3; t < bs; ++t) out[t] += 3 * out[t - 1] - 3 * out[t - 2] + out[t - 3]; break;
        case 4: for (int t = 4; t < bs; ++t) out[t] += 4 * out[t - 1] - 6 * out[t - 2] + 4 * out[t - 3] - out[t - 4]; break;
      }
    } else {
      for (int t = order; t < bs; ++t) {
        int64_t acc = 0;
        for (int j = 0; j < order; ++j) acc += (int64_t)coef[j] * out[t - 1 - j];
        out[t] += acc >> shift;
      }
    }
  }
  if (wasted)
    for (int i = 0; i < bs; ++i) out[i] = (int64_t)((uint64_t)out[i] << wasted);
  if (r.bad) { err = "truncated subframe"; return false; }
  return true;
}

// Decode every frame.  sink(frame samples: ch pointers, block size) is called per frame with int64 samples per channel.
template <typename Sink> inline bool decode_frames(const uint8_t* d, size_t n, const Info& info, Sink&& sink, int64_t* n_decoded, std::string& err) {
  size_t pos = info.audio_offset;
  std::vector<int64_t> buf((size_t)8 * 65536);
  int64_t done = 0;
  while (pos + 2 <= n) {
    if (info.total_samples > 0 && done >= info.total_samples) break;
    if (!(d[pos] == 0xff && (d[pos + 1] & 0xfe) == 0xf8)) {            // trailing bytes that are not a frame (e.g. an ID3v1 tag) end the stream
      if (done > 0 && info.total_samples == 0) break;
      err = "lost frame synchronisation";
      return false;
    }
    Reader r(d, n, pos);
    r.bits(15);
    r.bits(1);                                                          // blocking strategy (the coded number is not needed to decode in order)
    const int bs_code = (int)r.bits(4), sr_code = (int)r.bits(4), ch_code = (int)r.bits(4), ss_code = (int)r.bits(3);
    if (r.bits(1) != 0) { err = "reserved frame header bit set"; return false; }
    {                                                                   // UTF-8 style coded frame / sample number
      const uint32_t first = r.bits(8);
      int extra = 0;
      if (first >= 0xfe) { if (first == 0xff) { err = "invalid coded number"; return false; } extra = 6; }
      else if (first >= 0xfc) extra = 5;
      else if (first >= 0xf8) extra = 4;
      else if (first >= 0xf0) extra = 3;
      else if (first >= 0xe0) extra = 2;
      else if (first >= 0xc0) extra = 1;
      else if (first >= 0x80) { err = "invalid coded number"; return false; }
      for (int i = 0; i < extra; ++i)
        if ((r.bits(8) & 0xc0) != 0x80) { err = "invalid coded number"; return false; }
    }
    int bs;
    if (bs_code == 0) { err = "reserved block size code"; return false; }
    else if (bs_code == 1) bs = 192;
    else if (bs_code <= 5) bs = 576 << (bs_code - 2);
    else if (bs_code == 6) bs = (int)r.bits(8) + 1;
    else if (bs_code == 7) bs = (int)r.bits(16) + 1;
    else bs = 256 << (bs_code - 8);
    if (sr_code == 12) r.bits(8);
    else if (sr_code == 13 || sr_code == 14) r.bits(16);
    else if (sr_code == 15) { err = "invalid sample rate code"; return false; }
    if (r.bad) { err = "truncated frame header"; return false; }
    const size_t hdr_end = r.byte_pos();
    if (hdr_end + 1 > n || crc8(d + pos, hdr_end - pos) != d[hdr_end]) { err = "frame header CRC-8 mismatch"; return false; }
    r.bits(8);
    int bits = info.bits;
    switch (ss_code) {
      case 0: break;
      case 1: bits = 8; break;
      case 2: bits = 12; break;
      case 4: bits = 16; break;
      case 5: bits = 20; break;
      case 6: bits = 24; break;
      case 7: bits = 32; break;
      default: err = "reserved sample size code"; return false;
    }
    int nch;
    if (ch_code < 8) nch = ch_code + 1;
    else if (ch_code <= 10) nch = 2;
    else { err = "reserved channel assignment"; return false; }
    if (nch != info.channels || bits != info.bits) { err = "frame format differs from STREAMINFO"; return false; }
    if ((size_t)nch * bs > buf.size()) buf.resize((size_t)nch * bs);
    for (int c = 0; c < nch; ++c) {
      const int side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
      if (!decode_subframe(r, bs, bits + side, buf.data() + (size_t)c * bs, err)) return false;
    }
    int64_t* c0 = buf.data();
    int64_t* c1 = buf.data() + bs;
    if (ch_code == 8) for (int i = 0; i < bs; ++i) c1[i] = c0[i] - c1[i];
    else if (ch_code == 9) for (int i = 0; i < bs; ++i) c0[i] = c1[i] + c0[i];
    else if (ch_code == 10)
      for (int i = 0; i < bs; ++i) {
        const int64_t side = c1[i], mid = (int64_t)(((uint64_t)c0[i] << 1) | (uint64_t)(side & 1));
        c0[i] = (mid + side) >> 1;
        c1[i] = (mid - side) >> 1;
      }
    r.align();
    const size_t body_end = r.byte_pos();
    if (body_end + 2 > n) { err = "truncated frame"; return false; }
    const uint16_t want = (uint16_t)((d[body_end] << 8) | d[body_end + 1]);
    if (crc16(d + pos, body_end - pos) != want) { err = "frame CRC-16 mismatch"; return false; }
    int take = bs;
    if (info.total_samples > 0 && done + take > info.total_samples) take = (int)(info.total_samples - done);
    sink(buf.data(), bs, take, nch);
    done += take;
    pos = body_end + 2;
  }
  if (info.total_samples > 0 && done != info.total_samples) { err = "stream ends before STREAMINFO's sample count"; return false; }
  *n_decoded = done;
  return true;
}

}  // namespace ssr_flac
