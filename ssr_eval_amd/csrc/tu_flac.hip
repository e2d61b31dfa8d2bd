// libssrhip.so translation unit: FLAC ingest - host code behind the C ABI (ssr_flac.h).
#include <cstdio>

#include "ssr_flac.h"
#include "ssr_host.h"

namespace {

struct FileBytes {
  std::vector<uint8_t> data;
  bool read(const char* path, std::string& err) {
    FILE* f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); err = "seek failed"; return false; }
    const long n = ftell(f);
    if (n < 0) { fclose(f); err = "tell failed"; return false; }
    rewind(f);
    data.resize((size_t)n);
    const size_t got = n ? fread(data.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if (got != (size_t)n) { err = "short read"; return false; }
    return true;
  }
};

// T = int16_t (bits <= 16 only) or int32_t; out may be null (count / verify only)
template <typename T>
int decode_file(const char* path, T* out, int64_t capacity, int verify_md5, int64_t* frames_out, int max_bits) {
  if (!path) return ssr_fail(SSR_ERR_INVALID_ARG, "null path");
  FileBytes fb;
  std::string err;
  if (!fb.read(path, err)) return ssr_fail(SSR_ERR_INVALID_ARG, err);
  ssr_flac::Info info;
  if (!ssr_flac::parse_header(fb.data.data(), fb.data.size(), info, err)) return ssr_fail(SSR_ERR_INVALID_ARG, std::string(path) + ": " + err);
  if (info.bits > max_bits) return ssr_fail(SSR_ERR_UNSUPPORTED, std::string(path) + ": more bits per sample than this entry point returns");
  ssr_flac::Md5 md5;
  const bool hash = verify_md5 && info.has_md5;
  const int bytes_per = (info.bits + 7) / 8;
  std::vector<uint8_t> packed;
  int64_t written = 0;
  bool overflow = false;
  auto sink = [&](const int64_t* ch, int bs, int take, int nch) {
    if (hash) {
      packed.resize((size_t)take * nch * bytes_per);
      uint8_t* q = packed.data();
      for (int i = 0; i < take; ++i)
        for (int c = 0; c < nch; ++c) {
          const int64_t v = ch[(size_t)c * bs + i];
          for (int b = 0; b < bytes_per; ++b) *q++ = (uint8_t)((uint64_t)v >> (8 * b));
        }
      md5.update(packed.data(), packed.size());
    }
    if (out) {
      if ((written + take) * nch > capacity) { overflow = true; }
      else
        for (int i = 0; i < take; ++i)
          for (int c = 0; c < nch; ++c) out[(written + i) * nch + c] = (T)ch[(size_t)c * bs + i];
    }
    written += take;
  };
  int64_t n = 0;
  if (!ssr_flac::decode_frames(fb.data.data(), fb.data.size(), info, sink, &n, err)) return ssr_fail(SSR_ERR_INVALID_ARG, std::string(path) + ": " + err);
  if (frames_out) *frames_out = n;
  if (overflow) return ssr_fail(SSR_ERR_WORKSPACE, std::string(path) + ": output buffer too small");
  if (hash) {
    uint8_t got[16];
    md5.finish(got);
    if (memcmp(got, info.md5, 16) != 0)
      return ssr_fail(SSR_ERR_INVALID_ARG, std::string(path) + ": decoded PCM does not match the MD5 signature in STREAMINFO");
  }
  return SSR_OK;
}

}  // namespace

extern "C" int ssr_flac_info(const char* path, int* sample_rate, int* channels, int* bits, int64_t* total_samples, int* has_md5) {
  if (!path) return ssr_fail(SSR_ERR_INVALID_ARG, "null path");
  FILE* f = fopen(path, "rb");
  if (!f) return ssr_fail(SSR_ERR_INVALID_ARG, std::string("cannot open ") + path);
  std::vector<uint8_t> head((size_t)1 << 16);                       // STREAMINFO is the first block; an ID3v2 tag may precede it
  size_t got = fread(head.data(), 1, head.size(), f);
  ssr_flac::Info info;
  std::string err;
  bool ok = ssr_flac::parse_header(head.data(), got, info, err);
  if (!ok && got == head.size()) {                                   // large metadata (pictures): read the whole file
    fclose(f);
    FileBytes fb;
    if (!fb.read(path, err)) return ssr_fail(SSR_ERR_INVALID_ARG, err);
    ok = ssr_flac::parse_header(fb.data.data(), fb.data.size(), info, err);
    f = nullptr;
  }
  if (f) fclose(f);
  if (!ok) return ssr_fail(SSR_ERR_INVALID_ARG, std::string(path) + ": " + err);
  if (sample_rate) *sample_rate = info.sample_rate;
  if (channels) *channels = info.channels;
  if (bits) *bits = info.bits;
  if (total_samples) *total_samples = info.total_samples;
  if (has_md5) *has_md5 = info.has_md5 ? 1 : 0;
  return SSR_OK;
}

extern "C" int ssr_flac_decode_pcm16(const char* path, int16_t* out, int64_t capacity, int verify_md5, int64_t* frames_out) {
  return decode_file<int16_t>(path, out, capacity, verify_md5, frames_out, 16);
}

extern "C" int ssr_flac_decode_i32(const char* path, int32_t* out, int64_t capacity, int verify_md5, int64_t* frames_out) {
  return decode_file<int32_t>(path, out, capacity, verify_md5, frames_out, 32);
}
