// png_gray8.h -- minimal PNG reader of the file-based odometry entry: 8-bit grayscale, non-interlaced (what MulRan's
// polar_oxford_form holds; reference README.md:54-60).  zlib inflates; the per-row filters are undone by one loop per
// filter type straight into the caller's buffer (round 6: the first reader ran a `switch` per PIXEL and went through three
// temporary vectors per image -- 10.9 ms per 400 x 3371 scan and thread, of which the inflate is about 4).
#ifndef RSX_HOST_PNG_GRAY8_H
#define RSX_HOST_PNG_GRAY8_H

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rsxhost {

struct PngScratch {  // per thread, reused from image to image
  std::vector<uint8_t> file, idat, raw, zero_row;
};

inline uint32_t png_be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

// the image's rows, h x w bytes, into dst (dst_capacity bytes; nullptr: only *width / *height are set).  Throws std::runtime_error.
inline void read_png_gray8_into(const std::string &path, uint8_t *dst, size_t dst_capacity, int *width, int *height, PngScratch &s) {
  auto bad = [&](const char *what) { throw std::runtime_error(path + ": " + what); };
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) bad("cannot open");
  std::fseek(f, 0, SEEK_END);
  const long fsz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (fsz < 8) {
    std::fclose(f);
    bad("not a PNG");
  }
  s.file.resize((size_t)fsz);
  const size_t got = std::fread(s.file.data(), 1, (size_t)fsz, f);
  std::fclose(f);
  if (got != (size_t)fsz) bad("short read");
  const std::vector<uint8_t> &buf = s.file;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (std::memcmp(buf.data(), sig, 8) != 0) bad("not a PNG");
  size_t pos = 8;
  int w = 0, h = 0;
  const uint8_t *one_idat = nullptr;  // the usual case is a single IDAT chunk or a few: collect only when there are several
  size_t one_len = 0;
  int n_idat = 0;
  s.idat.clear();
  while (pos + 12 <= buf.size()) {
    const uint32_t len = png_be32(&buf[pos]);
    const char *type = reinterpret_cast<const char *>(&buf[pos + 4]);
    if (pos + 12 + (size_t)len > buf.size()) bad("truncated chunk");
    const uint8_t *data = &buf[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) bad("short IHDR");
      w = (int)png_be32(data);
      h = (int)png_be32(data + 4);
      if (data[8] != 8 || data[9] != 0 || data[12] != 0) bad("only 8-bit grayscale non-interlaced PNG is supported");
    } else if (!std::memcmp(type, "IDAT", 4)) {
      if (n_idat == 0) {
        one_idat = data;
        one_len = len;
      } else {
        if (n_idat == 1) s.idat.assign(one_idat, one_idat + one_len);
        s.idat.insert(s.idat.end(), data, data + len);
      }
      n_idat++;
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (w <= 0 || h <= 0) bad("no IHDR");
  *width = w;
  *height = h;
  if (!dst) return;
  if ((size_t)h * (size_t)w > dst_capacity) bad("image larger than the buffer");
  if (n_idat == 0) bad("no IDAT");
  const uint8_t *z = n_idat == 1 ? one_idat : s.idat.data();
  const size_t zlen = n_idat == 1 ? one_len : s.idat.size();
  s.raw.resize((size_t)h * ((size_t)w + 1));
  uLongf out_len = (uLongf)s.raw.size();
  if (uncompress(s.raw.data(), &out_len, z, (uLong)zlen) != Z_OK || out_len != s.raw.size()) bad("inflate failed");
  s.zero_row.assign((size_t)w, 0);
  for (int y = 0; y < h; y++) {  // undo the per-row filters (bpp = 1): one loop per filter type
    const uint8_t ft = s.raw[(size_t)y * ((size_t)w + 1)];
    const uint8_t *in = &s.raw[(size_t)y * ((size_t)w + 1) + 1];
    uint8_t *out = dst + (size_t)y * (size_t)w;
    const uint8_t *up = y ? out - w : s.zero_row.data();
    switch (ft) {
      case 0: std::memcpy(out, in, (size_t)w); break;
      case 1: {
        unsigned a = 0;
        for (int x = 0; x < w; x++) {
          a = (in[x] + a) & 0xffu;
          out[x] = (uint8_t)a;
        }
        break;
      }
      case 2:
        for (int x = 0; x < w; x++) out[x] = (uint8_t)(in[x] + up[x]);
        break;
      case 3: {
        unsigned a = 0;
        for (int x = 0; x < w; x++) {
          a = (in[x] + ((a + up[x]) >> 1)) & 0xffu;
          out[x] = (uint8_t)a;
        }
        break;
      }
      case 4: {
        int a = 0, c = 0;
        for (int x = 0; x < w; x++) {
          const int b = up[x];
          const int p = b - c, q = a - c;  // p = (a + b - c) - a, q = (a + b - c) - b
          const int pa = std::abs(p), pb = std::abs(q), pc = std::abs(p + q);
          const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          a = (in[x] + pred) & 0xff;
          out[x] = (uint8_t)a;
          c = b;
        }
        break;
      }
      default: bad("bad PNG filter");
    }
  }
}

inline std::vector<uint8_t> read_png_gray8(const std::string &path, int *width, int *height) {
  PngScratch s;
  read_png_gray8_into(path, nullptr, 0, width, height, s);
  std::vector<uint8_t> img((size_t)*height * (size_t)*width);
  read_png_gray8_into(path, img.data(), img.size(), width, height, s);
  return img;
}

}  // namespace rsxhost

#endif
