// Minimal PNG reader for the example CLIs (the reference's examples use cv::imread, examples/resume.cpp:9;
// OpenCV is not a dependency here).  Non-interlaced PNGs, colour types 0/2/3/4/6, bit depths 1-16.
// Output is 8-bit BGR, 3 channels, like cv::imread(path, cv::IMREAD_COLOR): alpha dropped, grey replicated,
// 16-bit samples reduced to their high byte.  Needs zlib.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pngdec {

struct Image {
  int rows = 0, cols = 0;
  std::vector<uint8_t> bgr;  // rows * cols * 3
};

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline Image decode(const std::vector<uint8_t>& file) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file");
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte;
  size_t p = 8;
  bool end = false;
  while (!end && p + 12 <= file.size()) {
    const uint32_t len = be32(&file[p]);
    const char* type = reinterpret_cast<const char*>(&file[p + 4]);
    if (p + 12 + (size_t)len > file.size()) throw std::runtime_error("truncated PNG chunk");
    const uint8_t* d = &file[p + 8];
    if (!memcmp(type, "IHDR", 4)) {
      if (len < 13) throw std::runtime_error("bad IHDR");
      w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
    } else if (!memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
    else if (!memcmp(type, "IEND", 4)) end = true;
    p += 12 + (size_t)len;
  }
  if (ctype < 0 || w == 0 || h == 0 || w > 65535 || h > 65535) throw std::runtime_error("bad PNG header");
  if (interlace) throw std::runtime_error("interlaced PNGs are not supported");
  int ch;
  switch (ctype) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: throw std::runtime_error("bad PNG colour type");
  }
  if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) throw std::runtime_error("bad PNG bit depth");
  const size_t bpp_bits = (size_t)ch * depth, stride = (w * bpp_bits + 7) / 8, bpp = bpp_bits < 8 ? 1 : bpp_bits / 8;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf rawlen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) throw std::runtime_error("PNG inflate failed");
  // undo the scanline filters in place
  std::vector<uint8_t> prev(stride, 0);
  for (uint32_t y = 0; y < h; ++y) {
    uint8_t* line = &raw[y * (stride + 1)];
    const int ft = line[0];
    uint8_t* cur = line + 1;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int pred = 0;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
          const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: throw std::runtime_error("bad PNG filter");
      }
      cur[i] = (uint8_t)(cur[i] + pred);
    }
    memcpy(prev.data(), cur, stride);
  }
  Image img;
  img.rows = (int)h; img.cols = (int)w;
  img.bgr.resize((size_t)h * w * 3);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* cur = &raw[y * (stride + 1) + 1];
    for (uint32_t x = 0; x < w; ++x) {
      auto sample = [&](int k) -> int {   // k-th channel of pixel x as 8 bits (index for palette images)
        if (depth == 8) return cur[(size_t)x * ch + k];
        if (depth == 16) return cur[((size_t)x * ch + k) * 2];
        const size_t bit = (size_t)x * depth;
        const int v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
        return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
      };
      int r, g, b;
      if (ctype == 3) {
        const size_t i = (size_t)sample(0) * 3;
        if (i + 2 >= plte.size()) throw std::runtime_error("palette index out of range");
        r = plte[i]; g = plte[i + 1]; b = plte[i + 2];
      } else if (ch <= 2) r = g = b = sample(0);
      else { r = sample(0); g = sample(1); b = sample(2); }
      uint8_t* o = &img.bgr[((size_t)y * w + x) * 3];
      o[0] = (uint8_t)b; o[1] = (uint8_t)g; o[2] = (uint8_t)r;
    }
  }
  return img;
}

inline Image read(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  fclose(f);
  return decode(buf);
}

}  // namespace pngdec
