// image_io.cpp — PNG / PPM / BMP -> BGR, PFM in/out.  See image_io.h for the role in the file-list feeder.
#include "image_io.h"

#include <cstdint>
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace hobot {
namespace stereonet {

namespace {

bool fail(std::string* err, const std::string& what) {
  if (err) *err = what;
  return false;
}

bool slurp(const std::string& path, std::vector<uint8_t>& out) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f.good()) return false;
  const std::streamoff n = f.tellg();
  if (n < 0) return false;
  out.resize((size_t)n);
  f.seekg(0);
  f.read(reinterpret_cast<char*>(out.data()), n);
  return (std::streamoff)f.gcount() == n;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
uint32_t le32(const uint8_t* p) { return ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0]; }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// ---- PNG ---------------------------------------------------------------------------------------------
int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Header sizes come from user-supplied files (the list feeder): bound them before any size arithmetic so that
// (size_t)w * h * k can neither wrap nor ask for an absurd allocation (a bad_alloc would kill the node).
constexpr int kMaxImageDim = 16384;
inline bool dims_ok(long w, long h) { return w > 0 && h > 0 && w <= kMaxImageDim && h <= kMaxImageDim; }

bool decode_png(const std::vector<uint8_t>& file, int& w, int& h, std::vector<uint8_t>& bgr, std::string* err) {
  size_t pos = 8;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte;
  bool seen_end = false;
  while (pos + 12 <= file.size() && !seen_end) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t* tag = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) return fail(err, "png: truncated chunk");
    const uint8_t* body = &file[pos + 8];
    if (!memcmp(tag, "IHDR", 4)) {
      if (len < 13) return fail(err, "png: bad IHDR");
      w = (int)be32(body);
      h = (int)be32(body + 4);
      depth = body[8];
      ctype = body[9];
      interlace = body[12];
    } else if (!memcmp(tag, "PLTE", 4)) {
      plte.assign(body, body + len);
    } else if (!memcmp(tag, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(tag, "IEND", 4)) {
      seen_end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (ctype < 0 || w <= 0 || h <= 0) return fail(err, "png: no IHDR");
  if (!dims_ok(w, h)) return fail(err, "png: image larger than 16384 x 16384");
  const bool packed = depth == 1 || depth == 2 || depth == 4;
  if (depth != 8 && !(packed && (ctype == 0 || ctype == 3)))
    return fail(err, "png: only 8-bit samples (or packed gray / palette) are supported for colour input");
  if (interlace) return fail(err, "png: interlaced files are not supported");
  int spp;   // samples per pixel
  switch (ctype) {
    case 0: spp = 1; break;
    case 2: spp = 3; break;
    case 3: spp = 1; break;
    case 4: spp = 2; break;
    case 6: spp = 4; break;
    default: return fail(err, "png: bad colour type");
  }
  if (ctype == 3 && plte.size() < 3) return fail(err, "png: palette image without PLTE");
  const size_t stride = ((size_t)w * spp * depth + 7) / 8;
  const size_t fdist = packed ? 1 : (size_t)spp;   // filter distance in bytes
  std::vector<uint8_t> raw((stride + 1) * (size_t)h);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
    return fail(err, "png: inflate failed");
  // undo the per-row filters in place (row r lives at raw[r*(stride+1)+1 ..])
  std::vector<uint8_t> zero(stride, 0);
  for (int r = 0; r < h; ++r) {
    uint8_t* cur = &raw[(size_t)r * (stride + 1) + 1];
    const uint8_t* up = r ? cur - (stride + 1) : zero.data();
    const int ft = cur[-1];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= fdist ? cur[i - fdist] : 0;
      const int b = up[i];
      const int c = i >= fdist ? up[i - fdist] : 0;
      int v = cur[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: return fail(err, "png: bad filter type");
      }
      cur[i] = (uint8_t)v;
    }
  }
  bgr.resize((size_t)w * h * 3);
  for (int r = 0; r < h; ++r) {
    const uint8_t* s = &raw[(size_t)r * (stride + 1) + 1];
    uint8_t* d = &bgr[(size_t)r * w * 3];
    for (int x = 0; x < w; ++x, s += (packed ? 0 : spp), d += 3) {
      uint8_t R, G, B;
      int sample = s[0];
      if (packed) {   // most significant bits first
        const int per = 8 / depth, sh = (per - 1 - x % per) * depth;
        sample = (s[x / per] >> sh) & ((1 << depth) - 1);
      }
      if (ctype == 0 || ctype == 4) {
        R = G = B = (uint8_t)(packed ? sample * 255 / ((1 << depth) - 1) : sample);
      } else if (ctype == 3) {
        const size_t k = (size_t)sample * 3;
        if (k + 3 > plte.size()) return fail(err, "png: palette index out of range");
        R = plte[k];
        G = plte[k + 1];
        B = plte[k + 2];
      } else {
        R = s[0];
        G = s[1];
        B = s[2];
      }
      d[0] = B;
      d[1] = G;
      d[2] = R;
    }
  }
  return true;
}

// ---- PPM / PGM ---------------------------------------------------------------------------------------
bool pnm_token(const std::vector<uint8_t>& f, size_t& pos, int& value) {
  for (;;) {
    while (pos < f.size() && isspace(f[pos])) ++pos;
    if (pos < f.size() && f[pos] == '#') {
      while (pos < f.size() && f[pos] != '\n') ++pos;
      continue;
    }
    break;
  }
  if (pos >= f.size() || !isdigit(f[pos])) return false;
  long v = 0;
  while (pos < f.size() && isdigit(f[pos])) {
    v = v * 10 + (f[pos++] - '0');
    if (v > 1000000000L) return false;          // would not fit an int: reject instead of truncating
  }
  value = (int)v;
  return true;
}

bool decode_pnm(const std::vector<uint8_t>& f, int& w, int& h, std::vector<uint8_t>& bgr, std::string* err) {
  const bool colour = f[1] == '6';
  size_t pos = 2;
  int maxv = 0;
  if (!pnm_token(f, pos, w) || !pnm_token(f, pos, h) || !pnm_token(f, pos, maxv)) return fail(err, "pnm: bad header");
  if (maxv != 255) return fail(err, "pnm: only maxval 255 is supported");
  if (!dims_ok(w, h)) return fail(err, "pnm: bad size (1..16384 per side)");
  ++pos;   // the single whitespace after maxval
  const size_t need = (size_t)w * h * (colour ? 3 : 1);          // <= 3 * 2^28: no wrap
  if (pos > f.size() || need > f.size() - pos) return fail(err, "pnm: truncated");
  bgr.resize((size_t)w * h * 3);
  const uint8_t* s = &f[pos];
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    if (colour) {
      bgr[3 * i] = s[3 * i + 2];
      bgr[3 * i + 1] = s[3 * i + 1];
      bgr[3 * i + 2] = s[3 * i];
    } else {
      bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = s[i];
    }
  }
  return true;
}

// ---- BMP ---------------------------------------------------------------------------------------------
bool decode_bmp(const std::vector<uint8_t>& f, int& w, int& h, std::vector<uint8_t>& bgr, std::string* err) {
  if (f.size() < 54) return fail(err, "bmp: truncated header");
  const uint32_t off = le32(&f[10]);
  const int32_t bw = (int32_t)le32(&f[18]), bh = (int32_t)le32(&f[22]);
  const int bpp = le16(&f[28]);
  const uint32_t comp = le32(&f[30]);
  if ((bpp != 24 && bpp != 32) || (comp != 0 && comp != 3)) return fail(err, "bmp: only uncompressed 24/32-bit is supported");
  w = bw;
  h = bh < 0 ? -bh : bh;
  if (bh == INT32_MIN || !dims_ok(w, h)) return fail(err, "bmp: bad size (1..16384 per side)");
  const size_t bytes = bpp / 8, pitch = ((size_t)w * bytes + 3) & ~(size_t)3;
  if ((size_t)off > f.size() || pitch * (size_t)h > f.size() - (size_t)off) return fail(err, "bmp: truncated pixels");
  bgr.resize((size_t)w * h * 3);
  for (int r = 0; r < h; ++r) {
    const int src_row = bh < 0 ? r : h - 1 - r;   // positive height = bottom-up
    const uint8_t* s = &f[off + pitch * src_row];
    uint8_t* d = &bgr[(size_t)r * w * 3];
    for (int x = 0; x < w; ++x, s += bytes, d += 3) {
      d[0] = s[0];
      d[1] = s[1];
      d[2] = s[2];
    }
  }
  return true;
}

}  // namespace

bool ReadImageBGR(const std::string& path, int& w, int& h, std::vector<uint8_t>& bgr, std::string* err) {
  std::vector<uint8_t> f;
  if (!slurp(path, f)) return fail(err, "cannot read " + path);
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (f.size() >= 8 && !memcmp(f.data(), png_sig, 8)) return decode_png(f, w, h, bgr, err);
  if (f.size() >= 3 && f[0] == 'P' && (f[1] == '6' || f[1] == '5')) return decode_pnm(f, w, h, bgr, err);
  if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') return decode_bmp(f, w, h, bgr, err);
  return fail(err, "unsupported image format: " + path);
}

bool WriteBytes(const std::string& path, const void* data, size_t n) {
  std::ofstream o(path, std::ios::binary);
  if (!o.good()) return false;
  o.write(reinterpret_cast<const char*>(data), (std::streamsize)n);
  return o.good();
}

bool WritePFM(const std::string& path, const float* data, int w, int h) {
  std::ofstream o(path, std::ios::binary);
  if (!o.good()) return false;
  o << "Pf\n" << w << " " << h << "\n-1.0\n";
  for (int r = h - 1; r >= 0; --r) o.write(reinterpret_cast<const char*>(data + (size_t)r * w), sizeof(float) * w);
  return o.good();
}

bool ReadPFM(const std::string& path, int& w, int& h, std::vector<float>& data, std::string* err) {
  std::ifstream f(path, std::ios::binary);
  if (!f.good()) return fail(err, "cannot read " + path);
  std::string magic;
  double scale = 0;
  f >> magic >> w >> h >> scale;
  if (magic != "Pf" || !dims_ok(w, h) || scale == 0) return fail(err, "pfm: bad header (single-channel Pf expected, 1..16384 per side)");
  f.get();   // the newline that ends the header
  data.resize((size_t)w * h);
  for (int r = h - 1; r >= 0; --r) {
    f.read(reinterpret_cast<char*>(&data[(size_t)r * w]), sizeof(float) * w);
    if ((size_t)f.gcount() != sizeof(float) * w) return fail(err, "pfm: truncated");
  }
  if (scale > 0)   // big-endian samples
    for (float& v : data) {
      uint8_t b[4];
      memcpy(b, &v, 4);
      const uint8_t s[4] = {b[3], b[2], b[1], b[0]};
      memcpy(&v, s, 4);
    }
  return true;
}

}  // namespace stereonet
}  // namespace hobot
