// jpeg_nv12.cpp — see include/jpeg_nv12.h.  Baseline encoder: 16x16 MCUs (4 Y + Cb + Cr blocks), standard
// quantisation tables scaled by the libjpeg quality rule, standard Huffman tables.
//   EncodeNv12ToJpeg           the product form: AAN 8x8 forward DCT on 8 columns at a time (the scale factors folded
//                              into the reciprocal quantisers), table-driven entropy coder with a 64-bit bit buffer
//                              writing into a pre-sized array (round 4: the node's left-eye JPEG, stereonet_node.cpp:
//                              749-786, was 40 ms per 1280x720 frame and capped the drop-in node near 25 frames/s)
//   EncodeNv12ToJpegReference  the round-3 form (exact separable float DCT, one push_back per byte): the checker the
//                              tests compare the fast form against (same headers; coefficients may differ by one
//                              quantisation step where a value sits on a rounding boundary)
#include "jpeg_nv12.h"

#include <cmath>
#include <cstring>

namespace hobot {
namespace stereonet {
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kQLum[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                           14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                           18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                           49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQChr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                           99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                           99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct Huff {
  uint16_t code[256];
  uint8_t len[256];
};

void build_huff(const uint8_t* bits, const uint8_t* vals, Huff* h) {
  memset(h, 0, sizeof *h);
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < bits[l - 1]; ++i) {
      h->code[vals[k]] = (uint16_t)code++;
      h->len[vals[k]] = (uint8_t)l;
      ++k;
    }
    code <<= 1;
  }
}

struct BitWriter {
  std::vector<uint8_t>& out;
  uint32_t acc = 0;
  int nbits = 0;
  explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
  void put(uint32_t code, int len) {
    acc = (acc << len) | (code & ((1u << len) - 1));
    nbits += len;
    while (nbits >= 8) {
      const uint8_t b = (uint8_t)(acc >> (nbits - 8));
      out.push_back(b);
      if (b == 0xFF) out.push_back(0x00);
      nbits -= 8;
    }
  }
  void flush() {
    if (nbits > 0) put((1u << (8 - nbits)) - 1, 8 - nbits);
  }
};

void fdct8x8(const float* in, float* out) {
  static float c[8][8];
  static bool init = false;
  if (!init) {
    for (int u = 0; u < 8; ++u)
      for (int x = 0; x < 8; ++x)
        c[u][x] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f);
    init = true;
  }
  float tmp[64];
  for (int y = 0; y < 8; ++y)
    for (int u = 0; u < 8; ++u) {
      float s = 0.f;
      for (int x = 0; x < 8; ++x) s += c[u][x] * in[y * 8 + x];
      tmp[y * 8 + u] = s;
    }
  for (int v = 0; v < 8; ++v)
    for (int u = 0; u < 8; ++u) {
      float s = 0.f;
      for (int y = 0; y < 8; ++y) s += c[v][y] * tmp[y * 8 + u];
      out[v * 8 + u] = s;
    }
}

void encode_block(const float* px, const uint8_t* q, const Huff& dc, const Huff& ac, int* last_dc, BitWriter* bw) {
  float coef[64];
  fdct8x8(px, coef);
  int zz[64];
  for (int i = 0; i < 64; ++i) zz[i] = (int)std::lrintf(coef[kZigzag[i]] / (float)q[kZigzag[i]]);
  auto magnitude = [](int v, int* nb, uint32_t* bits) {
    int a = v < 0 ? -v : v, n = 0;
    while (a) {
      ++n;
      a >>= 1;
    }
    *nb = n;
    *bits = (uint32_t)(v < 0 ? v + (1 << n) - 1 : v);
  };
  int nb;
  uint32_t bits;
  const int diff = zz[0] - *last_dc;
  *last_dc = zz[0];
  magnitude(diff, &nb, &bits);
  bw->put(dc.code[nb], dc.len[nb]);
  if (nb) bw->put(bits, nb);
  int run = 0;
  for (int i = 1; i < 64; ++i) {
    if (zz[i] == 0) {
      ++run;
      continue;
    }
    while (run > 15) {
      bw->put(ac.code[0xF0], ac.len[0xF0]);
      run -= 16;
    }
    magnitude(zz[i], &nb, &bits);
    const int sym = (run << 4) | nb;
    bw->put(ac.code[sym], ac.len[sym]);
    bw->put(bits, nb);
    run = 0;
  }
  if (run) bw->put(ac.code[0x00], ac.len[0x00]);
}

void put16(std::vector<uint8_t>& o, int v) {
  o.push_back((uint8_t)(v >> 8));
  o.push_back((uint8_t)v);
}

}  // namespace

bool EncodeNv12ToJpegReference(const uint8_t* nv12, int w, int h, int pitch, int quality, std::vector<uint8_t>& out) {
  if (!nv12 || w <= 0 || h <= 0 || (w & 1) || (h & 1) || pitch < w) return false;
  quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
  const int sf = quality < 50 ? 5000 / quality : 200 - quality * 2;
  uint8_t ql[64], qc[64];
  for (int i = 0; i < 64; ++i) {
    int a = (kQLum[i] * sf + 50) / 100, b = (kQChr[i] * sf + 50) / 100;
    ql[i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a));
    qc[i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
  }
  Huff dcl, dcc, acl, acc;
  build_huff(kDcLumBits, kDcVals, &dcl);
  build_huff(kDcChrBits, kDcVals, &dcc);
  build_huff(kAcLumBits, kAcLumVals, &acl);
  build_huff(kAcChrBits, kAcChrVals, &acc);

  out.clear();
  out.reserve((size_t)w * h / 4);
  const uint8_t soi_app0[] = {0xFF, 0xD8, 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
  out.insert(out.end(), soi_app0, soi_app0 + sizeof soi_app0);
  for (int t = 0; t < 2; ++t) {   // DQT
    out.push_back(0xFF);
    out.push_back(0xDB);
    put16(out, 67);
    out.push_back((uint8_t)t);
    for (int i = 0; i < 64; ++i) out.push_back((t ? qc : ql)[kZigzag[i]]);
  }
  out.push_back(0xFF);   // SOF0: 8 bit, 3 components, Y 2x2, Cb/Cr 1x1
  out.push_back(0xC0);
  put16(out, 17);
  out.push_back(8);
  put16(out, h);
  put16(out, w);
  out.push_back(3);
  const uint8_t comps[9] = {1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1};
  out.insert(out.end(), comps, comps + 9);
  auto dht = [&](int cls_id, const uint8_t* bits, const uint8_t* vals, int nvals) {
    out.push_back(0xFF);
    out.push_back(0xC4);
    put16(out, 3 + 16 + nvals);
    out.push_back((uint8_t)cls_id);
    out.insert(out.end(), bits, bits + 16);
    out.insert(out.end(), vals, vals + nvals);
  };
  dht(0x00, kDcLumBits, kDcVals, 12);
  dht(0x10, kAcLumBits, kAcLumVals, 162);
  dht(0x01, kDcChrBits, kDcVals, 12);
  dht(0x11, kAcChrBits, kAcChrVals, 162);
  const uint8_t sos[] = {0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
  out.insert(out.end(), sos, sos + sizeof sos);

  BitWriter bw(out);
  int dc_y = 0, dc_cb = 0, dc_cr = 0;
  const uint8_t* uv = nv12 + (size_t)h * pitch;
  const int cw = w / 2, chh = h / 2;
  float blk[64];
  for (int my = 0; my < h; my += 16)
    for (int mx = 0; mx < w; mx += 16) {
      for (int b = 0; b < 4; ++b) {
        const int by = my + (b >> 1) * 8, bx = mx + (b & 1) * 8;
        for (int y = 0; y < 8; ++y) {
          const int sy = by + y < h ? by + y : h - 1;
          for (int x = 0; x < 8; ++x) {
            const int sx = bx + x < w ? bx + x : w - 1;
            blk[y * 8 + x] = (float)nv12[(size_t)sy * pitch + sx] - 128.0f;
          }
        }
        encode_block(blk, ql, dcl, acl, &dc_y, &bw);
      }
      for (int comp = 0; comp < 2; ++comp) {
        for (int y = 0; y < 8; ++y) {
          const int sy = my / 2 + y < chh ? my / 2 + y : chh - 1;
          for (int x = 0; x < 8; ++x) {
            const int sx = mx / 2 + x < cw ? mx / 2 + x : cw - 1;
            blk[y * 8 + x] = (float)uv[(size_t)sy * pitch + 2 * sx + comp] - 128.0f;
          }
        }
        encode_block(blk, qc, dcc, acc, comp ? &dc_cr : &dc_cb, &bw);
      }
    }
  bw.flush();
  out.push_back(0xFF);
  out.push_back(0xD9);
  return true;
}


// ---------------------------------------------------------------------------------------------------------------
// Fast form
// ---------------------------------------------------------------------------------------------------------------
namespace {

// One AAN (Arai / Agui / Nakajima) 8-point DCT along the FIRST index of d[8][8], for all 8 second-index positions at
// once (the loop over c is what the compiler vectorises).  Outputs carry the AAN scale aan[u] (u = first index).
inline void aan_pass(float (*d)[8]) {
  for (int c = 0; c < 8; ++c) {
    const float t0 = d[0][c] + d[7][c], t7 = d[0][c] - d[7][c];
    const float t1 = d[1][c] + d[6][c], t6 = d[1][c] - d[6][c];
    const float t2 = d[2][c] + d[5][c], t5 = d[2][c] - d[5][c];
    const float t3 = d[3][c] + d[4][c], t4 = d[3][c] - d[4][c];
    const float e0 = t0 + t3, e3 = t0 - t3, e1 = t1 + t2, e2 = t1 - t2;
    d[0][c] = e0 + e1;
    d[4][c] = e0 - e1;
    const float z1 = (e2 + e3) * 0.707106781f;
    d[2][c] = e3 + z1;
    d[6][c] = e3 - z1;
    const float o0 = t4 + t5, o1 = t5 + t6, o2 = t6 + t7;
    const float z5 = (o0 - o2) * 0.382683433f;
    const float z2 = 0.541196100f * o0 + z5;
    const float z4 = 1.306562965f * o2 + z5;
    const float z3 = o1 * 0.707106781f;
    const float z11 = t7 + z3, z13 = t7 - z3;
    d[5][c] = z13 + z2;
    d[3][c] = z13 - z2;
    d[1][c] = z11 + z4;
    d[7][c] = z11 - z4;
  }
}

struct FastTables {
  int quality = -1;
  uint8_t ql[64], qc[64];          // natural order
  float rl[64], rc[64];            // reciprocal quantisers in ZIGZAG order, indexed like `src` below
  uint8_t src[64];                 // zigzag position i -> index into the transposed coefficient block
  Huff dcl, dcc, acl, acc;
};

void make_tables(int quality, FastTables* t) {
  static const double aan[8] = {1.0, 1.387039845, 1.306562965, 1.175875602, 1.0, 0.785694958, 0.541196100, 0.275899379};
  const int sf = quality < 50 ? 5000 / quality : 200 - quality * 2;
  for (int i = 0; i < 64; ++i) {
    int a = (kQLum[i] * sf + 50) / 100, b = (kQChr[i] * sf + 50) / 100;
    t->ql[i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a));
    t->qc[i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
  }
  for (int i = 0; i < 64; ++i) {
    const int nat = kZigzag[i], v = nat >> 3, u = nat & 7;      // coefficient (vertical v, horizontal u)
    // the two passes leave coefficient (v, u) at blk[u][v] (vertical pass, transpose, vertical pass again)
    t->src[i] = (uint8_t)(u * 8 + v);
    const double s = aan[u] * aan[v] * 8.0;
    t->rl[i] = (float)(1.0 / (t->ql[nat] * s));
    t->rc[i] = (float)(1.0 / (t->qc[nat] * s));
  }
  build_huff(kDcLumBits, kDcVals, &t->dcl);
  build_huff(kDcChrBits, kDcVals, &t->dcc);
  build_huff(kAcLumBits, kAcLumVals, &t->acl);
  build_huff(kAcChrBits, kAcChrVals, &t->acc);
  t->quality = quality;
}

// Branch-free bit appender: codes go MSB-first into a 64-bit window; after every symbol the window's whole bytes are
// stored (always eight bytes, the pointer advances by the count of complete ones).  Byte stuffing (0x00 after every
// 0xFF) is NOT done here: the unstuffed stream goes to a scratch buffer and stuff_copy() inserts the zeros afterwards —
// on noisy content the data-dependent branches of a stuffing writer cost more than the second pass.
struct FastBits {
  uint8_t* p;
  uint64_t acc = 0;                 // bits [63 .. 64-n] valid
  int n = 0;                        // < 8 between calls
  // len <= 27 (a 16-bit code + an 11-bit magnitude), code < 2^len
  inline void put(uint32_t code, int len) {
    acc |= (uint64_t)code << (64 - n - len);
    n += len;
    const uint64_t be = __builtin_bswap64(acc);
    memcpy(p, &be, 8);
    const int k = n >> 3;
    p += k;
    acc <<= 8 * k;
    n &= 7;
  }
  inline void finish() {            // pad the last partial byte with ones
    if (n > 0) {
      *p++ = (uint8_t)((acc >> 56) | ((1u << (8 - n)) - 1));
      n = 0;
    }
  }
};

// dst <- src with a zero byte after every 0xFF; returns the end of dst
inline uint8_t* stuff_copy(uint8_t* dst, const uint8_t* src, size_t n) {
  while (n) {
    const uint8_t* ff = static_cast<const uint8_t*>(memchr(src, 0xFF, n));
    if (!ff) {
      memcpy(dst, src, n);
      return dst + n;
    }
    const size_t k = (size_t)(ff - src) + 1;
    memcpy(dst, src, k);
    dst += k;
    *dst++ = 0;
    src += k;
    n -= k;
  }
  return dst;
}

// 8x8 block whose top-left sample is (bx, by) of a plane of pw x ph samples at `plane` (row stride `stride`, `step` bytes
// between horizontally adjacent samples): DCT, quantise, entropy-code.  Samples beyond the right / bottom edge replicate
// the last column / row (as the reference form does).
inline void fast_block(const uint8_t* plane, int stride, int step, int bx, int by, int pw, int ph, const float* recip,
                       const uint8_t* src, const Huff& dc, const Huff& ac, int* last_dc, FastBits* bw) {
  alignas(32) float a[8][8], b[8][8];
  if (bx + 8 <= pw && by + 8 <= ph) {
    const uint8_t* px = plane + (size_t)by * stride + (size_t)bx * step;
    for (int y = 0; y < 8; ++y)
      for (int x = 0; x < 8; ++x) a[y][x] = (float)px[(size_t)y * stride + x * step] - 128.0f;
  } else {
    for (int y = 0; y < 8; ++y) {
      const int sy = by + y < ph ? by + y : ph - 1;
      for (int x = 0; x < 8; ++x) {
        const int sx = bx + x < pw ? bx + x : pw - 1;
        a[y][x] = (float)plane[(size_t)sy * stride + (size_t)sx * step] - 128.0f;
      }
    }
  }
  aan_pass(a);                                   // vertical
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) b[x][y] = a[y][x];
  aan_pass(b);                                   // horizontal (on the transposed block)
  const float* cf = &b[0][0];
  int zz[64];
  for (int i = 0; i < 64; ++i) zz[i] = (int)lrintf(cf[src[i]] * recip[i]);
#ifdef JPEG_NO_ENTROPY
  { int sacc = 0; for (int i = 0; i < 64; ++i) sacc += zz[i]; *last_dc += sacc; bw->p[0] = (uint8_t)*last_dc; return; }
#endif
  // DC
  const int diff = zz[0] - *last_dc;
  *last_dc = zz[0];
  {
    const int ad = diff < 0 ? -diff : diff;
    const int nb = ad ? 32 - __builtin_clz((unsigned)ad) : 0;
    bw->put(dc.code[nb], dc.len[nb]);
    if (nb) bw->put((uint32_t)(diff < 0 ? diff + (1 << nb) - 1 : diff), nb);
  }
  // AC: walk the non-zero coefficients through a bit mask (bit i = zigzag position i)
  uint64_t nz = 0;
  for (int i = 1; i < 64; ++i) nz |= (uint64_t)(zz[i] != 0) << i;
  int prev = 0;
  while (nz) {
    const int i = __builtin_ctzll(nz);
    nz &= nz - 1;
    int run = i - prev - 1;
    prev = i;
    while (run > 15) {
      bw->put(ac.code[0xF0], ac.len[0xF0]);
      run -= 16;
    }
    const int v = zz[i];
    const int av = v < 0 ? -v : v;
    const int nb = 32 - __builtin_clz((unsigned)av);
    const int sym = (run << 4) | nb;
    bw->put(((uint32_t)ac.code[sym] << nb) | (uint32_t)(v < 0 ? v + (1 << nb) - 1 : v), ac.len[sym] + nb);
  }
  if (prev != 63) bw->put(ac.code[0x00], ac.len[0x00]);
}

}  // namespace

namespace {
const FastTables& tables_for(int quality) {      // tables of the last quality used by this thread (one quality per node)
  static thread_local FastTables tab;
  if (tab.quality != quality) make_tables(quality, &tab);
  return tab;
}
inline int clamp_quality(int q) { return q < 1 ? 1 : (q > 100 ? 100 : q); }
}  // namespace

int JpegMcuRows(int h) { return (h + 15) / 16; }

bool JpegAppendHeader(int w, int h, int quality, int restart_mcus, std::vector<uint8_t>& out) {
  if (w <= 0 || h <= 0 || (w & 1) || (h & 1) || restart_mcus < 0 || restart_mcus > 65535) return false;
  const FastTables& tab = tables_for(clamp_quality(quality));
  const size_t at = out.size();
  out.resize(at + 1024);        // 623 bytes (+ 6 with a restart interval)
  uint8_t* p = out.data() + at;
  auto put8 = [&](int v) { *p++ = (uint8_t)v; };
  auto put16b = [&](int v) {
    *p++ = (uint8_t)(v >> 8);
    *p++ = (uint8_t)v;
  };
  const uint8_t soi_app0[] = {0xFF, 0xD8, 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
  memcpy(p, soi_app0, sizeof soi_app0);
  p += sizeof soi_app0;
  for (int t = 0; t < 2; ++t) {   // DQT
    put8(0xFF);
    put8(0xDB);
    put16b(67);
    put8(t);
    for (int i = 0; i < 64; ++i) put8((t ? tab.qc : tab.ql)[kZigzag[i]]);
  }
  put8(0xFF);   // SOF0: 8 bit, 3 components, Y 2x2, Cb/Cr 1x1
  put8(0xC0);
  put16b(17);
  put8(8);
  put16b(h);
  put16b(w);
  put8(3);
  const uint8_t comps[9] = {1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1};
  memcpy(p, comps, 9);
  p += 9;
  auto dht = [&](int cls_id, const uint8_t* bits, const uint8_t* vals, int nvals) {
    put8(0xFF);
    put8(0xC4);
    put16b(3 + 16 + nvals);
    put8(cls_id);
    memcpy(p, bits, 16);
    p += 16;
    memcpy(p, vals, nvals);
    p += nvals;
  };
  dht(0x00, kDcLumBits, kDcVals, 12);
  dht(0x10, kAcLumBits, kAcLumVals, 162);
  dht(0x01, kDcChrBits, kDcVals, 12);
  dht(0x11, kAcChrBits, kAcChrVals, 162);
  if (restart_mcus > 0) {          // DRI: the entropy-coded data restarts (DC predictors = 0, byte aligned) every so many MCUs
    put8(0xFF);
    put8(0xDD);
    put16b(4);
    put16b(restart_mcus);
  }
  const uint8_t sos[] = {0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
  memcpy(p, sos, sizeof sos);
  p += sizeof sos;
  out.resize((size_t)(p - out.data()));
  return true;
}

bool JpegAppendMcuRows(const uint8_t* nv12, int w, int h, int pitch, int quality, int row0, int row1, std::vector<uint8_t>& out) {
  if (!nv12 || w <= 0 || h <= 0 || (w & 1) || (h & 1) || pitch < w || row0 < 0 || row1 > JpegMcuRows(h) || row0 >= row1) return false;
  const FastTables& tab = tables_for(clamp_quality(quality));
  // entropy-coded segment, unstuffed, into a per-thread scratch buffer: baseline Huffman coding spends at most 16 + 11
  // bits on a coefficient (27 bits < 4 bytes), so 4 bytes per sample + slack for the 8-byte stores can never overflow
  static thread_local std::vector<uint8_t> scratch;
  const size_t mcus = (size_t)((w + 15) / 16) * (size_t)(row1 - row0);
  const size_t worst = mcus * 384 * 4 + 64;
  if (scratch.size() < worst) scratch.resize(worst);
  FastBits bw;
  bw.p = scratch.data();
  int dc_y = 0, dc_cb = 0, dc_cr = 0;
  const uint8_t* uv = nv12 + (size_t)h * pitch;
  const int cw = w / 2, chh = h / 2;
  for (int my = row0 * 16; my < row1 * 16; my += 16)
    for (int mx = 0; mx < w; mx += 16) {
      for (int b = 0; b < 4; ++b)
        fast_block(nv12, pitch, 1, mx + (b & 1) * 8, my + (b >> 1) * 8, w, h, tab.rl, tab.src, tab.dcl, tab.acl, &dc_y, &bw);
      for (int comp = 0; comp < 2; ++comp)
        fast_block(uv + comp, pitch, 2, mx / 2, my / 2, cw, chh, tab.rc, tab.src, tab.dcc, tab.acc, comp ? &dc_cr : &dc_cb, &bw);
    }
  bw.finish();
  const size_t raw_n = (size_t)(bw.p - scratch.data());
  size_t ffs = 0;
  for (size_t i = 0; i < raw_n; ++i) ffs += scratch[i] == 0xFF;
  const size_t at = out.size();
  out.resize(at + raw_n + ffs);
  stuff_copy(out.data() + at, scratch.data(), raw_n);
  return true;
}

bool EncodeNv12ToJpeg(const uint8_t* nv12, int w, int h, int pitch, int quality, std::vector<uint8_t>& out) {
  out.clear();
  if (!nv12 || pitch < w || !JpegAppendHeader(w, h, quality, 0, out)) return false;
  if (!JpegAppendMcuRows(nv12, w, h, pitch, quality, 0, JpegMcuRows(h), out)) return false;
  out.push_back(0xFF);
  out.push_back(0xD9);
  return true;
}

bool EncodeNv12ToJpegSliced(const uint8_t* nv12, int w, int h, int pitch, int quality, int rows_per_slice,
                            std::vector<uint8_t>& out) {
  out.clear();
  const int rows = JpegMcuRows(h);
  if (rows_per_slice <= 0 || rows_per_slice >= rows) return EncodeNv12ToJpeg(nv12, w, h, pitch, quality, out);
  if (!nv12 || pitch < w || !JpegAppendHeader(w, h, quality, rows_per_slice * ((w + 15) / 16), out)) return false;
  int k = 0;
  for (int r = 0; r < rows; r += rows_per_slice, ++k) {
    const int r1 = r + rows_per_slice < rows ? r + rows_per_slice : rows;
    if (!JpegAppendMcuRows(nv12, w, h, pitch, quality, r, r1, out)) return false;
    out.push_back(0xFF);
    out.push_back(r1 < rows ? (uint8_t)(0xD0 + (k & 7)) : (uint8_t)0xD9);      // RSTm between slices, EOI after the last
  }
  return true;
}

}  // namespace stereonet
}  // namespace hobot
