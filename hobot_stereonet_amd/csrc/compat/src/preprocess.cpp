// preprocess.cpp — see include/preprocess.h.  Behaviour follows stereonet_infer/src/preprocess.cpp:913-1059
// and :1131-1136; the implementation is a single pass per plane instead of the reference's five buffer copies.
#include "preprocess.h"

#include <cmath>
#include <cstring>

#include "rclcpp/rclcpp.hpp"

namespace hobot {
namespace stereonet {

void Tools::YUV420TOYUV444(const unsigned char* inbuf, unsigned char* outbuf, int w, int h) {
  const size_t wh = (size_t)w * h;
  memcpy(outbuf, inbuf, wh);
  const unsigned char* cu = inbuf + wh;
  const unsigned char* cv = cu + wh / 4;
  unsigned char* du = outbuf + wh;
  unsigned char* dv = du + wh;
  for (int i = 0; i < h; ++i) {
    const int row = (i / 2) * w / 2;
    for (int j = 0; j < w; ++j) {
      du[(size_t)i * w + j] = cu[row + j / 2];
      dv[(size_t)i * w + j] = cv[row + j / 2];
    }
  }
}

int32_t Tools::BGRToNv12(const unsigned char* bgr, int w, int h, std::vector<unsigned char>& nv12) {
  if (!bgr || w <= 0 || h <= 0 || (w & 1) || (h & 1)) {
    RCLCPP_ERROR_STREAM(rclcpp::get_logger("hobot_stereonet"), "input img height and width must aligned by 2!");
    return -1;
  }
  // BT.601 studio range in Q20 (round(2^20 * {0.257, 0.504, 0.098 | -0.148, -0.291, 0.439 | 0.439, -0.368, -0.071}))
  constexpr int kQ = 20, kHalf = 1 << (kQ - 1);
  constexpr int kYR = 269484, kYG = 528482, kYB = 102760;
  constexpr int kUR = -155188, kUG = -305135, kUB = 460324;
  constexpr int kVR = 460324, kVG = -385875, kVB = -74448;
  auto clamp8 = [](int v) { return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); };
  nv12.resize((size_t)w * h * 3 / 2);
  unsigned char* yp = nv12.data();
  unsigned char* cp = yp + (size_t)w * h;
  for (int r = 0; r < h; ++r) {
    const unsigned char* s = bgr + (size_t)r * w * 3;
    for (int x = 0; x < w; ++x, s += 3) {
      const int b = s[0], g = s[1], rr = s[2];
      yp[(size_t)r * w + x] = clamp8((kYR * rr + kYG * g + kYB * b + kHalf + (16 << kQ)) >> kQ);
      if (!(r & 1) && !(x & 1)) {
        unsigned char* c = cp + (size_t)(r / 2) * w + x;   // interleaved U,V
        c[0] = clamp8((kUR * rr + kUG * g + kUB * b + kHalf + (128 << kQ)) >> kQ);
        c[1] = clamp8((kVR * rr + kVG * g + kVB * b + kHalf + (128 << kQ)) >> kQ);
      }
    }
  }
  return 0;
}

PreProcess::PreProcess(const std::string&) {}   // the reference ignores its config_file too (preprocess.cpp:35-36)

int8_t PreProcess::Quantize(float32_t value, float32_t const scale, float32_t const zero_point, float32_t const min,
                            float32_t const max) {
  value = std::floor(value / scale + zero_point);
  value = std::fmin(std::fmax(value, min), max);
  return static_cast<int8_t>(value);
}

int PreProcess::CvtNV12Data2Tensors(std::vector<std::shared_ptr<DNNTensor>>& input_tensors, Model* pmodel,
                                    const unsigned char* img_l, const unsigned char* img_r) {
  if (!pmodel || !img_l || !img_r) {
    RCLCPP_ERROR_STREAM(rclcpp::get_logger("hobot_stereonet"), "Invalid input data");
    return -1;
  }
  hbDNNTensorProperties properties;
  if (pmodel->GetInputTensorProperties(properties, 0) != 0) return -1;
  int h_index = 1, w_index = 2, c_index = 3;
  if (properties.tensorLayout == HB_DNN_LAYOUT_NCHW) {
    c_index = 1;
    h_index = 2;
    w_index = 3;
  }
  const int in_h = properties.validShape.dimensionSize[h_index];
  const int in_w = properties.validShape.dimensionSize[w_index];
  const int chn = properties.validShape.dimensionSize[c_index];
  if (chn != 6 || in_h <= 0 || in_w <= 0) return -1;
  const size_t plane = (size_t)in_h * in_w;

  std::shared_ptr<DNNTensor> t(new DNNTensor(), [](DNNTensor* p) {
    if (p) {
      if (p->sysMem[0].memSize > 0) hbSysFreeMem(&p->sysMem[0]);
      delete p;
    }
  });
  t->properties = properties;
  if (hbSysAllocCachedMem(&t->sysMem[0], (uint32_t)(plane * 6)) != 0) return -1;
  unsigned char* dst = static_cast<unsigned char*>(t->sysMem[0].virAddr);
  Tools::YUV420TOYUV444(img_l, dst, in_w, in_h);
  Tools::YUV420TOYUV444(img_r, dst + 3 * plane, in_w, in_h);
  // Quantize(((float)b - 128) / 128) == b - 128 == b ^ 0x80 for all 256 byte values
  // (tests/golden/preprocess_golden.npz holds the reference's own table)
  uint64_t* q = reinterpret_cast<uint64_t*>(dst);
  const size_t n8 = plane * 6 / 8;
  for (size_t i = 0; i < n8; ++i) q[i] ^= 0x8080808080808080ull;
  for (size_t i = n8 * 8; i < plane * 6; ++i) dst[i] ^= 0x80;
  hbSysFlushMem(&t->sysMem[0], HB_SYS_MEM_CACHE_CLEAN);
  input_tensors.emplace_back(t);
  return 0;
}

}  // namespace stereonet
}  // namespace hobot
