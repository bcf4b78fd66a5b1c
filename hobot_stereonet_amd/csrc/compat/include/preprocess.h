// preprocess.h — host mirror of the reference's live pre-processing API
// (stereonet_infer/include/preprocess.h:128-155 Tools::YUV420TOYUV444, :187-240 PreProcess).
// Same names, argument meaning and error behaviour; the dead variants of the reference
// (CvtImgData2Tensors, CvtBinData2Tensors, CvtNV12File2Tensors: unreachable) are
// out of scope.  The device-side twin is sn_preprocess_nv12 / sn_infer_sbs_nv12 (include/stereonet_hip.h).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "dnn_node/dnn_node_data.h"

namespace hobot {
namespace stereonet {

using hobot::dnn_node::DNNTensor;
using hobot::dnn_node::Model;
typedef float float32_t;

class Tools {
 public:
  // Y plane copied; the chroma 2x2-replicated while indexing `inbuf + w*h` as planar I420
  // ("U" = first w*h/4 bytes, "V" = next w*h/4 bytes) exactly as the reference does.
  static void YUV420TOYUV444(const unsigned char* inbuf, unsigned char* outbuf, int w, int h);

  // Offline feeders only (reference: preprocess.h:56-96, cv::cvtColor(COLOR_BGR2YUV_I420) + UV interleave).
  // 8-bit BGR (interleaved, top row first) -> NV12: BT.601 studio-range luma for every pixel, chroma taken
  // from the top-left pixel of each 2x2 block (no averaging), 20-bit fixed point with round-half-up — the
  // arithmetic OpenCV 4.x documents for its RGB->YUV420p path.  -1 when w or h is odd (as the reference).
  static int32_t BGRToNv12(const unsigned char* bgr, int w, int h, std::vector<unsigned char>& nv12);
};

class PreProcess {
 public:
  explicit PreProcess(const std::string& config_file);
  // -> one int8 NCHW 1x6xHxW tensor in hbSys memory (L-Y, L-U, L-V, R-Y, R-U, R-V, each byte ^ 0x80);
  // returns 0, or -1 on invalid arguments / allocation failure.
  int CvtNV12Data2Tensors(std::vector<std::shared_ptr<DNNTensor>>& input_tensors, Model* pmodel,
                          const unsigned char* img_l, const unsigned char* img_r);
  // scale 0.0078125, zero_point 0.5, clamp [-128,127]  (reference defaults)
  static int8_t Quantize(float32_t value, float32_t const scale = 0.0078125, float32_t const zero_point = 0.5,
                         float32_t const min = -128, float32_t const max = 127);
};

}  // namespace stereonet
}  // namespace hobot
