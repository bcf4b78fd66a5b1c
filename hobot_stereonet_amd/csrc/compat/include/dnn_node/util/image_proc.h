// image_proc.h — dnn_node/util/image_proc.h as hobot_stereonet includes it
// (stereonet_infer/include/stereonet_node.h:24).  The reference only INCLUDES this header; it never calls into it
// (its live path builds the model input with PreProcess::CvtNV12Data2Tensors, stereonet_infer/src/preprocess.cpp:913),
// so the one helper of the closed package whose types the reference names (NV12PyramidInput,
// stereonet_infer/include/preprocess.h:39) is provided in its plain host form and nothing else.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>

#include "dnn_node/dnn_node_data.h"

namespace hobot {
namespace dnn_node {

class ImageProc {
 public:
  // Wraps a contiguous NV12 image (h * 3 / 2 rows of w bytes) into an NV12PyramidInput that owns a host copy.
  // No scaling on this platform: the scaled size must equal the input size, otherwise nullptr (the reference's -1
  // convention for "not supported", stereonet_infer/include/parser.h:37-39).
  static std::shared_ptr<NV12PyramidInput> GetNV12PyramidFromNV12Img(const char* in_img_data, const int& in_img_height,
                                                                     const int& in_img_width, const int& scaled_img_height,
                                                                     const int& scaled_img_width) {
    if (!in_img_data || in_img_height <= 0 || in_img_width <= 0 || (in_img_height & 1) || (in_img_width & 1) ||
        scaled_img_height != in_img_height || scaled_img_width != in_img_width)
      return nullptr;
    const size_t y_size = static_cast<size_t>(in_img_height) * in_img_width, uv_size = y_size / 2;
    auto* y_mem = new hbSysMem;
    auto* uv_mem = new hbSysMem;
    if (hbSysAllocCachedMem(y_mem, static_cast<uint32_t>(y_size)) != 0 ||
        hbSysAllocCachedMem(uv_mem, static_cast<uint32_t>(uv_size)) != 0) {
      delete y_mem;
      delete uv_mem;
      return nullptr;
    }
    std::memcpy(y_mem->virAddr, in_img_data, y_size);
    std::memcpy(uv_mem->virAddr, in_img_data + y_size, uv_size);
    auto* pyramid = new NV12PyramidInput;
    pyramid->width = in_img_width;
    pyramid->height = in_img_height;
    pyramid->y_stride = in_img_width;
    pyramid->uv_stride = in_img_width;
    pyramid->y_phy_addr = y_mem->phyAddr;
    pyramid->y_vir_addr = y_mem->virAddr;
    pyramid->uv_phy_addr = uv_mem->phyAddr;
    pyramid->uv_vir_addr = uv_mem->virAddr;
    return std::shared_ptr<NV12PyramidInput>(pyramid, [y_mem, uv_mem](NV12PyramidInput* p) {
      hbSysFreeMem(y_mem);
      hbSysFreeMem(uv_mem);
      delete y_mem;
      delete uv_mem;
      delete p;
    });
  }
};

}  // namespace dnn_node
}  // namespace hobot
