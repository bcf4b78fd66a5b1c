// parser.h — host mirror of stereonet_infer/include/parser.h (Parse / StereonetResult): dequantises the
// int32 output tensor and converts it to metric depth.  The reference's OpenCV colour-map rendering after
// the conversion is dead code (parser.cpp:111-120 returns before using it) and is not reproduced.
#pragma once
#include <memory>
#include <vector>

#include "dnn_node/dnn_node_data.h"

namespace hobot {
namespace stereonet {

struct StereonetResult {
  std::vector<float> results;   // depth in metres, H*W (parser.cpp:86)
  std::vector<float> disparity; // extension: disparity in px (dis * 16 * 12)
};

// 0 success, -1 failure (parser.h:37-39)
int32_t Parse(const std::shared_ptr<hobot::dnn_node::DnnNodeOutput>& node_output,
              std::vector<std::shared_ptr<StereonetResult>>& results);
int get_tensor_hw(std::shared_ptr<hobot::dnn_node::DNNTensor> tensor, int* height, int* width, int* chn);

}  // namespace stereonet
}  // namespace hobot
