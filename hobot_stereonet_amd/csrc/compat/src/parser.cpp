// parser.cpp — see include/parser.h (behaviour: stereonet_infer/src/parser.cpp:33-94,169-199).
#include "parser.h"

namespace hobot {
namespace stereonet {

using hobot::dnn_node::DNNTensor;

int get_tensor_hw(std::shared_ptr<DNNTensor> tensor, int* height, int* width, int* chn) {
  if (!tensor || !height || !width || !chn) return -1;
  int h_index, w_index, c_index;
  if (tensor->properties.tensorLayout == HB_DNN_LAYOUT_NHWC) {
    h_index = 1; w_index = 2; c_index = 3;
  } else if (tensor->properties.tensorLayout == HB_DNN_LAYOUT_NCHW) {
    c_index = 1; h_index = 2; w_index = 3;
  } else {
    return -1;
  }
  *height = tensor->properties.validShape.dimensionSize[h_index];
  *width = tensor->properties.validShape.dimensionSize[w_index];
  *chn = tensor->properties.validShape.dimensionSize[c_index];
  return 0;
}

int32_t Parse(const std::shared_ptr<hobot::dnn_node::DnnNodeOutput>& node_output,
              std::vector<std::shared_ptr<StereonetResult>>& results) {
  if (!node_output || node_output->output_tensors.empty() || !node_output->output_tensors[0]) return -1;
  auto tensor = node_output->output_tensors[0];
  hbSysFlushMem(&tensor->sysMem[0], HB_SYS_MEM_CACHE_INVALIDATE);
  int height, width, chn;
  if (get_tensor_hw(tensor, &height, &width, &chn) != 0) return -1;
  const int32_t* data = static_cast<const int32_t*>(tensor->sysMem[0].virAddr);
  const float* scale = tensor->properties.scale.scaleData;
  if (!data || !scale) return -1;
  const float f = 527.1931762695312f;   // focal length, px      (parser.cpp:70)
  const float B = 119.89382172f;        // baseline, mm          (parser.cpp:71)
  auto res = std::make_shared<StereonetResult>();
  res->results.reserve((size_t)chn * height * width);
  res->disparity.reserve((size_t)chn * height * width);
  for (int c = 0; c < chn; ++c)
    for (int i = 0; i < height * width; ++i) {
      const float dis = static_cast<float>(data[(size_t)c * height * width + i]) * scale[c];
      // Z = f * B / (dis * 16 * 12) / 1000; zero disparity gives IEEE inf as in the reference
      res->results.push_back(f * B / (dis * 16.0 * 12.0) / 1000.0);
      res->disparity.push_back(dis * 16.0f * 12.0f);
    }
  results.push_back(res);
  return 0;
}

}  // namespace stereonet
}  // namespace hobot
