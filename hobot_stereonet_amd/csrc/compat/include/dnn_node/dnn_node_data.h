// dnn_node_data.h — data types of the (closed) ROS package `dnn_node` as used by hobot_stereonet:
// DNNTensor (preprocess.cpp:81-92), DnnNodeOutput (stereonet_node.cpp:693-696,1033-1034,1071-1083),
// DnnNodePara / ModelTaskType (stereonet_node.cpp:129-147), Model (stereonet_node.cpp:51-91), and the two input types
// the reference only names in using-declarations (stereonet_infer/include/preprocess.h:36,39): DNNInput, NV12PyramidInput.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "dnn_node/hb_dnn_compat.h"
#include "std_msgs/msg/header.hpp"

struct sn_handle;

namespace hobot {
namespace dnn_node {

struct DNNTensor {
  hbSysMem sysMem[4] = {};
  hbDNNTensorProperties properties = {};
};

// Base of the typed model inputs of dnn_node; hobot_stereonet feeds DNNTensor (below: preprocess.cpp:952-966) and only
// imports these two names into its namespace.
struct DNNInput {
  virtual ~DNNInput() = default;
  virtual void Reset() {}
};

struct NV12PyramidInput : DNNInput {
  uint64_t y_phy_addr = 0;
  void* y_vir_addr = nullptr;
  uint64_t uv_phy_addr = 0;
  void* uv_vir_addr = nullptr;
  int32_t height = 0, width = 0, y_stride = 0, uv_stride = 0;
};

enum class ModelTaskType { InvalidType = 0, ModelInferType = 1, ModelRoiInferType = 2 };

struct DnnNodePara {
  std::string model_file;
  std::string model_name;
  ModelTaskType model_task_type = ModelTaskType::ModelInferType;
  int task_num = 2;
  std::vector<int> bpu_core_ids;   // accepted and ignored (no BPU)
};

struct DnnNodeRunTimeStat {
  float input_fps = 0.f;
  float output_fps = 0.f;
  int infer_time_ms = 0;
  int parse_time_ms = 0;
  bool fps_updated = false;
  // Extension of this backend (filled together with fps_updated): what the refinement moved the last request's map by
  // (sn_get_refine_stats: residual_px) and the arithmetic its map was computed in ("f16", "f16x3", "fp32"; under the
  // default SN_PREC_AUTO the engine leaves the fp16 tower when that statistic leaves its envelope)
  float refine_residual_px = 0.f;
  const char* arithmetic = "f16";
};

struct DnnNodeOutput {
  virtual ~DnnNodeOutput() = default;
  std::shared_ptr<std_msgs::msg::Header> msg_header = nullptr;
  std::vector<std::shared_ptr<DNNTensor>> output_tensors;
  std::shared_ptr<DnnNodeRunTimeStat> rt_stat = nullptr;
};

// What `Model*` exposes to the node (stereonet_node.cpp:57-75,91).  Backed by an sn_handle.
class Model {
 public:
  explicit Model(sn_handle* h);
  ~Model();
  int32_t GetInputCount() const { return 1; }
  int32_t GetOutputCount() const { return 1; }
  int GetInputTensorProperties(hbDNNTensorProperties& properties, int32_t index) const;
  int GetOutputTensorProperties(hbDNNTensorProperties& properties, int32_t index) const;
  hbDNNHandle_t GetDNNHandle() const { return const_cast<Model*>(this); }
  sn_handle* engine() const { return h_; }
  int width() const { return w_; }
  int height() const { return hgt_; }
  int dmax() const { return dmax_; }

 private:
  sn_handle* h_;
  int w_ = 0, hgt_ = 0, dmax_ = 0;
  mutable float in_scale_[6];
  mutable float out_scale_[1];
};

}  // namespace dnn_node
}  // namespace hobot
