// dnn_node.h — hobot::dnn_node::DnnNode as hobot_stereonet derives from it
// (stereonet_infer/include/stereonet_node.h:61), re-implemented over libstereonet_hip.so.
// Members = exactly what the reference touches:
//   DnnNode(name, options)               stereonet_node.cpp:26
//   int Init()  -> virtual SetNodePara() stereonet_node.cpp:44,129
//   dnn_node_para_ptr_                   stereonet_node.cpp:130-144
//   GetModelInputSize / GetModel         stereonet_node.cpp:45,51
//   int Run(inputs, output, is_sync, alloc_timeout_ms, infer_timeout_ms)   stereonet_node.cpp:812,968
//   virtual int PostProcess(output)      stereonet_node.h:70-71 (called on a worker thread)
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dnn_node/dnn_node_data.h"
#include "rclcpp/rclcpp.hpp"

namespace hobot {
namespace dnn_node {

class DnnNode : public rclcpp::Node {
 public:
  explicit DnnNode(const std::string& node_name, const rclcpp::NodeOptions& options = rclcpp::NodeOptions());
  ~DnnNode() override;

  int Init();
  int GetModelInputSize(int32_t input_index, int& w, int& h);
  Model* GetModel();
  // returns 0, or <0 on failure.  is_sync_mode=false: returns after the request is queued; PostProcess
  // runs on the completion thread, requests complete in submission order, at most task_num in flight.
  int Run(std::vector<std::shared_ptr<DNNTensor>>& inputs, const std::shared_ptr<DnnNodeOutput>& output = nullptr,
          bool is_sync_mode = false, int alloc_chn_timeout_ms = -1, int infer_timeout_ms = 1000);
  // Extension of this backend (not part of the reference's dnn_node): Run() on FeedImg's raw 2W x H side-by-side NV12
  // message payload.  The split (stereonet_node.cpp:705-738) and CvtNV12Data2Tensors (preprocess.cpp:913-1059) run on
  // the GPU, bit-identical to the host steps, and the host ships 2.76 MB per frame instead of the 5.53 MB tensor.
  int RunSbsNv12(const uint8_t* sbs_nv12, int width2, int height, const std::shared_ptr<DnnNodeOutput>& output = nullptr,
                 bool is_sync_mode = false, int alloc_chn_timeout_ms = -1);
  // Extension (measurement / shutdown helper): blocks until every asynchronous request accepted so far has been through
  // PostProcess.
  void WaitIdle();

 protected:
  virtual int SetNodePara() = 0;
  virtual int PostProcess(const std::shared_ptr<DnnNodeOutput>& output);
  std::shared_ptr<DnnNodePara> dnn_node_para_ptr_ = nullptr;

 private:
  struct Pending {
    uint64_t ticket;
    std::shared_ptr<DnnNodeOutput> output;
    std::shared_ptr<DNNTensor> out_tensor;
  };
  void CompletionLoop();
  std::shared_ptr<DNNTensor> MakeOutputTensor();
  void UpdateStat(const std::shared_ptr<DnnNodeOutput>& out, float infer_ms);

  sn_handle* engine_ = nullptr;
  std::unique_ptr<Model> model_;
  std::thread worker_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Pending> pending_;
  int busy_ = 0;                      // asynchronous requests accepted and not yet through PostProcess
  std::condition_variable idle_cv_;
  std::atomic<bool> stop_{false};
  // fps statistics (the "input fps / out fps" log line, stereonet_node.cpp:1071-1086)
  int in_count_ = 0, out_count_ = 0;
  double stat_t0_ = 0.0;
  float last_in_fps_ = 0.f, last_out_fps_ = 0.f;
};

}  // namespace dnn_node
}  // namespace hobot
