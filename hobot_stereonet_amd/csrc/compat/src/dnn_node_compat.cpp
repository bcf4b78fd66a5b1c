// dnn_node_compat.cpp — hbSys*/hbDNN* shims, Model, and DnnNode on top of libstereonet_hip.so.
// Replaces the closed dnn_node + libdnn + BPU stack for the one model hobot_stereonet runs
// (reference call sites: stereonet_infer/src/stereonet_node.cpp:44-103,129-147,812,968,980-1089).
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "dnn_node/dnn_node.h"
#include "stereonet_hip.h"

extern "C" {

int32_t hbSysAllocCachedMem(hbSysMem* mem, uint32_t size) {
  if (!mem) return -1;
  void* p = nullptr;
  if (posix_memalign(&p, 256, size ? size : 1) != 0) return -1;
  mem->phyAddr = 0;
  mem->virAddr = p;
  mem->memSize = size;
  return 0;
}
int32_t hbSysAllocMem(hbSysMem* mem, uint32_t size) { return hbSysAllocCachedMem(mem, size); }
int32_t hbSysFreeMem(hbSysMem* mem) {
  if (!mem) return -1;
  free(mem->virAddr);
  mem->virAddr = nullptr;
  mem->memSize = 0;
  return 0;
}
// CPU caches are coherent with the staging copies the engine makes; nothing to clean or invalidate.
int32_t hbSysFlushMem(hbSysMem* mem, int32_t) { return mem ? 0 : -1; }

int32_t hbDNNGetInputTensorProperties(hbDNNTensorProperties* p, hbDNNHandle_t h, int32_t idx) {
  if (!p || !h) return -1;
  return static_cast<hobot::dnn_node::Model*>(h)->GetInputTensorProperties(*p, idx);
}
int32_t hbDNNGetOutputTensorProperties(hbDNNTensorProperties* p, hbDNNHandle_t h, int32_t idx) {
  if (!p || !h) return -1;
  return static_cast<hobot::dnn_node::Model*>(h)->GetOutputTensorProperties(*p, idx);
}

}  // extern "C"

namespace hobot {
namespace dnn_node {

namespace {
double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

Model::Model(sn_handle* h) : h_(h) {
  sn_io_info info;
  if (sn_get_io_info(h, &info) == SN_OK) {
    w_ = info.width;
    hgt_ = info.height;
    dmax_ = info.dmax;
    out_scale_[0] = info.out_scale;
  }
  for (auto& s : in_scale_) s = 1.0f / 128.0f;   // preprocess.cpp:1131 "scale: 0.0078125"
}
Model::~Model() = default;

int Model::GetInputTensorProperties(hbDNNTensorProperties& p, int32_t index) const {
  if (index != 0) return -1;
  memset(&p, 0, sizeof p);
  const int32_t dims[4] = {1, 6, hgt_, w_};   // int8 NCHW (stereonet_node.cpp:63-72, preprocess.cpp:937-945)
  for (int i = 0; i < 4; ++i) p.validShape.dimensionSize[i] = p.alignedShape.dimensionSize[i] = dims[i];
  p.validShape.numDimensions = p.alignedShape.numDimensions = 4;
  p.tensorLayout = HB_DNN_LAYOUT_NCHW;
  p.tensorType = HB_DNN_TENSOR_TYPE_S8;
  p.scale.scaleLen = 6;
  p.scale.scaleData = in_scale_;
  p.alignedByteSize = 6 * hgt_ * w_;
  return 0;
}

int Model::GetOutputTensorProperties(hbDNNTensorProperties& p, int32_t index) const {
  if (index != 0) return -1;
  memset(&p, 0, sizeof p);
  const int32_t dims[4] = {1, 1, hgt_, w_};   // int32 NCHW, scale 2.604e-6 (stereonet_node.cpp:282, parser.cpp:60)
  for (int i = 0; i < 4; ++i) p.validShape.dimensionSize[i] = p.alignedShape.dimensionSize[i] = dims[i];
  p.validShape.numDimensions = p.alignedShape.numDimensions = 4;
  p.tensorLayout = HB_DNN_LAYOUT_NCHW;
  p.tensorType = HB_DNN_TENSOR_TYPE_S32;
  p.scale.scaleLen = 1;
  p.scale.scaleData = out_scale_;
  p.alignedByteSize = 4 * hgt_ * w_;
  return 0;
}

DnnNode::DnnNode(const std::string& node_name, const rclcpp::NodeOptions& options)
    : rclcpp::Node(node_name, options), dnn_node_para_ptr_(std::make_shared<DnnNodePara>()) {}

DnnNode::~DnnNode() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
  }
  cv_.notify_all();
  if (worker_.joinable()) worker_.join();
  model_.reset();
  if (engine_) sn_destroy(engine_);
}

int DnnNode::Init() {
  if (engine_) return 0;
  if (SetNodePara() != 0 || !dnn_node_para_ptr_) return -1;
  sn_config cfg{};
  cfg.device = getenv("STEREONET_DEVICE") ? atoi(getenv("STEREONET_DEVICE")) : -1;
  cfg.max_batch = 1;
  cfg.task_num = dnn_node_para_ptr_->task_num;
  const char* prec = getenv("STEREONET_PRECISION");   // knob kept out of the ROS parameter surface
  // unset / "auto": SN_PREC_AUTO (the fp16 tower while the model stays inside its envelope, the split mode otherwise)
  cfg.precision = (prec && !strcmp(prec, "fp32")) ? SN_PREC_FP32 : (prec && !strcmp(prec, "f16x3")) ? SN_PREC_F16X3
                  : (prec && !strcmp(prec, "f16")) ? SN_PREC_F16 : SN_PREC_AUTO;
  const int rc = sn_create(dnn_node_para_ptr_->model_file.c_str(), &cfg, &engine_);
  if (rc != SN_OK) {
    RCLCPP_ERROR(rclcpp::get_logger("dnn"), "load model %s failed: %s", dnn_node_para_ptr_->model_file.c_str(),
                 sn_strerror(rc));
    engine_ = nullptr;
    return -1;
  }
  model_.reset(new Model(engine_));
  stat_t0_ = now_s();
  worker_ = std::thread(&DnnNode::CompletionLoop, this);
  return 0;
}

int DnnNode::GetModelInputSize(int32_t input_index, int& w, int& h) {
  if (!model_ || input_index != 0) return -1;
  w = model_->width();
  h = model_->height();
  return 0;
}

Model* DnnNode::GetModel() { return model_.get(); }

std::shared_ptr<DNNTensor> DnnNode::MakeOutputTensor() {
  // the runtime owns the output tensor; it lives as long as the DnnNodeOutput that references it
  std::shared_ptr<DNNTensor> t(new DNNTensor(), [](DNNTensor* p) {
    if (p) {
      hbSysFreeMem(&p->sysMem[0]);
      delete p;
    }
  });
  model_->GetOutputTensorProperties(t->properties, 0);
  if (hbSysAllocCachedMem(&t->sysMem[0], (uint32_t)(4 * model_->width() * model_->height())) != 0) return nullptr;
  return t;
}

void DnnNode::UpdateStat(const std::shared_ptr<DnnNodeOutput>& out, float infer_ms) {
  if (!out->rt_stat) out->rt_stat = std::make_shared<DnnNodeRunTimeStat>();
  out->rt_stat->infer_time_ms = (int)(infer_ms + 0.5f);
  ++out_count_;
  const double t = now_s();
  if (t - stat_t0_ >= 1.0) {   // refresh the fps pair once a second, flag the request that carries it
    last_in_fps_ = (float)(in_count_ / (t - stat_t0_));
    last_out_fps_ = (float)(out_count_ / (t - stat_t0_));
    in_count_ = out_count_ = 0;
    stat_t0_ = t;
    out->rt_stat->fps_updated = true;
    sn_refine_stats rs;
    if (sn_get_refine_stats(engine_, &rs) == SN_OK) {
      out->rt_stat->refine_residual_px = (float)rs.residual_px;
      out->rt_stat->arithmetic = rs.precision_last == SN_PREC_F16X3 ? "f16x3" : rs.precision_last == SN_PREC_FP32 ? "fp32" : "f16";
    }
  }
  out->rt_stat->input_fps = last_in_fps_;
  out->rt_stat->output_fps = last_out_fps_;
}

int DnnNode::Run(std::vector<std::shared_ptr<DNNTensor>>& inputs, const std::shared_ptr<DnnNodeOutput>& output,
                 bool is_sync_mode, int alloc_chn_timeout_ms, int /*infer_timeout_ms*/) {
  if (!engine_ || !model_ || inputs.size() != 1 || !inputs[0] || !inputs[0]->sysMem[0].virAddr) return -1;
  const size_t need = (size_t)6 * model_->width() * model_->height();
  if (inputs[0]->sysMem[0].memSize < need) return -1;
  std::shared_ptr<DnnNodeOutput> out = output ? output : std::make_shared<DnnNodeOutput>();
  auto ot = MakeOutputTensor();
  if (!ot) return -1;
  out->output_tensors.clear();
  out->output_tensors.push_back(ot);
  const int8_t* in = static_cast<const int8_t*>(inputs[0]->sysMem[0].virAddr);
  int32_t* raw = static_cast<int32_t*>(ot->sysMem[0].virAddr);
  if (is_sync_mode) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      ++in_count_;
    }
    const double t0 = now_s();
    if (sn_infer_i8(engine_, in, raw, nullptr, SN_MEM_HOST, nullptr) != SN_OK) {
      RCLCPP_ERROR(rclcpp::get_logger("dnn"), "infer failed: %s", sn_last_error(engine_));
      return -1;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      UpdateStat(out, (float)((now_s() - t0) * 1e3));
    }
    return PostProcess(out) < 0 ? -1 : 0;
  }
  uint64_t ticket = 0;
  const int rc = sn_submit(engine_, in, raw, nullptr, alloc_chn_timeout_ms, &ticket);
  if (rc != SN_OK) {
    RCLCPP_ERROR(rclcpp::get_logger("dnn"), "submit failed: %s", sn_strerror(rc));
    return -1;
  }
  {
    std::lock_guard<std::mutex> lk(mu_);
    ++in_count_;                       // only requests the engine accepted count as input frames
    pending_.push_back(Pending{ticket, out, ot});
    ++busy_;
  }
  cv_.notify_all();
  return 0;
}

int DnnNode::RunSbsNv12(const uint8_t* sbs, int width2, int height, const std::shared_ptr<DnnNodeOutput>& output,
                        bool is_sync_mode, int alloc_chn_timeout_ms) {
  if (!engine_ || !model_ || !sbs || width2 != 2 * model_->width() || height != model_->height()) return -1;
  std::shared_ptr<DnnNodeOutput> out = output ? output : std::make_shared<DnnNodeOutput>();
  auto ot = MakeOutputTensor();
  if (!ot) return -1;
  out->output_tensors.clear();
  out->output_tensors.push_back(ot);
  int32_t* raw = static_cast<int32_t*>(ot->sysMem[0].virAddr);
  if (is_sync_mode) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      ++in_count_;
    }
    const double t0 = now_s();
    if (sn_infer_sbs_nv12(engine_, sbs, width2, height, raw, nullptr, nullptr, SN_MEM_HOST, nullptr) != SN_OK) {
      RCLCPP_ERROR(rclcpp::get_logger("dnn"), "infer failed: %s", sn_last_error(engine_));
      return -1;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      UpdateStat(out, (float)((now_s() - t0) * 1e3));
    }
    return PostProcess(out) < 0 ? -1 : 0;
  }
  uint64_t ticket = 0;
  const int rc = sn_submit_nv12(engine_, sbs, width2, height, raw, nullptr, alloc_chn_timeout_ms, &ticket);
  if (rc != SN_OK) {
    RCLCPP_ERROR(rclcpp::get_logger("dnn"), "submit failed: %s", sn_strerror(rc));
    return -1;
  }
  {
    std::lock_guard<std::mutex> lk(mu_);
    ++in_count_;                       // only requests the engine accepted count as input frames
    pending_.push_back(Pending{ticket, out, ot});
    ++busy_;
  }
  cv_.notify_all();
  return 0;
}

void DnnNode::CompletionLoop() {
  for (;;) {
    Pending p;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return stop_ || !pending_.empty(); });
      if (pending_.empty()) return;   // stop requested and queue drained
      p = pending_.front();
      pending_.pop_front();
    }
    float ms = 0.f;
    if (sn_wait(engine_, p.ticket, &ms) != SN_OK) {
      RCLCPP_ERROR(rclcpp::get_logger("dnn"), "wait failed: %s", sn_last_error(engine_));
    } else {
      {
        std::lock_guard<std::mutex> lk(mu_);
        UpdateStat(p.output, ms);
      }
      PostProcess(p.output);
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      --busy_;
    }
    idle_cv_.notify_all();
  }
}

void DnnNode::WaitIdle() {
  std::unique_lock<std::mutex> lk(mu_);
  idle_cv_.wait(lk, [&] { return busy_ == 0; });
}

int DnnNode::PostProcess(const std::shared_ptr<DnnNodeOutput>&) { return 0; }

}  // namespace dnn_node
}  // namespace hobot
