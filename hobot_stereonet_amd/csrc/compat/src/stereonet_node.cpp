// stereonet_node.cpp — live path of the stereo node, implemented over the dnn_node compat layer.
// Behavioural reference (what must stay observable from outside): stereonet_infer/src/stereonet_node.cpp
//   :24-127  parameters, Init, subscription + publishers      -> StereonetNode(), DeclareAndReadParameters()
//   :129-147 SetNodePara (model file must exist, task_num 4)  -> SetNodePara()
//   :657-818 FeedImg (validate, split eyes, tensor, JPEG, async Run) -> OnStereoFrame()
//   :980-1089 PostProcess (payload = raw tensor || JPEG, fps log)    -> PostProcess()
//   :820-976 RunImglistFeedInfer (offline file-list feeder)          -> RunImglistFeedInfer()
// The reference's other disabled feeders and dump helpers (:149-655) are not reproduced.
#include "stereonet_node.h"

#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <mutex>
#include <thread>

#include "image_io.h"
#include "jpeg_nv12.h"
#include "jpeg_pool.h"

namespace hobot {
namespace stereonet {

namespace {
const rclcpp::Logger kLog = rclcpp::get_logger("stereonet_node");

// STEREONET_NODE_STATS=1: average microseconds per frame of the node's own stages, printed when the process ends
struct StageStats {
  const bool on = getenv("STEREONET_NODE_STATS") != nullptr && atoi(getenv("STEREONET_NODE_STATS")) == 1;
  std::atomic<long> us[6] = {};
  std::atomic<long> n[6] = {};
  void add(int k, std::chrono::steady_clock::time_point t0) {
    if (!on) return;
    us[k] += (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    ++n[k];
  }
  ~StageStats() {
    if (!on) return;
    static const char* name[6] = {"FeedImg total", "FeedImg: queue JPEG", "FeedImg: Run (submit)", "PostProcess: wait for the JPEG",
                                  "PostProcess: build the message", "PostProcess: publish"};
    for (int k = 0; k < 6; ++k)
      if (n[k]) fprintf(stderr, "[node stats] %-34s %8.1f us/frame over %ld frames\n", name[k], (double)us[k] / n[k], (long)n[k]);
  }
};
StageStats g_stats;

int elapsed_ms(std::chrono::steady_clock::time_point since) {
  return (int)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - since).count();
}
}  // namespace

StereonetNode::StereonetNode(const std::string& node_name, const rclcpp::NodeOptions& options)
    : hobot::dnn_node::DnnNode(node_name, options) {
  DeclareAndReadParameters();

  // Init() calls SetNodePara() and loads the model; any failure ends the process the way the reference does
  if (Init() != 0 || GetModelInputSize(0, net_w_, net_h_) < 0) {
    RCLCPP_ERROR(kLog, "Node init fail!");
    rclcpp::shutdown();
    return;
  }
  net_ = GetModel();
  if (net_ == nullptr) {
    RCLCPP_ERROR(kLog, "Invalid model");
    rclcpp::shutdown();
    return;
  }
  LogModelIo();

  pre_.reset(new PreProcess(""));
  if (cfg_.publish_output) {
    int n = cfg_.jpeg_threads;
    if (const char* e = getenv("STEREONET_JPEG_THREADS")) n = atoi(e);
    if (n <= 0) {
      n = (int)std::thread::hardware_concurrency() / 4;
      n = n < 2 ? 2 : (n > 32 ? 32 : n);
    }
    jpeg_pool_ = std::make_shared<JpegPool>(n);
    if (const char* e = getenv("STEREONET_JPEG_SLICES")) cfg_.jpeg_slices = atoi(e);
    RCLCPP_WARN_STREAM(kLog, "left-eye JPEG encoder threads: " << n << ", slices per frame: " << cfg_.jpeg_slices);
  }
  frames_in_ = create_subscription<hbm_img_msgs::msg::HbmMsg1080P>(
      cfg_.image_topic, 10, [this](hbm_img_msgs::msg::HbmMsg1080P::ConstSharedPtr m) { OnStereoFrame(m); });
  targets_out_ = create_publisher<ai_msgs::msg::PerceptionTargets>("/Stereonet_node_sample", 10);
  disparity_out_ = create_publisher<sensor_msgs::msg::Image>(cfg_.output_topic, 10);
  ready_ = true;
}

void StereonetNode::DeclareAndReadParameters() {
  struct Item {
    const char* name;
    std::string* value;
  };
  const Item items[] = {{"config_file", &cfg_.config_file},
                        {"model_file", &cfg_.model_file},
                        {"sub_hbmem_topic_name", &cfg_.image_topic},
                        {"ros_img_topic_name", &cfg_.output_topic}};
  for (const Item& it : items) {
    declare_parameter<std::string>(it.name, *it.value);
    get_parameter<std::string>(it.name, *it.value);
  }
  if (const char* e = getenv("STEREONET_PUB_OUTPUT")) cfg_.publish_output = atoi(e) != 0;
  RCLCPP_WARN_STREAM(kLog, "\n config_file: " << cfg_.config_file << "\n model_file: " << cfg_.model_file
                                              << "\n sub_hbmem_topic_name: " << cfg_.image_topic
                                              << "\n ros_img_topic_name: " << cfg_.output_topic);
}

void StereonetNode::LogModelIo() {
  RCLCPP_WARN_STREAM(kLog, "model_input_count: " << net_->GetInputCount() << ", model_input_width: " << net_w_
                                                 << ", model_input_height: " << net_h_);
  hbDNNTensorProperties p;
  for (int i = 0; i < net_->GetInputCount(); ++i)
    if (hbDNNGetInputTensorProperties(&p, net_->GetDNNHandle(), i) == 0)
      RCLCPP_INFO_STREAM(kLog, "input_idx: " << i << ", tensorType = " << p.tensorType << ", tensorLayout = "
                                             << p.tensorLayout << ", shape " << p.validShape.dimensionSize[0] << "x"
                                             << p.validShape.dimensionSize[1] << "x" << p.validShape.dimensionSize[2]
                                             << "x" << p.validShape.dimensionSize[3]);
  for (int i = 0; i < net_->GetOutputCount(); ++i)
    if (hbDNNGetOutputTensorProperties(&p, net_->GetDNNHandle(), i) == 0)
      RCLCPP_WARN_STREAM(kLog, "output_idx: " << i << ", tensorType = " << p.tensorType
                                              << ", tensorLayout = " << p.tensorLayout);
}

int StereonetNode::SetNodePara() {
  if (!dnn_node_para_ptr_) return -1;
  if (access(cfg_.model_file.c_str(), F_OK) != 0) {
    RCLCPP_ERROR_STREAM(rclcpp::get_logger("hobot_stereonet"), "File is not exist! model_file: " << cfg_.model_file);
    return -1;
  }
  dnn_node_para_ptr_->model_file = cfg_.model_file;
  dnn_node_para_ptr_->model_task_type = hobot::dnn_node::ModelTaskType::ModelInferType;
  dnn_node_para_ptr_->task_num = 4;   // requests in flight
  return 0;
}

void StereonetNode::OnStereoFrame(const hbm_img_msgs::msg::HbmMsg1080P::ConstSharedPtr frame) {
  if (!rclcpp::ok() || !frame) return;

  // accept only NV12 frames that hold both eyes side by side at the model's resolution
  const char* enc = reinterpret_cast<const char*>(frame->encoding.data());
  if (strncmp(enc, "nv12", frame->encoding.size()) != 0) {
    RCLCPP_ERROR(kLog, "Only support nv12 img encoding!");
    return;
  }
  if ((int)frame->height != net_h_ || (int)frame->width != 2 * net_w_) {
    RCLCPP_ERROR_STREAM(kLog, "recved img msg h: " << frame->height << ", w: " << frame->width
                                                   << " is unmatch with model_input_width: " << net_w_
                                                   << ", model_input_height: " << net_h_);
    return;
  }
  const int w = net_w_, h = net_h_, pitch = 2 * w, rows = h + h / 2;
  if (frame->data.size() < (size_t)pitch * rows) return;

  auto request = std::make_shared<StereonetNodeOutput>();
  request->msg_header = std::make_shared<std_msgs::msg::Header>();
  request->msg_header->frame_id = std::to_string(frame->index);
  request->msg_header->stamp = frame->time_stamp;

  const auto t_pre = std::chrono::steady_clock::now();
  // STEREONET_INGEST=tensor keeps the reference's host steps (split both eyes, CvtNV12Data2Tensors, Run on the int8
  // tensor); the default hands the message payload to the backend, which does the same byte mapping on the GPU
  // The device ingest reads the frame in 8-byte units (sn_submit_nv12: the side-by-side width must be a multiple of 8,
  // the height even); any other even geometry the reference accepts goes through the host steps, as it does there.
  static const bool want_tensor = getenv("STEREONET_INGEST") != nullptr && !strcmp(getenv("STEREONET_INGEST"), "tensor");
  const bool host_tensor = want_tensor || ((2 * net_w_) & 7) != 0 || (net_h_ & 1) != 0;
  // de-interleave the eyes: every source row carries w bytes of the left eye, then w bytes of the right eye
  const unsigned char* row = frame->data.data();
  if (host_tensor) {
    eye_l_.resize((size_t)w * rows);
    eye_r_.resize((size_t)w * rows);
    for (int r = 0; r < rows; ++r, row += pitch) {
      memcpy(eye_l_.data() + (size_t)r * w, row, w);
      memcpy(eye_r_.data() + (size_t)r * w, row + w, w);
    }
  }
  std::vector<std::shared_ptr<DNNTensor>> tensors;
  if (host_tensor && pre_->CvtNV12Data2Tensors(tensors, net_, eye_l_.data(), eye_r_.data()) < 0) {
    RCLCPP_ERROR(kLog, "Preprocess fail");
    rclcpp::shutdown();
    return;
  }
  if (cfg_.publish_output) {
    // the left eye is the left half of every row of the side-by-side frame: the encoder reads it in place (pitch 2w); the
    // job keeps the message alive, nothing is copied on this thread
    auto left = std::make_shared<BinDataType>();
    left->w = w;
    left->h = h;
    request->sp_left_nv12 = left;
    const int quality = cfg_.jpeg_quality;
    const auto tq = std::chrono::steady_clock::now();
    request->jpeg_ready = SubmitSlicedJpeg(*jpeg_pool_, frame, frame->data.data(), w, h, pitch, quality, cfg_.jpeg_slices, left);
    g_stats.add(1, tq);
  }
  request->preprocess_time_ms = elapsed_ms(t_pre);
  RCLCPP_INFO(kLog, "Preprocess done, time cost %d ms", request->preprocess_time_ms);

  const auto tr = std::chrono::steady_clock::now();
  const int rc = host_tensor ? Run(tensors, request, /*is_sync_mode=*/false, -1, -1)
                             : RunSbsNv12(frame->data.data(), 2 * w, h, request, /*is_sync_mode=*/false, -1);
  g_stats.add(2, tr);
  g_stats.add(0, t_pre);
  if (rc < 0) {
    RCLCPP_ERROR(kLog, "Run infer fail!");
    return;
  }
  RCLCPP_INFO(kLog, "Run infer done");
}

namespace {
// one path per line; every entry must exist (stereonet_node.cpp:832-878)
bool read_list(const std::string& list_file, std::vector<std::string>& out) {
  std::ifstream in(list_file);
  if (!in.good()) {
    RCLCPP_ERROR_STREAM(kLog, "Open file failed: " << list_file);
    return false;
  }
  std::string line;
  while (std::getline(in, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
    if (access(line.c_str(), F_OK) != 0) {
      RCLCPP_ERROR_STREAM(kLog, "File is not exist! img_name: " << line);
      return false;
    }
    out.push_back(line);
  }
  return true;
}
}  // namespace

int StereonetNode::RunImglistFeedInfer(std::string left_img_list, std::string right_img_list) {
  if (!rclcpp::ok() || !ready_) return 0;
  RCLCPP_INFO_STREAM(kLog, "Feedback with left_img_list: " << left_img_list << " right_img_list: " << right_img_list);
  std::vector<std::string> left, right;
  if (!read_list(left_img_list, left) || !read_list(right_img_list, right)) {
    rclcpp::shutdown();
    return 0;
  }
  if (left.size() != right.size()) {
    RCLCPP_ERROR_STREAM(kLog, "Imgs size error! left_imgs.size: " << left.size() << ", right_imgs.size: " << right.size());
    rclcpp::shutdown();
    return 0;
  }
  int start_ms = cfg_.feed_start_pause_ms, frame_ms = cfg_.feed_frame_pause_ms;
  if (const char* e = getenv("STEREONET_FEED_PAUSE_MS")) start_ms = frame_ms = atoi(e);
  std::this_thread::sleep_for(std::chrono::milliseconds(start_ms));

  int done = 0;
  std::vector<uint8_t> bgr;
  std::vector<unsigned char> nv12[2];
  for (size_t idx = 0; idx < left.size(); ++idx) {
    RCLCPP_WARN_STREAM(kLog, "Feed " << idx << "/" << left.size());
    if (!rclcpp::ok()) return done;
    const std::string* path[2] = {&left[idx], &right[idx]};
    for (int eye = 0; eye < 2; ++eye) {
      int w = 0, h = 0;
      std::string why;
      if (!ReadImageBGR(*path[eye], w, h, bgr, &why)) {
        RCLCPP_ERROR_STREAM(kLog, "BGRToNv12 Fail: " << why);
        rclcpp::shutdown();
        return done;
      }
      // the reference feeds whatever imread returned; a size other than the model's would read out of bounds
      // there, so it is an error here
      if (w != net_w_ || h != net_h_ || Tools::BGRToNv12(bgr.data(), w, h, nv12[eye]) != 0) {
        RCLCPP_ERROR_STREAM(kLog, "BGRToNv12 Fail: " << *path[eye] << " is " << w << "x" << h << ", model input is "
                                                     << net_w_ << "x" << net_h_);
        rclcpp::shutdown();
        return done;
      }
    }
    auto request = std::make_shared<StereonetNodeOutput>();
    request->msg_header = std::make_shared<std_msgs::msg::Header>();
    request->msg_header->frame_id = std::to_string(idx);
    std::vector<std::shared_ptr<DNNTensor>> tensors;
    if (pre_->CvtNV12Data2Tensors(tensors, net_, nv12[0].data(), nv12[1].data()) < 0) {
      RCLCPP_ERROR(kLog, "Preprocess fail");
      rclcpp::shutdown();
      return done;
    }
    if (cfg_.publish_output) {
      auto jpg = std::make_shared<BinDataType>();
      jpg->w = net_w_;
      jpg->h = net_h_;
      if (!EncodeNv12ToJpeg(nv12[0].data(), net_w_, net_h_, net_w_, cfg_.jpeg_quality, jpg->jpeg)) {
        RCLCPP_ERROR(kLog, "invalid sp_left_nv12");
        rclcpp::shutdown();
        return done;
      }
      request->sp_left_nv12 = jpg;
    }
    if (Run(tensors, request, /*is_sync_mode=*/true, -1, -1) < 0) {
      RCLCPP_ERROR(kLog, "Run infer fail!");
      return done;
    }
    ++done;
    RCLCPP_INFO(kLog, "Run infer done");
    std::this_thread::sleep_for(std::chrono::milliseconds(frame_ms));
  }
  return done;
}

int StereonetNode::PostProcess(const std::shared_ptr<hobot::dnn_node::DnnNodeOutput>& node_output) {
  if (!rclcpp::ok()) return 0;
  auto request = std::dynamic_pointer_cast<StereonetNodeOutput>(node_output);
  if (!request) {
    RCLCPP_ERROR(kLog, "Cast dnn node output fail!");
    return -1;
  }
  const auto t_pub = std::chrono::steady_clock::now();
  int pack_ms = 0;
  const auto tw = std::chrono::steady_clock::now();
  const bool jpeg_bad = cfg_.publish_output && request->sp_left_nv12 && request->jpeg_ready.valid() && !request->jpeg_ready.get();
  g_stats.add(3, tw);
  if (jpeg_bad) {
    RCLCPP_ERROR(kLog, "invalid sp_left_nv12");      // the worker's encode failed (FeedImg's check, stereonet_node.cpp:797)
    rclcpp::shutdown();
    return -1;
  }
  if (cfg_.publish_output && request->sp_left_nv12 && !request->output_tensors.empty()) {
    // wire format consumed by the render node: sensor_msgs/Image, encoding "jpeg",
    // data = the raw int32 output tensor followed by the JPEG of the left eye, step = total length
    const hbSysMem& out = request->output_tensors[0]->sysMem[0];
    const std::vector<uint8_t>& jpeg = request->sp_left_nv12->jpeg;
    sensor_msgs::msg::Image msg;
    msg.header = *request->msg_header;
    msg.width = request->sp_left_nv12->w;
    msg.height = request->sp_left_nv12->h;
    msg.encoding = "jpeg";
    // one allocation, two copies, no zero fill (resize() + memcpy wrote the 3.7 MB twice)
    const uint8_t* raw = static_cast<const uint8_t*>(out.virAddr);
    msg.data.reserve(out.memSize + jpeg.size());
    msg.data.insert(msg.data.end(), raw, raw + out.memSize);
    msg.data.insert(msg.data.end(), jpeg.begin(), jpeg.end());
    msg.step = (uint32_t)msg.data.size();
    pack_ms = elapsed_ms(t_pub);
    RCLCPP_INFO(kLog, "publish output with msg index: %s, topic: %s, time cost ms: %d",
                request->msg_header->frame_id.c_str(), cfg_.output_topic.c_str(), pack_ms);
    g_stats.add(4, t_pub);
    const auto tp = std::chrono::steady_clock::now();
    disparity_out_->publish(std::move(msg));
    g_stats.add(5, tp);
  } else {
    RCLCPP_INFO(kLog, "publish is unable");
  }
  const auto& st = node_output->rt_stat;
  if (st && st->fps_updated)
    RCLCPP_WARN(kLog,
                "input fps: %.2f, out fps: %.2f, preprocess time ms: %d, infer time ms: %d, msg preparation for pub "
                "time cost ms: %d, refinement residual px: %.3f, arithmetic: %s",
                st->input_fps, st->output_fps, request->preprocess_time_ms, st->infer_time_ms, pack_ms,
                st->refine_residual_px, st->arithmetic);
  return 0;
}

}  // namespace stereonet
}  // namespace hobot
