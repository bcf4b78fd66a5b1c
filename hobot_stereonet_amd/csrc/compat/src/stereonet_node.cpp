// stereonet_node.cpp — behaviour of the reference node's live path, re-implemented:
//   constructor / parameters / Init      stereonet_infer/src/stereonet_node.cpp:24-127
//   SetNodePara                          :129-147
//   FeedImg                              :657-818
//   PostProcess                          :980-1089
// The reference's disabled offline feeders and dump helpers (:149-655, :820-976) are out of scope.
#include "stereonet_node.h"

#include <unistd.h>

#include <chrono>
#include <cstring>

#include "jpeg_nv12.h"

namespace hobot {
namespace stereonet {

StereonetNode::StereonetNode(const std::string& node_name, const rclcpp::NodeOptions& options)
    : hobot::dnn_node::DnnNode(node_name, options) {
  this->declare_parameter<std::string>("config_file", config_file_);
  this->declare_parameter<std::string>("model_file", model_file_);
  this->declare_parameter<std::string>("sub_hbmem_topic_name", sub_hbmem_topic_name_);
  this->declare_parameter<std::string>("ros_img_topic_name", ros_img_topic_name_);
  this->get_parameter<std::string>("config_file", config_file_);
  this->get_parameter<std::string>("model_file", model_file_);
  this->get_parameter<std::string>("sub_hbmem_topic_name", sub_hbmem_topic_name_);
  this->get_parameter<std::string>("ros_img_topic_name", ros_img_topic_name_);

  RCLCPP_WARN_STREAM(rclcpp::get_logger("stereonet_node"),
                     "\n config_file: " << config_file_ << "\n model_file: " << model_file_
                                        << "\n sub_hbmem_topic_name: " << sub_hbmem_topic_name_
                                        << "\n ros_img_topic_name: " << ros_img_topic_name_);

  if (Init() != 0 || GetModelInputSize(0, model_input_width_, model_input_height_) < 0) {
    RCLCPP_ERROR(rclcpp::get_logger("stereonet_node"), "Node init fail!");
    rclcpp::shutdown();
    return;
  }
  model_ = GetModel();
  if (!model_) {
    RCLCPP_ERROR(rclcpp::get_logger(""), "Invalid model");
    rclcpp::shutdown();
    return;
  }
  RCLCPP_WARN_STREAM(rclcpp::get_logger("stereonet_node"),
                     "model_input_count: " << model_->GetInputCount() << ", model_input_width: " << model_input_width_
                                           << ", model_input_height: " << model_input_height_);
  hbDNNHandle_t handle = model_->GetDNNHandle();
  input_model_info_.resize(model_->GetInputCount());
  for (int i = 0; i < model_->GetInputCount(); ++i) {
    hbDNNGetInputTensorProperties(&input_model_info_[i], handle, i);
    RCLCPP_INFO_STREAM(rclcpp::get_logger(""), "input_idx: " << i << ", tensorType = " << input_model_info_[i].tensorType
                                                             << ", tensorLayout = " << input_model_info_[i].tensorLayout);
  }
  output_model_info_.resize(model_->GetOutputCount());
  for (int i = 0; i < model_->GetOutputCount(); ++i) {
    hbDNNGetOutputTensorProperties(&output_model_info_[i], handle, i);
    RCLCPP_WARN_STREAM(rclcpp::get_logger(""), "output_idx: " << i << ", tensorType = " << output_model_info_[i].tensorType
                                                              << ", tensorLayout = " << output_model_info_[i].tensorLayout);
  }

  sp_preprocess_ = std::make_shared<PreProcess>("");
  subscription_hbmem_img_ = this->create_subscription<hbm_img_msgs::msg::HbmMsg1080P>(
      sub_hbmem_topic_name_, 10, std::bind(&StereonetNode::FeedImg, this, std::placeholders::_1));
  msg_publisher_ = this->create_publisher<ai_msgs::msg::PerceptionTargets>("/Stereonet_node_sample", 10);
  ros_img_publisher_ = this->create_publisher<sensor_msgs::msg::Image>(ros_img_topic_name_, 10);
}

int StereonetNode::SetNodePara() {
  if (!dnn_node_para_ptr_) return -1;
  if (access(model_file_.c_str(), F_OK) != 0) {
    RCLCPP_ERROR_STREAM(rclcpp::get_logger("hobot_stereonet"), "File is not exist! model_file: " << model_file_);
    return -1;
  }
  dnn_node_para_ptr_->model_file = model_file_;
  dnn_node_para_ptr_->model_task_type = hobot::dnn_node::ModelTaskType::ModelInferType;
  dnn_node_para_ptr_->task_num = 4;
  return 0;
}

void StereonetNode::FeedImg(const hbm_img_msgs::msg::HbmMsg1080P::ConstSharedPtr img_msg) {
  if (!rclcpp::ok() || !img_msg) return;
  // 1. only NV12 is handled
  if ("nv12" != std::string(reinterpret_cast<const char*>(img_msg->encoding.data()))) {
    RCLCPP_ERROR(rclcpp::get_logger("stereonet_node"), "Only support nv12 img encoding!");
    return;
  }
  // side-by-side frame: width = 2 * model width, height = model height
  if (img_msg->height != static_cast<uint32_t>(model_input_height_) ||
      img_msg->width != static_cast<uint32_t>(model_input_width_) * 2) {
    RCLCPP_ERROR_STREAM(rclcpp::get_logger("stereonet_node"),
                        "recved img msg h: " << img_msg->height << ", w: " << img_msg->width
                                             << " is unmatch with model_input_width: " << model_input_width_
                                             << ", model_input_height: " << model_input_height_);
    return;
  }
  const int w = img_msg->width / 2, h = img_msg->height, pitch = img_msg->width;
  if (img_msg->data.size() < (size_t)pitch * h * 3 / 2) return;

  // 2. output holder: header carries the frame index and the camera time stamp
  auto dnn_output = std::make_shared<StereonetNodeOutput>();
  dnn_output->msg_header = std::make_shared<std_msgs::msg::Header>();
  dnn_output->msg_header->set__frame_id(std::to_string(img_msg->index));
  dnn_output->msg_header->set__stamp(img_msg->time_stamp);

  // 3. pre-processing: split the eyes (h luma rows then h/2 chroma rows each), build the model input
  const auto tp_start = std::chrono::system_clock::now();
  const size_t eye = (size_t)w * h * 3 / 2;
  std::vector<unsigned char> left(eye), right(eye);
  const unsigned char* src = img_msg->data.data();
  for (int r = 0; r < h + h / 2; ++r) {
    memcpy(&left[(size_t)r * w], src + (size_t)r * pitch, w);
    memcpy(&right[(size_t)r * w], src + (size_t)r * pitch + w, w);
  }
  std::vector<std::shared_ptr<DNNTensor>> input_tensors;
  if (sp_preprocess_->CvtNV12Data2Tensors(input_tensors, model_, left.data(), right.data()) < 0) {
    RCLCPP_ERROR(rclcpp::get_logger("stereonet_node"), "Preprocess fail");
    rclcpp::shutdown();
    return;
  }
  if (enable_pub_output_) {   // JPEG of the left eye rides along with the model output
    auto bin = std::make_shared<BinDataType>();
    bin->w = w;
    bin->h = h;
    if (!EncodeNv12ToJpeg(left.data(), w, h, w, 95, bin->jpeg)) {
      RCLCPP_ERROR(rclcpp::get_logger("stereonet_node"), "invalid sp_left_nv12");
      rclcpp::shutdown();
      return;
    }
    dnn_output->sp_left_nv12 = bin;
  }
  const auto interval =
      std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - tp_start).count();
  RCLCPP_INFO(rclcpp::get_logger("stereonet_node"), "Preprocess done, time cost %d ms", (int)interval);
  dnn_output->preprocess_time_ms = (int)interval;

  if (Run(input_tensors, dnn_output, false, -1, -1) < 0) {
    RCLCPP_ERROR(rclcpp::get_logger("stereonet_node"), "Run infer fail!");
    return;
  }
  RCLCPP_INFO(rclcpp::get_logger("stereonet_node"), "Run infer done");
}

int StereonetNode::PostProcess(const std::shared_ptr<hobot::dnn_node::DnnNodeOutput>& node_output) {
  if (!rclcpp::ok()) return 0;
  const auto tp_start = std::chrono::system_clock::now();
  auto out = std::dynamic_pointer_cast<StereonetNodeOutput>(node_output);
  if (!out) {
    RCLCPP_ERROR(rclcpp::get_logger("stereonet_node"), "Cast dnn node output fail!");
    return -1;
  }
  int interval = 0;
  if (enable_pub_output_ && out->sp_left_nv12 && !out->output_tensors.empty()) {
    // wire format: sensor_msgs/Image, encoding "jpeg", data = raw int32 tensor bytes || JPEG(left), step = len
    sensor_msgs::msg::Image msg;
    msg.height = out->sp_left_nv12->h;
    msg.width = out->sp_left_nv12->w;
    msg.encoding = "jpeg";
    msg.header = *out->msg_header;
    const char* infer = reinterpret_cast<const char*>(out->output_tensors[0]->sysMem[0].virAddr);
    const size_t infer_len = out->output_tensors[0]->sysMem[0].memSize;
    const auto& jpeg = out->sp_left_nv12->jpeg;
    msg.step = (uint32_t)(infer_len + jpeg.size());
    msg.data.resize(infer_len + jpeg.size());
    memcpy(msg.data.data(), infer, infer_len);
    memcpy(msg.data.data() + infer_len, jpeg.data(), jpeg.size());
    interval = (int)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - tp_start)
                   .count();
    RCLCPP_INFO(rclcpp::get_logger("stereonet_node"), "publish output with msg index: %s, topic: %s, time cost ms: %d",
                out->msg_header->frame_id.data(), ros_img_topic_name_.data(), interval);
    ros_img_publisher_->publish(std::move(msg));
  } else {
    RCLCPP_INFO(rclcpp::get_logger("stereonet_node"), "publish is unable");
  }
  if (node_output->rt_stat && node_output->rt_stat->fps_updated) {
    RCLCPP_WARN(rclcpp::get_logger("stereonet_node"),
                "input fps: %.2f, out fps: %.2f, preprocess time ms: %d, infer time ms: %d, msg preparation for pub "
                "time cost ms: %d",
                node_output->rt_stat->input_fps, node_output->rt_stat->output_fps, out->preprocess_time_ms,
                node_output->rt_stat->infer_time_ms, interval);
  }
  return 0;
}

}  // namespace stereonet
}  // namespace hobot
