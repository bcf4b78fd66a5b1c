// stereonet_node.h — hobot::stereonet::StereonetNode with the reference's public surface
// (stereonet_infer/include/stereonet_node.h:40-126): same class, constructor signature, ROS parameters
// (config_file, model_file, sub_hbmem_topic_name, ros_img_topic_name), topics and output wire format, so the
// reference's launch files keep working; the inference behind DnnNode::Run is libstereonet_hip.so.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "ai_msgs/msg/perception_targets.hpp"
#include "dnn_node/dnn_node.h"
#include "hbm_img_msgs/msg/hbm_msg1080_p.hpp"
#include "preprocess.h"
#include "rclcpp/rclcpp.hpp"
#include "sensor_msgs/msg/image.hpp"

namespace hobot {
namespace stereonet {

using hobot::dnn_node::DNNTensor;
using hobot::dnn_node::DnnNodeOutput;
using hobot::dnn_node::Model;

struct BinDataType {
  char* data = nullptr;
  int len = 0;
  int w = 1280;
  int h = 720;
  std::vector<uint8_t> jpeg;
};

struct StereonetNodeOutput : public hobot::dnn_node::DnnNodeOutput {
  float ratio = 1.0;
  std::shared_ptr<BinDataType> sp_left_nv12 = nullptr;   // carries the JPEG of the left eye to PostProcess
  int preprocess_time_ms = 0;
};

class StereonetNode : public hobot::dnn_node::DnnNode {
 public:
  StereonetNode(const std::string& node_name = "stereonet_node",
                const rclcpp::NodeOptions& options = rclcpp::NodeOptions());

  bool IsReady() const { return model_ != nullptr; }

 protected:
  int SetNodePara() override;
  int PostProcess(const std::shared_ptr<hobot::dnn_node::DnnNodeOutput>& node_output) override;

 private:
  void FeedImg(const hbm_img_msgs::msg::HbmMsg1080P::ConstSharedPtr msg);

  Model* model_ = nullptr;
  int model_input_width_ = -1;
  int model_input_height_ = -1;
  std::vector<hbDNNTensorProperties> input_model_info_;
  std::vector<hbDNNTensorProperties> output_model_info_;

  rclcpp::Subscription<hbm_img_msgs::msg::HbmMsg1080P>::ConstSharedPtr subscription_hbmem_img_ = nullptr;
  std::string sub_hbmem_topic_name_ = "hbmem_stereo_img";
  rclcpp::Publisher<ai_msgs::msg::PerceptionTargets>::SharedPtr msg_publisher_ = nullptr;
  rclcpp::Publisher<sensor_msgs::msg::Image>::SharedPtr ros_img_publisher_ = nullptr;
  std::string ros_img_topic_name_ = "/stereonet_node_output";
  bool enable_pub_output_ = true;

  std::string config_file_ = "config/hobot_stereonet_config.json";
  std::string model_file_ = "config/hobot_stereonet.hbm";
  std::shared_ptr<PreProcess> sp_preprocess_ = nullptr;
};

}  // namespace stereonet
}  // namespace hobot
