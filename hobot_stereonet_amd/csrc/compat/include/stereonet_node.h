// stereonet_node.h — the ROS 2 node of this repo's host mirror.
//
// Public contract kept identical to the reference node so that launch files and downstream nodes need no
// change (reference: stereonet_infer/include/stereonet_node.h:61-71 for the class surface, :97,:106,:120-121
// for the parameter defaults):
//   class hobot::stereonet::StereonetNode : public hobot::dnn_node::DnnNode
//   StereonetNode(node_name = "stereonet_node", options)
//   overrides SetNodePara() / PostProcess(output)
//   parameters config_file, model_file, sub_hbmem_topic_name, ros_img_topic_name
// Everything private is this implementation's own: settings are grouped in one struct, per-frame scratch is
// reused across frames, and the inference behind DnnNode::Run is libstereonet_hip.so.
#pragma once

#include <cstdint>
#include <future>
#include <memory>
#include <string>
#include <vector>

#include "rclcpp/rclcpp.hpp"
#include "sensor_msgs/msg/image.hpp"
#include "ai_msgs/msg/perception_targets.hpp"
#include "hbm_img_msgs/msg/hbm_msg1080_p.hpp"

#include "dnn_node/dnn_node.h"
#include "bin_data.h"
#include "preprocess.h"

namespace hobot {
namespace stereonet {

// Per-request context handed through DnnNode::Run.
struct StereonetNodeOutput : public hobot::dnn_node::DnnNodeOutput {
  std::shared_ptr<BinDataType> sp_left_nv12;
  int preprocess_time_ms = 0;
  // this implementation's own: set when the left-eye JPEG of this request is being encoded on a worker thread
  // (sp_left_nv12->jpeg is complete once it yields true); PostProcess waits for it
  std::shared_future<bool> jpeg_ready;
};

class JpegPool;      // worker threads that encode the left-eye JPEGs of requests in flight (stereonet_node.cpp)

class StereonetNode : public hobot::dnn_node::DnnNode {
 public:
  StereonetNode(const std::string& node_name = "stereonet_node",
                const rclcpp::NodeOptions& options = rclcpp::NodeOptions());

  // true once the model is loaded and the subscription exists (the harness checks it; the reference shuts
  // rclcpp down instead, which this node does as well)
  bool IsReady() const { return ready_; }

  // Offline feeder (reference: stereonet_node.h:118, stereonet_node.cpp:820-976, disabled in its constructor):
  // two text files with one image path per line; frame i = (left[i], right[i]) -> BGR -> NV12 -> model tensor,
  // synchronous Run with frame_id = i, result published like a live frame.  Any unreadable list or image, a
  // length mismatch, or an image that is not the model's size ends the run with rclcpp::shutdown(), as the
  // reference does.  Returns the number of frames that went through PostProcess.
  int RunImglistFeedInfer(std::string left_img_list, std::string right_img_list);

 protected:
  int SetNodePara() override;
  int PostProcess(const std::shared_ptr<hobot::dnn_node::DnnNodeOutput>& node_output) override;

 private:
  struct Settings {
    std::string config_file = "config/hobot_stereonet_config.json";   // declared, never opened (as in the reference)
    std::string model_file = "config/hobot_stereonet.hbm";
    std::string image_topic = "hbmem_stereo_img";
    std::string output_topic = "/stereonet_node_output";
    bool publish_output = true;       // the reference's enable_pub_output_ (a constant there); STEREONET_PUB_OUTPUT=0 turns it off
    int jpeg_quality = 95;
    int jpeg_threads = 0;             // 0 = hardware threads / 4, clamped to 2..32; STEREONET_JPEG_THREADS overrides
    int jpeg_slices = 8;              // restart-interval slices per frame, encoded in parallel (1 = one scan, no RSTm);
                                      // STEREONET_JPEG_SLICES overrides
    int feed_start_pause_ms = 1000;   // the reference waits for the viewer before / between offline frames
    int feed_frame_pause_ms = 300;    // (stereonet_node.cpp:890,974); STEREONET_FEED_PAUSE_MS overrides both
  };

  void DeclareAndReadParameters();
  void LogModelIo();
  void OnStereoFrame(const hbm_img_msgs::msg::HbmMsg1080P::ConstSharedPtr frame);   // the FeedImg role

  Settings cfg_;
  bool ready_ = false;
  int net_w_ = -1, net_h_ = -1;
  hobot::dnn_node::Model* net_ = nullptr;
  std::unique_ptr<PreProcess> pre_;
  std::vector<unsigned char> eye_l_, eye_r_;     // split NV12 eyes, reused across frames
  std::shared_ptr<JpegPool> jpeg_pool_;          // live path: the left eye is encoded off the executor thread

  rclcpp::Subscription<hbm_img_msgs::msg::HbmMsg1080P>::ConstSharedPtr frames_in_;
  rclcpp::Publisher<sensor_msgs::msg::Image>::SharedPtr disparity_out_;
  rclcpp::Publisher<ai_msgs::msg::PerceptionTargets>::SharedPtr targets_out_;   // created for parity, never used
};

}  // namespace stereonet
}  // namespace hobot
