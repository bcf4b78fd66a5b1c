// filelist_harness.cpp — the non-ROS command line around StereonetNode::RunImglistFeedInfer (reference:
// stereonet_infer/src/stereonet_node.cpp:820-976, the feeder its authors used for offline runs):
//   stereonet_filelist <model.snw> <left.list> <right.list> <out_dir>
// For every published frame it writes <out_dir>/<frame_id>.raw.bin (the int32 output tensor),
// <frame_id>.disp.pfm (disparity in pixels, Parse()'s formula) and <frame_id>.jpg (the left eye), and prints
// one line.  Exit codes: 2 usage, 3 node did not initialise, 5 the feeder stopped early (bad list / image).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

#include "image_io.h"
#include "parser.h"
#include "stereonet_node.h"

using namespace hobot::stereonet;

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s model.snw left.list right.list out_dir\n", argv[0]);
    return 2;
  }
  const std::string model = argv[1], left = argv[2], right = argv[3], out_dir = argv[4];
  rclcpp::init(argc, argv);
  rclcpp::NodeOptions opt;
  opt.append_parameter_override("model_file", model);
  auto node = std::make_shared<StereonetNode>("stereonet_node", opt);
  if (!rclcpp::ok() || !node->IsReady()) {
    fprintf(stderr, "node init failed\n");
    return 3;
  }
  int received = 0;
  rclcpp::Node listener("listener");
  auto sub = listener.create_subscription<sensor_msgs::msg::Image>(
      "stereonet_node_output", 10, [&](sensor_msgs::msg::Image::ConstSharedPtr m) {
        const size_t px = (size_t)m->width * m->height, tensor_bytes = px * 4;
        if (m->data.size() < tensor_bytes) return;
        const std::string stem = out_dir + "/" + m->header.frame_id;
        WriteBytes(stem + ".raw.bin", m->data.data(), tensor_bytes);
        WriteBytes(stem + ".jpg", m->data.data() + tensor_bytes, m->data.size() - tensor_bytes);
        // payload -> pixels exactly as the render node does (publisher_member_function.py:64-72):
        // depth = raw * 2.60443857769133e-6, disparity = depth * 16 * 12
        std::vector<float> disp(px);
        const uint32_t* raw = reinterpret_cast<const uint32_t*>(m->data.data());
        for (size_t i = 0; i < px; ++i) disp[i] = (float)((double)raw[i] * 2.60443857769133e-6 * 16.0 * 12.0);
        WritePFM(stem + ".disp.pfm", disp.data(), (int)m->width, (int)m->height);
        printf("frame_id=%s height=%u width=%u encoding=%s len=%zu jpeg=%zu\n", m->header.frame_id.c_str(), m->height,
               m->width, m->encoding.c_str(), m->data.size(), m->data.size() - tensor_bytes);
        ++received;
      });
  const int fed = node->RunImglistFeedInfer(left, right);
  printf("fed=%d received=%d\n", fed, received);
  const bool stopped = !rclcpp::ok();
  node.reset();
  rclcpp::shutdown();
  return stopped ? 5 : 0;
}
