// main.cpp — same entry point as stereonet_infer/src/main.cpp:17-22.
#include "stereonet_node.h"

int main(int argc, char** argv) {
  rclcpp::init(argc, argv);
  rclcpp::spin(std::make_shared<hobot::stereonet::StereonetNode>());
  rclcpp::shutdown();
  return 0;
}
