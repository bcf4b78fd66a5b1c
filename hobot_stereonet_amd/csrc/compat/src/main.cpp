// Entry point of the `hobot_stereonet` executable (same role as stereonet_infer/src/main.cpp): bring ROS up,
// run one StereonetNode until shutdown.  A node that failed to initialise has already requested shutdown, so
// spin() returns at once and the process exits non-zero.
//   hobot_stereonet [--imglist left.list right.list]
// --imglist runs the offline feeder first (the call the reference keeps commented out in its constructor,
// stereonet_node.cpp:120) and then keeps spinning like a live node.
#include <cstdio>
#include <cstring>

#include "stereonet_node.h"

int main(int argc, char** argv) {
  rclcpp::init(argc, argv);
  int rc = 0;
  {
    auto node = std::make_shared<hobot::stereonet::StereonetNode>();
    if (!node->IsReady()) {
      fprintf(stderr, "hobot_stereonet: node did not initialise (model file / GPU), exiting\n");
      rc = 1;
    }
    for (int i = 1; i + 2 < argc; ++i)
      if (!strcmp(argv[i], "--imglist")) node->RunImglistFeedInfer(argv[i + 1], argv[i + 2]);
    rclcpp::spin(node);
  }
  rclcpp::shutdown();
  return rc;
}
