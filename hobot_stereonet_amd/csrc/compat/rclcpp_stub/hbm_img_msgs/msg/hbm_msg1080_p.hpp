// Shape of hbm_img_msgs/msg/HbmMsg1080P (the hbmem zero-copy image message the camera node publishes,
// stereonet_infer/include/stereonet_node.h:26,95-96): fixed-capacity payload, fields as the reference reads
// them (stereonet_node.cpp:663-738).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <vector>
#include "builtin_interfaces/msg/time.hpp"
namespace hbm_img_msgs { namespace msg {
struct HbmMsg1080P {
  using SharedPtr = std::shared_ptr<HbmMsg1080P>;
  using ConstSharedPtr = std::shared_ptr<const HbmMsg1080P>;
  uint32_t index = 0;
  builtin_interfaces::msg::Time time_stamp;
  uint32_t height = 0, width = 0;
  uint32_t data_size = 0;
  std::array<uint8_t, 12> encoding{};
  std::vector<uint8_t> data;   // the real message is std::array<uint8_t, 6220800>; a vector keeps tests light
};
}}
