#pragma once
#include <memory>
#include "std_msgs/msg/header.hpp"
namespace ai_msgs { namespace msg {
struct PerceptionTargets {   // created but never published by the reference (stereonet_node.cpp:114-115)
  using SharedPtr = std::shared_ptr<PerceptionTargets>;
  std_msgs::msg::Header header;
};
}}
