#pragma once
#include <memory>
#include <string>
#include "builtin_interfaces/msg/time.hpp"
namespace std_msgs { namespace msg {
struct Header {
  using SharedPtr = std::shared_ptr<Header>;
  builtin_interfaces::msg::Time stamp;
  std::string frame_id;
  Header& set__frame_id(const std::string& v) { frame_id = v; return *this; }
  Header& set__stamp(const builtin_interfaces::msg::Time& v) { stamp = v; return *this; }
};
}}
