// bin_data.h — BinDataType (reference: stereonet_infer/include/stereonet_node.h:40-47), in a header of its own so that the
// encoder threads (jpeg_pool.h) do not pull in the node.
#pragma once
#include <cstdint>
#include <vector>

namespace hobot {
namespace stereonet {

// JPEG of the left eye that travels with a request from FeedImg to PostProcess.
struct BinDataType {
  std::vector<uint8_t> jpeg;
  int w = 1280;
  int h = 720;
};

}  // namespace stereonet
}  // namespace hobot
