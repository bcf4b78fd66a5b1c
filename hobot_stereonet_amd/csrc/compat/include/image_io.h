// image_io.h — file readers/writers of the offline (file-list) feeder.  The reference reads its list entries
// with cv::imread(path, cv::IMREAD_COLOR) (stereonet_infer/src/stereonet_node.cpp:901,913), i.e. "any image
// file -> 8-bit BGR".  OpenCV is not part of this build, so the formats stereo datasets actually ship are
// decoded here: PNG (8-bit gray / RGB / RGBA / palette, non-interlaced; zlib inflate), binary PPM/PGM and
// uncompressed 24/32-bit BMP.  PFM is the float disparity format of SceneFlow / Middlebury.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace hobot {
namespace stereonet {

// -> interleaved B,G,R bytes, row-major, top row first (the cv::Mat CV_8UC3 memory order).
// Returns false with a message in *err when the file is missing, truncated or of an unsupported kind.
bool ReadImageBGR(const std::string& path, int& w, int& h, std::vector<uint8_t>& bgr, std::string* err = nullptr);

// Little-endian single-channel PFM ("Pf", scale -1.0, bottom row first).
bool WritePFM(const std::string& path, const float* data, int w, int h);
bool ReadPFM(const std::string& path, int& w, int& h, std::vector<float>& data, std::string* err = nullptr);

bool WriteBytes(const std::string& path, const void* data, size_t n);

}  // namespace stereonet
}  // namespace hobot
