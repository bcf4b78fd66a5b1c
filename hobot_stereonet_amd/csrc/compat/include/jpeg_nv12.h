// jpeg_nv12.h — baseline JPEG (YCbCr 4:2:0, ITU T.81 Annex K tables) straight from an NV12 image.
// Stands in for the reference's cv::cvtColor(NV12->BGR) + cv::imencode(".jpg") of the left eye
// (stereonet_infer/src/stereonet_node.cpp:749-786): OpenCV is not available here, and NV12 is already
// the Y/Cb/Cr 4:2:0 that baseline JPEG stores, so no colour conversion is needed.
#pragma once
#include <cstdint>
#include <vector>

namespace hobot {
namespace stereonet {
// `pitch` = bytes per source row (w for a contiguous eye, 2w for the left half of a side-by-side frame).
bool EncodeNv12ToJpeg(const uint8_t* nv12, int w, int h, int pitch, int quality, std::vector<uint8_t>& out);
// The exact-DCT form of round 3 (slow): what the tests compare EncodeNv12ToJpeg against.
bool EncodeNv12ToJpegReference(const uint8_t* nv12, int w, int h, int pitch, int quality, std::vector<uint8_t>& out);
}  // namespace stereonet
}  // namespace hobot
