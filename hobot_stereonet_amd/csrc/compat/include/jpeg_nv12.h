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
// The same image as a stream with a restart interval of `rows_per_slice` MCU rows (DRI + RSTm markers): every slice is coded
// independently (DC predictors reset, byte aligned), so slices can be encoded on different threads; decodes to exactly the
// image of EncodeNv12ToJpeg.  The pieces, for callers that run the slices themselves (the node's encoder threads):
//   JpegAppendHeader   SOI .. SOS (restart_mcus = MCUs per slice, 0 = no DRI segment)
//   JpegAppendMcuRows  stuffed entropy-coded bytes of MCU rows [row0, row1), byte aligned, no marker
// a stream = header, slice 0, FF D0, slice 1, FF D1, ... (RSTm counts modulo 8), last slice, FF D9.
bool EncodeNv12ToJpegSliced(const uint8_t* nv12, int w, int h, int pitch, int quality, int rows_per_slice,
                            std::vector<uint8_t>& out);
int JpegMcuRows(int h);
bool JpegAppendHeader(int w, int h, int quality, int restart_mcus, std::vector<uint8_t>& out);
bool JpegAppendMcuRows(const uint8_t* nv12, int w, int h, int pitch, int quality, int row0, int row1, std::vector<uint8_t>& out);
// The exact-DCT form of round 3 (slow): what the tests compare EncodeNv12ToJpeg against.
bool EncodeNv12ToJpegReference(const uint8_t* nv12, int w, int h, int pitch, int quality, std::vector<uint8_t>& out);
}  // namespace stereonet
}  // namespace hobot
