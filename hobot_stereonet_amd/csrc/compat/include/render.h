// render.h — C++ twin of the reference's render node, the consumer of /stereonet_node_output
// (stereonet_render_tools/hobot_stereonet_render/publisher_member_function.py:45-163).
// The reference is Python + cv2; this restates the three cv2 calls of its colourising step so that a C++ host can
// publish /image_jpeg without Python:  payload split (:57-66, uint32 view), dequantisation and depth (:72-81),
// convertScaleAbs(alpha = 9) + COLORMAP_JET (:82), and the vertical stack under the left image with the reference's
// accidental R/B swap of the colour map (:98-137).  JPEG decode / encode stay with the caller (jpeg_nv12.h encodes).
// The JET table is the closed form of OpenCV's colormap.cpp — PARITY UNPINNED (no cv2 in this image to diff against);
// hobot_stereonet_amd/render.py is the same arithmetic in numpy and the two are tested against each other.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace hobot {
namespace stereonet {

struct RenderConstants {
  double scale = 0.00000260443857769133;    // publisher_member_function.py:29
  double focal = 527.1931762695312;         // :30
  double baseline = 119.89382172;           // :31
  double alpha = 9.0;                       // :82 (the dead C++ path of the reference uses 11, parser.cpp:116)
};

// COLORMAP_JET, 256 x (B, G, R)
const uint8_t* JetLutBGR();
// cv::convertScaleAbs for one value: saturate_cast<uint8>(|v * alpha|), round-half-even, NaN -> 0, inf -> 255
uint8_t ConvertScaleAbs(double v, double alpha);
// payload = int32 tensor (viewed as uint32, as the reference does) || JPEG.  Returns false if the payload is too short.
// disp / depth (nullable): w*h doubles; color_bgr: w*h*3 bytes in cv2's BGR order; jpeg_off: where the JPEG starts.
bool RenderDepth(const uint8_t* payload, size_t len, int w, int h, const RenderConstants& k, double* disp, double* depth,
                 uint8_t* color_bgr, size_t* jpeg_off);
// the published canvas: left image (true RGB, w*h*3) on top, the BGR colour map reinterpreted as RGB below (the
// reference's PIL round trip) -> 2h x w x 3 RGB
void StackJoint(const uint8_t* left_rgb, const uint8_t* color_bgr, int w, int h, std::vector<uint8_t>& joint_rgb);

}  // namespace stereonet
}  // namespace hobot
