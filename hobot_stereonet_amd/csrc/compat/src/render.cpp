// render.cpp — see render.h.  Follows publisher_member_function.py:57-137 of the reference's render node.
#include "render.h"

#include <cmath>
#include <cstring>

namespace hobot {
namespace stereonet {

namespace {
struct Jet {
  uint8_t t[256 * 3];
  Jet() {
    auto ramp = [](double x, double centre) {
      double v = 1.5 - std::fabs(4.0 * x - centre);
      return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    };
    for (int i = 0; i < 256; ++i) {
      const double x = i / 255.0;
      t[3 * i + 0] = (uint8_t)std::nearbyint(ramp(x, 1.0) * 255.0);   // B
      t[3 * i + 1] = (uint8_t)std::nearbyint(ramp(x, 2.0) * 255.0);   // G
      t[3 * i + 2] = (uint8_t)std::nearbyint(ramp(x, 3.0) * 255.0);   // R
    }
  }
};
}  // namespace

const uint8_t* JetLutBGR() {
  static const Jet jet;
  return jet.t;
}

uint8_t ConvertScaleAbs(double v, double alpha) {
  const double a = std::fabs(v * alpha);
  if (std::isnan(a)) return 0;
  const double r = std::nearbyint(a);      // round-half-even in the default rounding mode, as cvRound
  return r >= 255.0 ? 255 : (uint8_t)r;
}

bool RenderDepth(const uint8_t* payload, size_t len, int w, int h, const RenderConstants& k, double* disp, double* depth,
                 uint8_t* color_bgr, size_t* jpeg_off) {
  const size_t n = (size_t)w * h;
  if (!payload || w <= 0 || h <= 0 || len < n * 4) return false;
  if (jpeg_off) *jpeg_off = n * 4;
  const uint8_t* lut = JetLutBGR();
  for (size_t i = 0; i < n; ++i) {
    uint32_t raw;
    memcpy(&raw, payload + 4 * i, 4);                          // :65-66 np.uint32 view
    const double d = (double)raw * k.scale * 16.0 * 12.0;      // :73-75
    const double z = k.focal * k.baseline / d / 1000.0;        // :81 (zero disparity -> inf, kept)
    if (disp) disp[i] = d;
    if (depth) depth[i] = z;
    if (color_bgr) memcpy(color_bgr + 3 * i, lut + 3 * ConvertScaleAbs(z, k.alpha), 3);   // :82
  }
  return true;
}

void StackJoint(const uint8_t* left_rgb, const uint8_t* color_bgr, int w, int h, std::vector<uint8_t>& joint_rgb) {
  const size_t n = (size_t)w * h * 3;
  joint_rgb.resize(2 * n);
  memcpy(joint_rgb.data(), left_rgb, n);
  memcpy(joint_rgb.data() + n, color_bgr, n);                  // BGR bytes published as if they were RGB (:108-137)
}

}  // namespace stereonet
}  // namespace hobot
