// test_hooks.cpp — extern "C" entry points so the CPU test-suite can exercise the host mirror's pure
// functions through ctypes without a GPU or ROS (tests/test_host_mirror.py).
#include <cstring>

#include "image_io.h"
#include "jpeg_nv12.h"
#include "jpeg_pool.h"
#include "parser.h"
#include "preprocess.h"
#include "render.h"

extern "C" {

// render twin: payload -> disparity, depth, BGR colour map and the stacked RGB canvas (left_rgb given by the caller)
int snhost_render(const unsigned char* payload, long len, int w, int h, double* disp, double* depth,
                  unsigned char* color_bgr, const unsigned char* left_rgb, unsigned char* joint_rgb) {
  hobot::stereonet::RenderConstants k;
  size_t off = 0;
  if (!hobot::stereonet::RenderDepth(payload, (size_t)len, w, h, k, disp, depth, color_bgr, &off)) return -1;
  if (left_rgb && joint_rgb) {
    std::vector<uint8_t> j;
    hobot::stereonet::StackJoint(left_rgb, color_bgr, w, h, j);
    memcpy(joint_rgb, j.data(), j.size());
  }
  return (int)off;
}

void snhost_jet_lut(unsigned char* out) { memcpy(out, hobot::stereonet::JetLutBGR(), 768); }

void snhost_yuv420_to_yuv444(const unsigned char* in, unsigned char* out, int w, int h) {
  hobot::stereonet::Tools::YUV420TOYUV444(in, out, w, h);
}

int snhost_quantize_byte(int b) {
  return hobot::stereonet::PreProcess::Quantize(((float)b - 128.0) / 128.0);   // preprocess.cpp:1038 call-site form
}

int snhost_bgr_to_nv12(const unsigned char* bgr, int w, int h, unsigned char* nv12) {
  std::vector<unsigned char> out;
  if (hobot::stereonet::Tools::BGRToNv12(bgr, w, h, out) != 0) return -1;
  memcpy(nv12, out.data(), out.size());
  return 0;
}

// decodes an image file; returns 0 and w/h, copying at most cap bytes of BGR (call once with cap 0 for the size)
int snhost_read_image_bgr(const char* path, int* w, int* h, unsigned char* bgr, long cap) {
  std::vector<uint8_t> px;
  std::string err;
  if (!hobot::stereonet::ReadImageBGR(path, *w, *h, px, &err)) return -1;
  if (bgr && cap > 0) memcpy(bgr, px.data(), (size_t)cap < px.size() ? (size_t)cap : px.size());
  return 0;
}

int snhost_pfm_roundtrip(const char* path, const float* in, int w, int h, float* out) {
  if (!hobot::stereonet::WritePFM(path, in, w, h)) return -1;
  int rw = 0, rh = 0;
  std::vector<float> d;
  if (!hobot::stereonet::ReadPFM(path, rw, rh, d) || rw != w || rh != h) return -2;
  memcpy(out, d.data(), sizeof(float) * d.size());
  return 0;
}

// returns the JPEG size (<= cap) or -1
long snhost_jpeg_nv12(const unsigned char* nv12, int w, int h, int pitch, int quality, unsigned char* out, long cap) {
  std::vector<uint8_t> j;
  if (!hobot::stereonet::EncodeNv12ToJpeg(nv12, w, h, pitch, quality, j) || (long)j.size() > cap) return -1;
  memcpy(out, j.data(), j.size());
  return (long)j.size();
}

// restart-interval form (rows_per_slice MCU rows per slice); returns the JPEG size (<= cap) or -1
long snhost_jpeg_nv12_sliced(const unsigned char* nv12, int w, int h, int pitch, int quality, int rows_per_slice, unsigned char* out, long cap) {
  std::vector<uint8_t> j;
  if (!hobot::stereonet::EncodeNv12ToJpegSliced(nv12, w, h, pitch, quality, rows_per_slice, j) || (long)j.size() > cap) return -1;
  memcpy(out, j.data(), j.size());
  return (long)j.size();
}

// The node's encoder-thread path without the node: `frames` copies of one image go through a JpegPool of `threads`
// threads as `slices` slices each, all in flight together; every assembled stream must equal EncodeNv12ToJpegSliced's.
// Returns the stream size (written to out, <= cap), -1 on an encoder failure, -2 on a mismatch.
long snhost_jpeg_pool(const unsigned char* nv12, int w, int h, int pitch, int quality, int threads, int slices, int frames,
                      unsigned char* out, long cap) {
  using namespace hobot::stereonet;
  const int rows = JpegMcuRows(h);
  int nsl = slices < 1 ? 1 : (slices > rows ? rows : slices);
  const int per = (rows + nsl - 1) / nsl;
  std::vector<uint8_t> want;
  if (!EncodeNv12ToJpegSliced(nv12, w, h, pitch, quality, per, want)) return -1;
  std::vector<std::shared_ptr<BinDataType>> outs;
  std::vector<std::shared_future<bool>> futs;
  {
    JpegPool pool(threads);
    for (int f = 0; f < frames; ++f) {
      outs.push_back(std::make_shared<BinDataType>());
      futs.push_back(SubmitSlicedJpeg(pool, nullptr, nv12, w, h, pitch, quality, slices, outs.back()));
    }
    for (auto& fu : futs)
      if (!fu.get()) return -1;
  }      // the pool joins its threads here
  for (auto& o : outs)
    if (o->jpeg != want) return -2;
  if ((long)want.size() > cap) return -1;
  memcpy(out, want.data(), want.size());
  return (long)want.size();
}

// the exact-DCT form of round 3 (the checker of the fast encoder); returns the JPEG size (<= cap) or -1
long snhost_jpeg_nv12_reference(const unsigned char* nv12, int w, int h, int pitch, int quality, unsigned char* out, long cap) {
  std::vector<uint8_t> j;
  if (!hobot::stereonet::EncodeNv12ToJpegReference(nv12, w, h, pitch, quality, j) || (long)j.size() > cap) return -1;
  memcpy(out, j.data(), j.size());
  return (long)j.size();
}

// Parse() on a caller-provided int32 NCHW 1x1xhxw tensor
int snhost_parse(const int32_t* raw, int w, int h, float scale, float* depth_m, float* disp_px) {
  using namespace hobot::dnn_node;
  auto out = std::make_shared<DnnNodeOutput>();
  auto t = std::make_shared<DNNTensor>();
  t->sysMem[0].virAddr = const_cast<int32_t*>(raw);
  t->sysMem[0].memSize = (uint32_t)(4 * w * h);
  const int32_t dims[4] = {1, 1, h, w};
  for (int i = 0; i < 4; ++i) t->properties.validShape.dimensionSize[i] = dims[i];
  t->properties.validShape.numDimensions = 4;
  t->properties.tensorLayout = HB_DNN_LAYOUT_NCHW;
  t->properties.tensorType = HB_DNN_TENSOR_TYPE_S32;
  t->properties.scale.scaleLen = 1;
  t->properties.scale.scaleData = &scale;
  out->output_tensors.push_back(t);
  std::vector<std::shared_ptr<hobot::stereonet::StereonetResult>> res;
  if (hobot::stereonet::Parse(out, res) != 0 || res.empty()) return -1;
  memcpy(depth_m, res[0]->results.data(), sizeof(float) * w * h);
  memcpy(disp_px, res[0]->disparity.data(), sizeof(float) * w * h);
  return 0;
}

}  // extern "C"
