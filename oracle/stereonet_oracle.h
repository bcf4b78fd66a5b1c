/*
 * stereonet_oracle.h — CPU-float ORACLE for the StereoNet hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (hobot_stereonet_amd/csrc → libstereonet_hip.so) never links or calls it.
 *
 * PARITY STATUS
 *   - pre-processing / wire format / dequant: PINNED — restated from the
 *     reference's own integer code and checked against vectors produced by
 *     compiling the reference's pure functions (tests/golden/make_preprocess_golden.py).
 *   - the network (DnnNode::Run, stereonet_infer/src/stereonet_node.cpp:812):
 *     PARITY UNPINNED.  The arithmetic lives in Horizon's closed `dnn_node`
 *     (package.xml:16, no version pin; tros 2.0.1) → libdnn → the BPU binary
 *     hobot_stereonet.hbm, which is absent (.MISSING_LARGE_BLOBS:1) and has no
 *     test or golden vector in the reference.  This file restates the published
 *     StereoNet algorithm (Khamis et al., ECCV 2018) with K=4 / 12 planes as
 *     pinned by `16.0 * 12.0` at stereonet_infer/src/parser.cpp:86; the exact
 *     layer list ("SN-K4") is DESIGN.md §2.  It is cross-checked op-by-op
 *     against torch.nn.functional (tests/golden/make_network_golden.py).
 */
#ifndef STEREONET_ORACLE_H_
#define STEREONET_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SO_C 32            /* feature channels                              */
#define SO_NDOWN 4         /* 5x5 stride-2 convs  → 1/16 resolution (K=4)   */
#define SO_NFRES 6         /* feature-tower residual blocks                 */
#define SO_NAGG 4          /* 3-D aggregation convs (C→C)                   */
#define SO_NRRES 6         /* refinement residual blocks                    */
#define SO_MULTI_LEVELS 4  /* hierarchical refinement: towers at 1/8, 1/4, 1/2, 1  */
#define SO_LRELU 0.2f
#define SO_OUT_SCALE 2.60443857769133e-6f   /* stereonet_node.cpp:282, render.py:29 */

/* ---- reference host-side integer code, restated ------------------------ */

/* stereonet_node.cpp:705-738 — split a side-by-side 2w x h NV12 frame into
 * two contiguous w x h NV12 images (Y rows, then the h/2 chroma rows). */
void so_split_sbs_nv12(const uint8_t *sbs, int w, int h, uint8_t *left, uint8_t *right);

/* preprocess.h:128-155 (Tools::YUV420TOYUV444) — Y copied, chroma 2x2
 * replicated while indexing the source as PLANAR I420 (quirk B-1). */
void so_yuv420_to_yuv444(const uint8_t *in, uint8_t *out, int w, int h);

/* preprocess.cpp:1131-1136 (PreProcess::Quantize), defaults preprocess.h:236-240 */
int8_t so_quantize(float value, float scale, float zero_point, float mn, float mx);

/* preprocess.cpp:975-1056 — two NV12 eyes → int8 NCHW 1x6xhxw tensor */
void so_preprocess_nv12(const uint8_t *img_l, const uint8_t *img_r, int w, int h, int8_t *out6);

/* preprocess.h:56-96 (Tools::BGRToNv12, offline feeders only): cv::cvtColor(COLOR_BGR2YUV_I420) into a planar
 * I420 buffer, then U/V interleaved into NV12.  OpenCV is an un-vendored dependency of the reference
 * (stereonet_infer/CMakeLists.txt find_package(OpenCV)); its RGB->YUV420p arithmetic is restated from the
 * published source (imgproc color_yuv: BT.601 studio range, Q20 coefficients 269484/528482/102760,
 * -155188/-305135/460324, 460324/-385875/-74448, + half, >> 20, chroma from the top-left pixel of each 2x2).
 * PARITY UNPINNED against OpenCV itself (not in this image); pinned to the BT.601 known answers in
 * tests/test_filelist.py.  Returns -1 for odd w or h. */
int so_bgr_to_nv12(const uint8_t *bgr, int w, int h, uint8_t *nv12);

/* parser.cpp:79-94 / render.py:72-81 — dequant + metric depth */
void so_dequant_depth(const int32_t *raw, int n, float scale, float dmax,
                      float *disp_px, float *depth_m);

/* ---- network: weight blob layout --------------------------------------- */
/* Flat fp32 blob, PyTorch tensor layouts ([co][ci][kh][kw], [co][ci][kd][kh][kw]),
 * order (each layer = weight then bias):
 *   feat.down0..3  feat.res0..5.{1,2}  feat.out
 *   agg.conv0..3   agg.out
 *   ref.in         ref.res0..5.{1,2}   ref.out
 * a hierarchical ("multi") blob continues with the towers of the coarser levels, same layer list each:
 *   ref1.* (1/2 resolution)   ref2.* (1/4)   ref3.* (1/8)                       */
long so_weight_count(void);          /* = 423586 (single-scale network) */
long so_weight_count_levels(int levels);   /* 1 -> 423586, SO_MULTI_LEVELS -> 760933, else -1 */
long so_weight_offset(const char *name);   /* "<layer>.w|b", e.g. "feat.res3.2.w", "agg.out.b"; -1 if unknown */

/* ---- network: primitive ops (NCHW fp32, zero padding) ------------------ */
void so_conv2d(const float *in, int cin, int h, int w,
               const float *wt, const float *bias, int cout,
               int k, int stride, int pad, int dil, float *out);
void so_conv3d(const float *in, int cin, int d, int h, int w,
               const float *wt, const float *bias, int cout, float *out);  /* 3x3x3 pad 1 */
void so_lrelu(float *x, long n, float slope);
void so_cost_volume(const float *fl, const float *fr, int c, int dl, int h, int w, float *cv);
void so_soft_argmin(const float *cost, int dl, int h, int w, float *disp);
void so_upsample_bilinear(const float *in, int h, int w, int factor, float mul, float *out);

/* ---- network: stages ---------------------------------------------------- */
/* planes: 3 x hp x wp float (already int8/128); out: C x hp/16 x wp/16 */
void so_features(const float *weights, const float *planes, int hp, int wp, float *feat);
/* cost: dl x hl x wl (pre-softmax) */
void so_aggregate(const float *weights, const float *fl, const float *fr,
                  int dl, int hl, int wl, float *cost);
/* disp_up: hp x wp (px); img: 3 x hp x wp; out disparity hp x wp (px) */
void so_refine(const float *weights, const float *disp_up, const float *img,
               int hp, int wp, int dmax, float *disp);

/* Whole path: int8 NCHW 1x6xhxw → float disparity (h x w, px) and wire int32.
 * Any of disp/raw/disp_low may be NULL.  disp_low: (hp/16)x(wp/16) low-res
 * soft-argmin output (units of low-res px). Returns 0, or -1 on bad args. */
int so_forward(const float *weights, const int8_t *in6, int w, int h, int dmax,
               float *disp, int32_t *raw, float *disp_low);

/* Hierarchical refinement pieces: 2x2 average pooling (image pyramid), one refinement level with the tower named
 * ref<level> (0 = "ref") normalised by dnorm = D / 2^level, and the whole path with `levels` refinement levels
 * (1 = so_forward; SO_MULTI_LEVELS = `multi`, weights = the longer blob).  level_maps: NULL, or levels-1 pointers;
 * level_maps[k-1] receives the level-k map (hp/2^k x wp/2^k, level-k pixel units), k = 1 .. levels-1. */
void so_avgpool2(const float *in, int c, int h, int w, float *out);
void so_refine_level(const float *weights, int level, const float *disp_up, const float *img,
                     int hp, int wp, float dnorm, float *disp);
int so_forward_levels(const float *weights, int levels, const int8_t *in6, int w, int h, int dmax,
                      float *disp, int32_t *raw, float *disp_low, float *const *level_maps);

int so_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
