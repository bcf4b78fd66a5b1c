// sn_kernels.hpp — gfx950 (CDNA4) device kernels for the StereoNet path.
//
// Everything here replaces what the reference runs inside the opaque BPU blob behind
// DnnNode::Run (stereonet_infer/src/stereonet_node.cpp:812) plus the byte shuffling of
// PreProcess::CvtNV12Data2Tensors (stereonet_infer/src/preprocess.cpp:913-1059).
// Layer semantics: DESIGN.md §2 (SN-K4).  wave = 64 lanes; MFMA = v_mfma_f32_32x32x2_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#ifndef SN_X3S_FPK
#define SN_X3S_FPK 2
#endif
#ifndef SN_X3S_WAGPR
#define SN_X3S_WAGPR 1
#endif

namespace sn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int kC = 32;            // feature channels == MFMA N
constexpr int kRefPad = 8;        // zero border (px) of the fp16 refinement tensors = max dilation
constexpr float kSplitScale = 2048.0f;         // F16X3: value = hi + lo * 2^-11
constexpr float kSplitInv = 1.0f / 2048.0f;
constexpr float kSlope = 0.2f;    // LeakyReLU

// Bijective XCD-aware remap: hardware places block b on XCD b % 8; give each XCD a contiguous
// run of logical tiles so neighbouring tiles (which share halos) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Workgroup barrier for persistent kernels whose waves have global stores / prefetch loads in flight:
// __syncthreads() also drains vmcnt, i.e. every tile would wait for its predecessor's stores to be acknowledged.
// This one only retires the wave's own LDS traffic (lgkmcnt) before the s_barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// K0: NV12 pair -> int8 NCHW 6xHxW, bit-exact with preprocess.cpp:975-1056.
// The reference indexes the chroma as planar I420 (preprocess.h:131-133): "U" = in + w*h,
// "V" = U + w*h/4, sample (i/2)*w/2 + j/2; each byte then goes through Quantize(((b-128)/128))
// which equals b ^ 0x80 for all 256 values (tests/golden/preprocess_golden.npz).
// `src_pitch`/`eye_off` let the same kernel read the two eyes straight out of the 2W-wide
// side-by-side message (stereonet_node.cpp:705-738) without the host-side split.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pre_nv12(const uint8_t* __restrict__ left,
                                                  const uint8_t* __restrict__ right, int src_pitch,
                                                  int w, int h, int8_t* __restrict__ out6) {
  // one thread = 4 consecutive output bytes of one plane row
  const int quads = w >> 2;
  const long total = 6L * h * quads;
  for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += (long)gridDim.x * 256L) {
    const int q = (int)(t % quads);
    const long rowid = t / quads;
    const int i = (int)(rowid % h);
    const int plane = (int)(rowid / h);   // 0..5
    const uint8_t* src = plane < 3 ? left : right;
    const int p = plane % 3;
    const int j = q << 2;
    uint32_t v;
    if (p == 0) {
      // w % 4 == 0 and 4-byte aligned eye pointers are checked on the host
      v = *reinterpret_cast<const uint32_t*>(src + (size_t)i * src_pitch + j);
    } else {
      // linear index into the eye's contiguous w*h/2 chroma bytes, as the reference computes it
      const int base = (p == 1 ? 0 : (w * h) / 4) + (i / 2) * w / 2 + j / 2;
      // chroma byte k of the contiguous eye image lives at row h + k / w, col k % w of the source
      const int k0 = base, k1 = base + 1;
      const uint8_t c0 = src[(size_t)(h + k0 / w) * src_pitch + (k0 % w)];
      const uint8_t c1 = src[(size_t)(h + k1 / w) * src_pitch + (k1 % w)];
      v = (uint32_t)c0 * 0x0101u | ((uint32_t)c1 * 0x0101u << 16);
    }
    v ^= 0x80808080u;
    *reinterpret_cast<uint32_t*>(out6 + ((size_t)plane * h + i) * w + j) = v;
  }
}

// ------------------------------------------------------------------------------------------
// Input loaders for the implicit-GEMM convolution.  A loader maps (image, virtual channel,
// y, x) -> fp32 input value, returning 0 outside the tensor (zero padding) — this is where the
// cost volume, the 3-D plane gathering, the int8 dequantisation and the bilinear upsample are
// fused into the consuming convolution instead of being materialised in HBM.
// ------------------------------------------------------------------------------------------
struct LoadF32 {            // plain NCHW fp32 tensor [nimg][C][H][W]
  const float* p;
  int C, H, W;
  __device__ __forceinline__ float operator()(int img, int c, int y, int x) const {
    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W || c >= C) return 0.f;
    return p[(((unsigned)img * C + c) * H + y) * W + x];      // 32-bit index: tensors are < 2^32 elements (host check)
  }
};

struct LoadI8Eye {          // model input int8 [n][6][H][W]; image = n*2 + eye; 3 real channels
  const int8_t* p;
  int H, W;
  __device__ __forceinline__ float operator()(int img, int c, int y, int x) const {
    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W || c >= 3) return 0.f;
    const int n = img >> 1, eye = img & 1;
    return (float)p[(((unsigned)n * 6 + eye * 3 + c) * H + y) * W + x] * (1.0f / 128.0f);
  }
};

// 3-D conv as a 2-D conv over 96 virtual channels: c' = dz*32 + ci reads plane d+dz-1.
// Volume layout [n][Dl][32][H][W] (plane-major: every (n,d) is an ordinary NCHW image).
struct LoadVol3D {
  const float* p;
  int Dl, H, W;
  __device__ __forceinline__ float operator()(int img, int c, int y, int x) const {
    const int n = img / Dl, d = img - n * Dl;
    const int dz = c >> 5, ci = c & 31;
    const int dd = d + dz - 1;
    if ((unsigned)dd >= (unsigned)Dl || (unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W)
      return 0.f;
    return p[((((unsigned)n * Dl + dd) * kC + ci) * H + y) * W + x];
  }
};

// Same, but the volume is the cost volume computed on the fly from the two feature maps:
// cv[ci][dd][y][x] = fL[ci][y][x] - fR[ci][y][x-dd], 0 where x-dd < 0.   feat: [n*2+eye][32][H][W]
struct LoadCostVol {
  const float* feat;
  int Dl, H, W;
  __device__ __forceinline__ float operator()(int img, int c, int y, int x) const {
    const int n = img / Dl, d = img - n * Dl;
    const int dz = c >> 5, ci = c & 31;
    const int dd = d + dz - 1;
    if ((unsigned)dd >= (unsigned)Dl || (unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W ||
        x < dd)
      return 0.f;
    const unsigned plane = (unsigned)H * W;
    const unsigned li = ((unsigned)(2 * n) * kC + ci) * plane + (unsigned)y * W + x;
    return feat[li] - feat[li + kC * plane - dd];
  }
};

// Split-slot activations of the low-resolution branch (fp16 modes): [image][channel block (4)][hi | lo][H][W]
// 16-byte slots of 8 fp16 channels, v = hi + lo / 2048 (the operand format of the split-operand MFMAs, produced by
// the previous layer's epilogue instead of being re-derived from fp32 in every consumer's staging loop).
// Same bytes per element as fp32 NCHW, but a halo row of a tile is ONE contiguous run per (block, part) instead
// of eight 4-byte-strided plane reads, and staging is a plain 16-byte copy.  Dl > 0: the tensor is a volume
// [n][Dl] of such images and the virtual block vb = dz * 4 + cb reads plane d + dz - 1 (3-D convolution).
struct SlotIn {
  const uint4* p;
  int Dl, H, W;
  // k_conv_x3s staging: slot offset of virtual block vb relative to block 0 of image img, and its validity
  __device__ __forceinline__ int plane_off(int vb) const {
    const int hw2 = 2 * H * W;
    return Dl > 0 ? (((vb >> 2) - 1) * 4 + (vb & 3)) * hw2 : vb * hw2;
  }
  __device__ __forceinline__ bool plane_valid(int img, int vb) const {
    if (Dl <= 0) return true;
    const int d = img % Dl;
    return (unsigned)(d + (vb >> 2) - 1) < (unsigned)Dl;
  }
};
__device__ __forceinline__ size_t low_slot_index(int img, int cb, int part, int y, int x, int H, int W) {
  return ((((size_t)img * 4 + cb) * 2 + part) * H + y) * (size_t)W + x;
}

// Cost volume in split-slot form for the first 3-D conv of the slot pipeline:
//   vol[n][d][c][y][x] = x >= d ? fL[n][c][y][x] - fR[n][c][y][x - d] : 0      (LoadCostVol, materialised)
// feat fp32 [2n + eye][32][H][W]  ->  slots [n * Dl + d][4 blocks][hi | lo][H][W].  5.5 MB per pair at 1280x720:
// materialising it costs less than building it element-wise inside the conv's staging loop (two differently
// aligned scalar reads per element) — that loader made the first aggregation layer 2.6x slower than the others.
__global__ __launch_bounds__(256) void k_cost_slots(const float* __restrict__ feat, uint4* __restrict__ vol, int Dl,
                                                    int H, int W, int npairs) {
  const int plane = H * W;
  const long total = (long)npairs * Dl * 4 * plane;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int pix = (int)(i % plane);
  long t = i / plane;
  const int cb = (int)(t % 4);
  t /= 4;
  const int d = (int)(t % Dl), n = (int)(t / Dl);
  const int y = pix / W, x = pix - y * W;
  half8 hi, lo;
  const float* fl = feat + ((size_t)(2 * n) * kC + cb * 8) * plane + pix;
  const float* fr = fl + (size_t)kC * plane - d;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = x >= d ? fl[(size_t)k * plane] - fr[(size_t)k * plane] : 0.f;
    hi[k] = (_Float16)v;
    lo[k] = (_Float16)((v - (float)hi[k]) * kSplitScale);
  }
  uint4* o = vol + low_slot_index(n * Dl + d, cb, 0, y, x, H, W);
  o[0] = *reinterpret_cast<const uint4*>(&hi);
  o[plane] = *reinterpret_cast<const uint4*>(&lo);
}

// Scale of the bilinear upsample that feeds a refinement level: single-scale refinement reads the soft-argmin map
// x16 ({1/16, 16}); every level of the hierarchical refinement reads the map of the level below x2 ({1/2, 2}).
struct UpScale {
  float rs, mul;            // 1 / factor, factor (the values are disparities in pixels of the finer grid)
};

// Refinement statistic (sn_get_refine_stats, SN_PREC_AUTO): every head form adds the |D r| of the pixels it writes — what the
// refinement level moves the map by, in the level's pixels.  Per-lane partial sums meet in a wave reduction and leave as ONE
// 64-bit fixed-point atomic per wave (units of 2^-20 px): integer addition does not depend on the order the waves arrive in,
// so the statistic is identical from run to run, and nothing of it feeds back into the maps (outputs bit-unchanged).
// (|dmax * acc| and not |d - up|: `up` is itself a product (map value x factor) and `up + dmax * acc` contracts into ONE fma
// with whichever product has fewer uses; a second use of `up` flipped that choice and the last bit of d — seen as a
// difference against the round-5 library wherever dmax is not a power of two, profiles/r06_ab_outputs.txt)
// Where the sums meet: atomics on ONE address are resolved one after the other at about 10 ns each on this part (measured:
// the fp32 head with 4368 per-wave atomics per 1280x720 map ran 82.7 us, 41.3 us without them; profiles/r06_stat_atomics.txt),
// so a statistic word is kStatSlots partial sums in separate 128-byte lines (slot = workgroup index mod kStatSlots; the host
// adds them up — still integers, still order-independent), and the kernels that can meet in LDS first commit once per
// workgroup (refine_stat_commit_block).
constexpr float kStatScale = 1048576.0f;
constexpr int kStatSlots = 16, kStatLine = 16;              // partial sums per word; 64-bit words per 128-byte line
constexpr int kStatWordStride = kStatSlots * kStatLine;     // 64-bit words between two statistic words (levels)
__device__ __forceinline__ void refine_stat_commit(unsigned long long* stat, float lane_sum) {
#pragma unroll
  for (int off = 32; off; off >>= 1) lane_sum += __shfl_xor(lane_sum, off);
  if (stat != nullptr && (threadIdx.x & 63) == 0)
    atomicAdd(stat + (blockIdx.x % kStatSlots) * kStatLine, (unsigned long long)(lane_sum * kStatScale + 0.5f));
}
// every thread of a (<= 512-thread) workgroup calls this: one atomic per workgroup
__device__ __forceinline__ void refine_stat_commit_block(unsigned long long* stat, float lane_sum) {
  __shared__ float s_stat[8];
#pragma unroll
  for (int off = 32; off; off >>= 1) lane_sum += __shfl_xor(lane_sum, off);
  if ((threadIdx.x & 63) == 0) s_stat[threadIdx.x >> 6] = lane_sum;
  __syncthreads();
  if (stat != nullptr && threadIdx.x == 0) {
    float sum = 0.f;
    for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) sum += s_stat[w];
    const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    atomicAdd(stat + (wg % kStatSlots) * kStatLine, (unsigned long long)(sum * kStatScale + 0.5f));
  }
}

// bilinear upsample, align_corners=False (half-pixel centres, edge clamp), values x factor
__device__ __forceinline__ float upsample_map(const float* low, int hl, int wl, int y, int x, UpScale u) {
  float sy = ((float)y + 0.5f) * u.rs - 0.5f;
  float sx = ((float)x + 0.5f) * u.rs - 0.5f;
  sy = sy < 0.f ? 0.f : sy;
  sx = sx < 0.f ? 0.f : sx;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 < hl - 1 ? y0 + 1 : y0, x1 = x0 < wl - 1 ? x0 + 1 : x0;
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  const float v = hy * (hx * low[y0 * wl + x0] + lx * low[y0 * wl + x1]) +
                  ly * (hx * low[y1 * wl + x0] + lx * low[y1 * wl + x1]);
  return v * u.mul;
}

// Refinement input: channel 0 = upsampled disparity / D, channels 1..3 = left image planes — the int8 model input
// (full resolution) or, for the coarser levels of the hierarchical refinement, the float image pyramid.
struct LoadRefineIn {
  const float* disp_low;    // [n][hl][wl]
  const int8_t* in6;        // [n][6][H][W]
  int hl, wl, H, W, Hp, Wp;
  float inv_d;
  UpScale up;
  const float* pyr;         // nullptr, or [n][3][Hp][Wp] average-pooled left planes of this level
  __device__ __forceinline__ float operator()(int img, int c, int y, int x) const {
    if ((unsigned)y >= (unsigned)Hp || (unsigned)x >= (unsigned)Wp || c >= 4) return 0.f;
    if (c == 0) return upsample_map(disp_low + (size_t)img * hl * wl, hl, wl, y, x, up) * inv_d;
    if (pyr) return pyr[(((size_t)img * 3 + (c - 1)) * Hp + y) * Wp + x];
    if (y >= H || x >= W) return 0.f;
    return (float)in6[(((size_t)img * 6 + (c - 1)) * H + y) * W + x] * (1.0f / 128.0f);
  }
};

// Image pyramid of the hierarchical refinement: 2x2 average pooling of the LEFT eye's three planes.
// FIRST = true: source = int8 model input [n][6][H][W] (v / 128, zero outside H x W), output [n][3][Ho][Wo] with
// Ho = Hp / 2; otherwise source = the float level above, [n][3][2 Ho][2 Wo].  All values are small integers over a
// power of two, so the sums are exact in fp32 and the order of the additions does not matter.
template <bool FIRST>
__global__ __launch_bounds__(256) void k_img_pool2(const void* __restrict__ src, int H, int W, int Ho, int Wo,
                                                   float* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % Wo);
  long t = i / Wo;
  const int y = (int)(t % Ho);
  t /= Ho;
  const int c = (int)(t % 3), n = (int)(t / 3);
  float s = 0.f;
  if (FIRST) {
    const int8_t* p = reinterpret_cast<const int8_t*>(src) + ((size_t)n * 6 + c) * H * (size_t)W;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int yy = 2 * y + dy, xx = 2 * x + dx;
        if (yy < H && xx < W) s += (float)p[(size_t)yy * W + xx];
      }
    s *= 1.0f / 512.0f;
  } else {
    const float* p = reinterpret_cast<const float*>(src) + (((size_t)n * 3 + c) * (2 * Ho) + 2 * y) * (size_t)(2 * Wo) + 2 * x;
    s = ((p[0] + p[1]) + (p[2 * Wo] + p[2 * Wo + 1])) * 0.25f;
  }
  out[i] = s;
}

// ------------------------------------------------------------------------------------------
// Implicit-GEMM convolution, C_out = 32, on the exact-fp32 matrix core.
//   D[cout][pixel] += A[cout][k] * B[k][pixel],  k = (virtual channel, tap)
//   v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
//   lane l receives D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31], r = 0..15.
// so one wave owns a 32-pixel row segment x all 32 output channels, and for a fixed register r the
// two half-waves store two 128-byte runs of consecutive pixels (coalesced NCHW stores).
// Block = 4 waves, tile = TR rows x TC cols of output; input halo tile and the weight slice of CH
// input channels are staged through LDS per chunk.  For STRIDE 2 the staged columns are split by
// parity so that the 32 lanes of a half-wave read consecutive LDS words (no bank conflicts).
// ------------------------------------------------------------------------------------------
struct ConvArgs {
  const float* wpk;    // packed weights [cin_pad][taps][32]
  const float* bias;   // [32]
  float* out;          // [nimg][32][Ho][Wo]
  const float* res;    // nullable residual, same layout as out (may alias out)
  int nimg, cin_pad, Ho, Wo;
  int dil, pad;
  int lrelu;
  int tiles_x, tiles_y;
};

template <int KS, int STRIDE, int DIL, int CH, int TR, int TC, class Loader, bool PF = true, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void k_conv_c32_mfma(ConvArgs a, Loader ld) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TAPS = KS * KS;
  constexpr int CSEG = TC / 32;
  constexpr int NSEG = TR * CSEG;
  static_assert(NSEG % 4 == 0, "tile must split evenly over 4 waves");
  static_assert(STRIDE == 1 || DIL == 1, "strided convs are not dilated");
  constexpr int SPW = NSEG / 4;
  constexpr int ROWS_IN = (TR - 1) * STRIDE + (KS - 1) * DIL + 1;
  constexpr int COLS_IN = (TC - 1) * STRIDE + (KS - 1) * DIL + 1;
  constexpr int HALF = (COLS_IN + 1) / 2;
  constexpr int PITCH = STRIDE == 1 ? COLS_IN : 2 * HALF;
  constexpr int NELEM = CH * ROWS_IN * COLS_IN;          // input elements staged per chunk
  constexpr int EPT = (NELEM + 255) / 256;               // ... per thread
  constexpr int NW4 = CH * TAPS * 8;                     // float4s of weights per chunk
  constexpr int WPT = (NW4 + 255) / 256;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = a.tiles_x * a.tiles_y * a.nimg;
  const int b = xcd_remap(blockIdx.x, nwg);
  const int tx = b % a.tiles_x;
  const int t2 = b / a.tiles_x;
  const int ty = t2 % a.tiles_y;
  const int img = t2 / a.tiles_y;

  float* s_w = smem;                       // [CH][TAPS][32]
  float* s_in = smem + CH * TAPS * 32;     // [CH][ROWS_IN][PITCH]
  const int iy0 = ty * TR * STRIDE - a.pad, ix0 = tx * TC * STRIDE - a.pad;

  f32x16 acc[SPW];
#pragma unroll
  for (int s = 0; s < SPW; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

  const int kh = lane >> 5, j = lane & 31;

  // Register staging (issue early / write late): all global loads of the NEXT chunk are issued
  // back-to-back before the MFMAs of the current chunk and written to LDS after them, so HBM/L2
  // latency is paid once per chunk and hidden under the matrix work.
  float pre[EPT];
  float4 wpre[WPT];
  auto fetch = [&](int c0) {
    // opaque copy of the thread id: keeps hipcc from hoisting all EPT address computations out of the
    // chunk loop (that cost ~2 VGPRs per staged element and halved the occupancy)
    int tq = tid;
    asm volatile("" : "+v"(tq));
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int idx = e * 256 + tq;
      const int c = idx / (ROWS_IN * COLS_IN);
      const int rem = idx - c * (ROWS_IN * COLS_IN);
      const int r = rem / COLS_IN;
      const int cc = rem - r * COLS_IN;
      pre[e] = idx < NELEM ? ld(img, c0 + c, iy0 + r, ix0 + cc) : 0.f;
    }
    const float4* wsrc = reinterpret_cast<const float4*>(a.wpk + (size_t)c0 * TAPS * 32);
#pragma unroll
    for (int e = 0; e < WPT; ++e) {
      const int idx = e * 256 + tq;
      wpre[e] = idx < NW4 ? wsrc[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&]() {
    int tq = tid;
    asm volatile("" : "+v"(tq));
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int idx = e * 256 + tq;
      const int c = idx / (ROWS_IN * COLS_IN);
      const int rem = idx - c * (ROWS_IN * COLS_IN);
      const int r = rem / COLS_IN;
      const int cc = rem - r * COLS_IN;
      const int di = STRIDE == 1 ? cc : (cc & 1) * HALF + (cc >> 1);
      if (idx < NELEM) s_in[(c * ROWS_IN + r) * PITCH + di] = pre[e];
    }
#pragma unroll
    for (int e = 0; e < WPT; ++e) {
      const int idx = e * 256 + tq;
      if (idx < NW4) reinterpret_cast<float4*>(s_w)[idx] = wpre[e];
    }
  };

  fetch(0);
  commit();
  __syncthreads();

  for (int c0 = 0; c0 < a.cin_pad; c0 += CH) {
    const bool more = c0 + CH < a.cin_pad;
    if (PF && more) fetch(c0 + CH);     // PF: next chunk's loads fly under this chunk's MFMAs (more VGPRs)

#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
      const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
      for (int kk = 0; kk < CH; kk += 2) {
        const float wa = s_w[((kk + kh) * TAPS + tap) * 32 + j];
        const float* plane = s_in + (kk + kh) * ROWS_IN * PITCH;
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
          const int seg = wave * SPW + s;
          const int srow = seg / CSEG, scol = (seg - srow * CSEG) * 32;
          const int r = srow * STRIDE + ky * DIL;
          int di;
          if (STRIDE == 1) {
            di = scol + j + kx * DIL;
          } else {
            di = (kx & 1) * HALF + scol + j + (kx >> 1);
          }
          const float xb = plane[r * PITCH + di];
          acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, xb, acc[s], 0, 0, 0);
        }
      }
    }
    __syncthreads();     // everyone is done reading this chunk
    if (more) {
      if (!PF) fetch(c0 + CH);          // !PF: staging registers are dead during the MFMAs -> higher occupancy,
      commit();                         //      other resident blocks keep the matrix pipe busy meanwhile
      __syncthreads();
    }
  }

  // epilogue: bias (+ residual) (+ LeakyReLU), coalesced NCHW stores
  const size_t plane_o = (size_t)a.Ho * a.Wo;
#pragma unroll
  for (int s = 0; s < SPW; ++s) {
    const int seg = wave * SPW + s;
    const int srow = seg / CSEG, scol = (seg - srow * CSEG) * 32;
    const int y = ty * TR + srow, x = tx * TC + scol + j;
    if (y < a.Ho && x < a.Wo) {
      const size_t base = (size_t)img * kC * plane_o + (size_t)y * a.Wo + x;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * kh;
        float v = acc[s][r] + a.bias[co];
        const size_t idx = base + (size_t)co * plane_o;
        if (a.res) v += a.res[idx];
        if (a.lrelu) v = v > 0.f ? v : v * kSlope;
        a.out[idx] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Weights-stationary split-operand implicit GEMM for the LOW-RESOLUTION layers of the fp16 modes: three fp16 MFMAs
// per product on hi/lo operand pairs (x * w ~ xh*wh + (xh*wl + xl*wh) * 2^-11, two fp32 accumulators).  A kernel
// that re-reads the A fragments of every tap for every 32-pixel segment (5x5: 51 KB of weights per wave and
// 16-channel chunk through L1/L2 for 75 MFMAs) is bound by that traffic.  Here
//   * the workgroup is persistent and each wave keeps ITS weights in registers for the whole launch: K is split
//     across the two wave pairs by virtual channel halves (waves 0,1: channels [0, VCH/2), waves 2,3: the rest),
//     so a wave holds TAPS * VCH/32 K-steps x (hi, lo) fragments (5x5 C=32: 200 VGPRs, 3x3x3: 216, 3x3: 72);
//   * the whole VCH-channel halo tile is staged once per tile (16-byte slot copies, one tile ahead in registers),
//     each wave pair works on its own channel blocks of it;
//   * the two K-halves of a segment are summed through LDS: a wave owns one of its pair-partner's two segments,
//     writes the partial of the other one, reads the partner's, and finishes (bias, residual, LeakyReLU, store).
// A segment is 32 output pixels = (32 / SEGW) rows x SEGW columns (SEGW = 16 tiles an 80-column map exactly).
// The input is a split-slot tensor (SlotIn supplies the virtual-channel mapping of the 3-D layers); the output is a
// split-slot tensor (OUTSLOT) or fp32 NCHW (the feature map and the last aggregation volume).
// ------------------------------------------------------------------------------------------
// Exact unsigned division by a loop-invariant divisor: q = umulhi(t, floor((2^32 - 1) / d)) is floor(t / d) or one
// less for every t < 2^32 (t m / 2^32 > t / d - 1), so one conditional correction finishes it.
struct FastDiv {
  unsigned d, m;
  __device__ __forceinline__ explicit FastDiv(unsigned d_) : d(d_), m(0xFFFFFFFFu / d_) {}
  __device__ __forceinline__ unsigned divmod(unsigned t, unsigned& rem) const {
    unsigned q = __umulhi(t, m);
    unsigned r = t - q * d;
    const bool up = r >= d;
    q += up ? 1u : 0u;
    rem = up ? r - d : r;
    return q;
  }
};

template <int KS, int STRIDE, int VCH, int TR, int TC, int SEGW>
struct X3sTile {
  static constexpr int TAPS = KS * KS;
  static constexpr int NCB = VCH / 8, HCB = NCB / 2, NKC = HCB / 2, NK = TAPS * NKC;
  static constexpr int SEGH = 32 / SEGW;
  static constexpr int NSEG = (TR / SEGH) * (TC / SEGW), SPW = NSEG / 2;
  static constexpr int ROWS_IN = (TR - 1) * STRIDE + KS;
  static constexpr int COLS_IN = (TC - 1) * STRIDE + KS;
  // stride 2: columns are split by parity into two halves of a row.  The staging store (ds_write_b128, groups of 8
  // consecutive lanes, bank = dword address mod 32) alternates between the halves, so the second half must start
  // 4 slots (mod 8) after the first for the eight slots of a group to cover the 32 banks once: HALF = 4 (mod 8)
  // (34 -> 36 for the 5x5 tile; PMC: 22 % of that kernel's LDS cycles were bank conflicts).
  static constexpr int HALF0 = (COLS_IN + 1) / 2;
  static constexpr int HALF = STRIDE == 1 ? HALF0 : HALF0 + ((4 - HALF0 % 8) + 8) % 8;
  static constexpr int PITCH = STRIDE == 1 ? COLS_IN : 2 * HALF;
  static constexpr int PLANE = ROWS_IN * PITCH;
  static constexpr int NSLOT = NCB * ROWS_IN * COLS_IN;
  static constexpr int SPT = (NSLOT + 255) / 256;
  static constexpr int RED_FLOATS = 4 * 16 * 64;           // one segment partial per wave
  static constexpr size_t LDS_BYTES = (size_t)2 * NCB * PLANE * 16 + (size_t)RED_FLOATS * 4 + 32 * 4 + 16;
  static_assert(VCH % 32 == 0 && SPW == 2, "two segments per wave pair member");
  static_assert(TR % SEGH == 0 && TC % SEGW == 0, "tile must be whole segments");
};

// Staging is a 16-byte copy per (block, part, pixel) of the SlotIn tensor;
// OUTSLOT: the epilogue writes (and reads the residual from) a split-slot tensor instead of fp32 NCHW.
// HASRES: a residual tensor (same format as the output) is added before the activation.
template <int KS, int STRIDE, int VCH, int TR, int TC, int SEGW, int MINB, bool OUTSLOT, bool HASRES, class Loader>
__global__ __launch_bounds__(256, MINB) void k_conv_x3s(ConvArgs a, Loader ld) {
  using T = X3sTile<KS, STRIDE, VCH, TR, TC, SEGW>;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  uint4* s_xh = smem4;
  uint4* s_xl = smem4 + T::NCB * T::PLANE;
  float* s_red = reinterpret_cast<float*>(smem4 + 2 * T::NCB * T::PLANE);
  float* s_bias = s_red + T::RED_FLOATS;        // 32 floats: read back per tile through lgkmcnt, not vmcnt

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = wave >> 1, pset = wave & 1;
  const int gh = lane >> 5, j = lane & 31;

  // this wave's A fragments: chunks [khalf * NKC, (khalf + 1) * NKC) of the host packing [chunk][tap][hi|lo][lane]
  half8 wh[T::NK], wl[T::NK];
  {
    const uint4* wsrc = reinterpret_cast<const uint4*>(a.wpk) + (size_t)khalf * T::NKC * T::TAPS * 2 * 64 + lane;
#pragma unroll
    for (int k = 0; k < T::NK; ++k) {
      const uint4 x = wsrc[(2 * k) * 64], y = wsrc[(2 * k + 1) * 64];
      wh[k] = *reinterpret_cast<const half8*>(&x);
      wl[k] = *reinterpret_cast<const half8*>(&y);
    }
#if SN_X3S_WAGPR
    // Home the weight fragments in the accumulator half of the (unified) register file when the kernel needs more
    // than 256 registers: an MFMA reads its A operand from an AGPR directly, whereas fragments that the allocator
    // SPILLS to AGPRs come back through four v_accvgpr_read each (one extra instruction per MFMA in an issue-bound loop)
    if (T::NK > 12) {
#pragma unroll
      for (int k = 0; k < T::NK; ++k) asm volatile("" : "+a"(wh[k]), "+a"(wl[k]));
    }
#endif
  }
  // Pixel of the segment that lane j computes: row pr, column pc.  A ds_read_b128 is served in groups of 16 lanes
  // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} per half-wave) that must hit 64 distinct banks = 16 distinct slots modulo
  // 16.  With one row of 32 pixels per segment the slots are consecutive: conflict-free.  With two rows of 16 (SEGW = 16)
  // the second row starts PITCH slots later (18 for a 16-column 3x3 tile): taken in plain order, lanes 20-27 land on
  // the banks of lanes 12-15 (PMC: 35-40 % of these kernels' LDS cycles were bank conflicts).  Rotating the columns of
  // row pr by -pr * PITCH makes the slot of lane j congruent to j modulo 16 again; the epilogue uses the same map.
  constexpr int ROT = (SEGW == 16 && STRIDE == 1) ? T::PITCH % 16 : 0;
  const int pr = j / SEGW, pc = (SEGW == 16) ? ((j % SEGW) - pr * ROT) & 15 : j % SEGW;
  // per-lane slot offset of each of the wave's two segments (pixel (pr, pc) of the segment, channel-block parity gh)
  // slot 0 = the segment this wave finishes (pset * 2 + khalf), slot 1 = the one whose partial it ships to its
  // pair partner (wave ^ 2: same pixels, other K half, the roles swapped)
  int lane_base[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int seg = pset * 2 + (s == 0 ? khalf : 1 - khalf);
    const int srow = (seg / (TC / SEGW)) * T::SEGH + pr, scol = (seg % (TC / SEGW)) * SEGW + pc;
    lane_base[s] = (khalf * T::HCB + gh) * T::PLANE + srow * STRIDE * T::PITCH + scol;
  }
  if (tid < kC) s_bias[tid] = a.bias[tid];
  const int total = a.tiles_x * a.tiles_y * a.nimg;
  // Staging: a unit is one 16-byte slot of (virtual block, part, row, column): load -> LDS, no VALU.
  static_assert(std::is_same<Loader, SlotIn>::value, "k_conv_x3s reads split-slot tensors");
  constexpr int NSL = 2 * T::NSLOT, LPT = (NSL + 255) / 256;       // slots per tile (hi and lo) / per thread
  constexpr int FPK = SN_X3S_FPK;                                   // staging copies issued per K-step
  float pre[LPT * 4];
  // Which slot a thread copies in round e never changes, so (block, part, row, column, LDS offset) are
  // decoded ONCE into one packed register per round; per tile a copy is then ~10 instructions of address math
  // (decoding idx -> coordinates per tile cost 9,000 of the 21,000 cycles a 5x5 tile took).
  //   bits 0..12 LDS slot offset (inside s_xh / s_xl), 13..19 column, 20..24 row, 25..28 virtual block, 29 part,
  //   30 valid
  unsigned stab[LPT];
  unsigned srel[LPT];       // byte offset of the copy relative to (image, block 0, hi, row iy0, column ix0), mod 2^32
  constexpr int DUMMY = 2 * T::NCB * T::PLANE + T::RED_FLOATS / 4 + 8;     // spare LDS slot behind the bias
  {
    static_assert(DUMMY < 8192 && T::COLS_IN < 128 && T::ROWS_IN < 32 && T::NCB <= 16, "packed staging table");
    static_assert(LPT <= 32, "one validity bit per staging round");
#pragma unroll
    for (int e = 0; e < LPT; ++e) {
      const int idx = e * 256 + tid;
      const int vp = idx / (T::ROWS_IN * T::COLS_IN);
      const int rem = idx - vp * (T::ROWS_IN * T::COLS_IN);
      const int r = rem / T::COLS_IN;
      const int cc = rem - r * T::COLS_IN;
      const int di = STRIDE == 1 ? cc : (cc & 1) * T::HALF + (cc >> 1);
      const unsigned off = (vp & 1) * (T::NCB * T::PLANE) + (vp >> 1) * T::PLANE + r * T::PITCH + di;
      const bool ok = idx < NSL;
      stab[e] = ok ? (off | (cc << 13) | (r << 20) | ((vp >> 1) << 25) | (1u << 30)) : (unsigned)DUMMY;
      srel[e] = ok ? (unsigned)(ld.plane_off(vp >> 1) + (vp & 1) * (ld.H * ld.W) + r * ld.W + cc) * 16u : 0u;
    }
  }
  unsigned vmask = 0;                   // bit e: the slot fetched in round e lies inside the tensor (else LDS gets 0)
  // Every global address of the staging is  tensor base (SGPR pair)  +  32-bit byte offset (one VGPR):
  //   offset = f_toff (uniform: image, row iy0, column ix0 of the tile, mod 2^32)  +  srel[e] (per lane, fixed)
  // (the host keeps every tensor below 4 GiB; the previous depth plane of a volume lies BEFORE the image base, which
  // the modular arithmetic handles).  The loop around the MFMAs is ISSUE bound — the first form of this copy cost 45
  // instructions (64-bit address math, four short-circuit branches), i.e. 4.5 non-matrix instructions per MFMA and
  // 64 cycles per MFMA instead of 32 — so the validity test is branch-free integer arithmetic, and tiles that lie
  // inside the image with all their depth planes present (f_fast, wave-uniform) skip it altogether.
  const char* const f_base = reinterpret_cast<const char*>(ld.p);
  const FastDiv div_tx((unsigned)a.tiles_x), div_ty((unsigned)a.tiles_y);    // tile index -> (image, row, column)
  int f_iy0 = 0, f_ix0 = 0;
  unsigned f_toff = 0, f_pmask = 0;
  bool f_fast = false;
  auto fetch_one = [&](int e) {         // copy number e of the tile prepared by fetch()
    unsigned t = stab[e];
    unsigned rel = srel[e];
    asm volatile("" : "+v"(t), "+v"(rel));      // keep the unpacking inside the tile loop (hoisted it spills)
    unsigned off = rel + f_toff;
    if (!f_fast) {                              // wave-uniform
      const unsigned cc = (t >> 13) & 127u, r = (t >> 20) & 31u, vb = (t >> 25) & 15u;
      const unsigned ok = (unsigned)((unsigned)(f_iy0 + (int)r) < (unsigned)ld.H) & (unsigned)((unsigned)(f_ix0 + (int)cc) < (unsigned)ld.W) &
                          ((f_pmask >> vb) & 1u) & (t >> 30);
      off = ok ? off : 0u;                      // a slot outside the tensor reads slot 0 and is zeroed at commit
      vmask |= ok << e;
    }
    typedef float f4v __attribute__((ext_vector_type(4)));
    *reinterpret_cast<f4v*>(&pre[e * 4]) = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(f_base + off));
  };
  auto fetch = [&](int tile) {
    unsigned txu, tyu;
    const unsigned t2 = div_tx.divmod((unsigned)tile, txu);
    const int img = (int)div_ty.divmod(t2, tyu);
    const int tx = (int)txu, ty = (int)tyu;
    const int iy0 = ty * TR * STRIDE - a.pad, ix0 = tx * TC * STRIDE - a.pad;
    {
      // branch-free: every lane loads; a slot outside the tensor (zero padding, missing depth plane) reads slot 0
      // of the image instead and is replaced by zeros when it is committed to LDS.  Only the per-tile constants
      // are set up here; the LPT copies themselves are issued from inside the MFMA loop (fetch_one), where their
      // address arithmetic runs in the shadow of the matrix pipe.
      f_iy0 = iy0;
      f_ix0 = ix0;
      f_toff = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)img * 8u * (unsigned)(ld.H * ld.W) +
                                                                 (unsigned)(iy0 * ld.W + ix0)) * 16u));
      f_pmask = 0;                      // valid virtual blocks of this image (3-D: neighbouring depth planes)
#pragma unroll
      for (int vb = 0; vb < T::NCB; ++vb) f_pmask |= ld.plane_valid(img, vb) ? (1u << vb) : 0u;
      // interior tile with every depth plane present: all NSL slots are valid (rounds past NSL copy slot 0 + srel 0
      // = a valid address whose value lands in the spare LDS slot)
      f_fast = iy0 >= 0 && iy0 + T::ROWS_IN <= ld.H && ix0 >= 0 && ix0 + T::COLS_IN <= ld.W &&
               f_pmask == (1u << T::NCB) - 1u;
      vmask = f_fast ? 0xffffffffu : 0u;
    }
  };
  auto commit = [&]() {
    {
#pragma unroll
      for (int e = 0; e < LPT; ++e) {
        unsigned t = stab[e];
        asm volatile("" : "+v"(t));
        uint4 v = *reinterpret_cast<const uint4*>(&pre[e * 4]);
        if (!((vmask >> e) & 1u)) v = make_uint4(0, 0, 0, 0);
        s_xh[t & 8191u] = v;            // rounds past the tile's last slot write a spare slot
      }
    }
  };

  // XCD-aware persistent schedule: workgroup b runs on XCD b % 8 (round-robin dispatch); each XCD walks its own
  // contiguous band of tiles so that neighbouring tiles (shared halo rows / columns) meet in ONE L2.
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
  const int t_end = (int)((long)(xcd + 1) * total / 8);
  int tile = (int)((long)xcd * total / 8) + lb;
  if (tile >= t_end) return;
  fetch(tile);
#pragma unroll
  for (int e = 0; e < LPT; ++e) fetch_one(e);
  commit();
  __syncthreads();
  const size_t plane_o = (size_t)a.Ho * a.Wo;
  for (; tile < t_end; tile += nlb) {
    const int nxt = tile + nlb;
    // residual of the segment this wave finishes: requested before the MFMAs, consumed in the epilogue
    unsigned e_txu, e_tyu;
    const unsigned e_t2 = div_tx.divmod((unsigned)tile, e_txu);
    const int e_img = (int)div_ty.divmod(e_t2, e_tyu);
    const int e_tx = (int)e_txu, e_ty = (int)e_tyu;
    const int e_seg = pset * 2 + khalf;
    const int e_y = e_ty * TR + (e_seg / (TC / SEGW)) * T::SEGH + pr;
    const int e_x = e_tx * TC + (e_seg % (TC / SEGW)) * SEGW + pc;
    const bool e_in = e_y < a.Ho && e_x < a.Wo;
    float rv[HASRES ? 16 : 1];
#pragma unroll
    for (int r = 0; r < (HASRES ? 16 : 1); ++r) rv[r] = 0.f;
    if (HASRES && e_in) {
      if (OUTSLOT) {
        const char* rs = reinterpret_cast<const char*>(a.res) + (size_t)e_img * 8 * plane_o * 16;     // uniform
        const unsigned lane_off = ((unsigned)e_y * (unsigned)a.Wo + (unsigned)e_x) * 16u + gh * 8u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const char* rq = rs + (size_t)(2 * q) * plane_o * 16;
          const half4 rh = *reinterpret_cast<const half4*>(rq + lane_off);
          const half4 rl = *reinterpret_cast<const half4*>(rq + plane_o * 16 + lane_off);
#pragma unroll
          for (int e = 0; e < 4; ++e) rv[4 * q + e] = (float)rh[e] + (float)rl[e] * kSplitInv;
        }
      } else {
        const size_t base = (size_t)e_img * kC * plane_o + (size_t)e_y * a.Wo + e_x;
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = a.res[base + (size_t)((r & 3) + 8 * (r >> 2) + 4 * gh) * plane_o];
      }
    }
    const int more = __builtin_amdgcn_readfirstlane(nxt < t_end ? 1 : 0);
    fetch(more ? nxt : tile);             // no next tile: the staging loads run once more on this one (no branch per K-step)

    f32x16 acc0[2], acc1[2];
    f32x16 zero;                        // C operand of the first K-step (an inline constant: no seeding moves)
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    // B operands are fetched one K-step ahead (this kernel runs one wave per SIMD: an LDS read issued right
    // before its MFMA would expose the full LDS latency 100 times per tile); the scheduling barriers keep hipcc
    // from sinking the reads back to their uses.
    auto koff_of = [&](int k) {
      const int kc = k / T::TAPS, tap = k - kc * T::TAPS;
      const int ky = tap / KS, kx = tap - ky * KS;
      return 2 * kc * T::PLANE + ky * T::PITCH + (STRIDE == 1 ? kx : (kx & 1) * T::HALF + (kx >> 1));
    };
    uint4 bh[2][2], bl[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bh[0][s] = s_xh[lane_base[s] + koff_of(0)];
      bl[0][s] = s_xl[lane_base[s] + koff_of(0)];
    }
#pragma unroll
    for (int k = 0; k < T::NK; ++k) {
      const int cur = k & 1, nx = cur ^ 1;
      if (k + 1 < T::NK) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bh[nx][s] = s_xh[lane_base[s] + koff_of(k + 1)];
          bl[nx][s] = s_xl[lane_base[s] + koff_of(k + 1)];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const half8 xh = *reinterpret_cast<const half8*>(&bh[cur][s]);
        const half8 xl = *reinterpret_cast<const half8*>(&bl[cur][s]);
        acc0[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xh, k == 0 ? zero : acc0[s], 0, 0, 0);
        acc1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xh, k == 0 ? zero : acc1[s], 0, 0, 0);
        acc1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xl, acc1[s], 0, 0, 0);
      }
      // staging copies of the next tile: FPK per K-step from the start of the loop, so that the last of them still
      // has most of the loop's MFMAs (not one K-step) between its issue and the commit that waits for it
#pragma unroll
      for (int f = 0; f < FPK; ++f)
        if (FPK * k + f < LPT) fetch_one(FPK * k + f);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = FPK * T::NK; e < LPT; ++e) fetch_one(e);     // (only when a tile has more copies than the loop takes)
    // ship the partial of the segment the pair partner finishes
    {
      float* dst = s_red + (size_t)wave * 16 * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = acc0[1][r] + acc1[1][r] * kSplitInv;
    }
    lds_barrier();                      // halo tile free, partials visible
    if (more) commit();


    {
      const float* src = s_red + (size_t)(wave ^ 2) * 16 * 64 + lane;      // partner: same pixel set, other K half
      if (e_in) {
        if (OUTSLOT) {
          // channel block q = r >> 2 holds couts 8q .. 8q+7; this lane owns 4gh .. 4gh+3 of it: one 8-byte store
          // into the hi slot and one into the lo slot (the two half-waves fill the 16-byte slot together)
          // address = uniform 64-bit base of (image, block q, hi) + the lane's pixel offset inside the plane
          char* const o = reinterpret_cast<char*>(a.out) + (size_t)e_img * 8 * plane_o * 16;
          const unsigned lane_off = ((unsigned)e_y * (unsigned)a.Wo + (unsigned)e_x) * 16u + gh * 8u;
          const float slope = a.lrelu ? kSlope : 1.0f;        // max(v, v) = v: one code path
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            char* oq = o + (size_t)(2 * q) * plane_o * 16;                          // uniform
            half4 hh, hl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * q + e;
              float v = acc0[0][r] + acc1[0][r] * kSplitInv + src[r * 64] + s_bias[8 * q + 4 * gh + e];
              if (HASRES) v += rv[HASRES ? r : 0];
              v = fmaxf(v, v * slope);
              const _Float16 hi = (_Float16)v;
              hh[e] = hi;
              hl[e] = (_Float16)((v - (float)hi) * kSplitScale);
            }
            // non-temporal (as the staging loads): the low-resolution branch streams ~1.3 GB per piece through the
            // memory system while the towers on the other stream live off the 256 MB Infinity Cache (+1.1 % end to end)
            __builtin_nontemporal_store(hh, reinterpret_cast<half4*>(oq + lane_off));
            __builtin_nontemporal_store(hl, reinterpret_cast<half4*>(oq + plane_o * 16 + lane_off));
          }
        } else {
          const size_t base = (size_t)e_img * kC * plane_o + (size_t)e_y * a.Wo + e_x;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * gh;
            float v = acc0[0][r] + acc1[0][r] * kSplitInv + src[r * 64] + s_bias[co];
            if (HASRES) v += rv[HASRES ? r : 0];
            if (a.lrelu) v = v > 0.f ? v : v * kSlope;
            a.out[base + (size_t)co * plane_o] = v;
          }
        }
      }
    }
    lds_barrier();                      // next tile staged; partial buffer free
  }
}

// ------------------------------------------------------------------------------------------
// K1a: first down-conv (3 -> 32, 5x5, stride 2) straight from the int8 model input, fp16 modes.
// With three input channels the generic implicit GEMM spends 50 exact-fp32 MFMAs (64 cycles each) per 32 output
// pixels on K = 4 (padded) x 25.  The int8 input / 128 is EXACT in fp16, so only the weights need the hi/lo split
// (two fp16 MFMAs per K step), and K is packed densely as  (ci, ky) rows x kx[0..7]:
//   k = 8 * rho + kx,  rho = ci * 5 + ky in 0..14 (row 15 and kx 5..7 carry zero weights)  ->  8 K-steps of 16
//   = 16 MFMAs of 32 cycles per 32 pixels (6.25x fewer matrix cycles).
// The B fragment of lane (pixel j, k-group g) at K-step t is 8 CONSECUTIVE input columns of row rho = 2t + g,
// starting at column 2*ox - 2: 16 bytes at a 4-byte aligned LDS address -> four ds_read_b32 (or two
// ds_read2_b32), no packing VALU.  The three extra columns read under the zero weights are real neighbouring
// pixels (finite), never uninitialised LDS.
// Persistent workgroups (static stride over 8 x TC output tiles), both weight fragment sets (16 x half8) in
// registers, LDS tile double buffered: the next tile's int8 dwords are in flight during the MFMAs.
// Output fp32 NCHW [2n][32][Ho][Wo], exactly what the generic kernel writes.
// ------------------------------------------------------------------------------------------
template <int TC>
struct Down0Tile {
  static constexpr int TR = 8;
  static constexpr int ROWS = 2 * TR + 3;            // input rows of a tile
  static constexpr int NDW = (2 * TC + 3 + 3 + 2 + 3) / 4;   // staged dwords per row: window starts 2 px left of ix0 (4-aligned)
  static constexpr int PITCH = NDW * 4 + 4;          // halves per LDS row (+4: keeps rows 8-byte aligned, staggers banks)
  static constexpr int BUF = 3 * ROWS * PITCH;       // halves per buffer
  static constexpr int LDS_BYTES = 2 * BUF * 2;
  static constexpr int NLOAD = (3 * ROWS * NDW + 255) / 256;
  static constexpr int CSEG = TC / 32, SPW = TR * CSEG / 4;
};

template <int TC>
__global__ __launch_bounds__(256, 2) void k_down0_f16(const int8_t* __restrict__ in6, int H, int W,
                                                   const uint4* __restrict__ wfrag,   // [8][hi|lo][64]
                                                   const float* __restrict__ bias, float* __restrict__ out, int Ho,
                                                   int Wo, int tiles_x, int tiles_y, int nimg, int lrelu,
                                                   int al4,     // W % 4 == 0 and in6 4-byte aligned: dword loads
                                                   int oPH, int oPW, int opy, int opx) {   // output slot grid and image origin (plain: Ho, Wo, 0, 0)
  using T = Down0Tile<TC>;
  extern __shared__ __attribute__((aligned(16))) _Float16 s_x[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, g = lane >> 5;

  half8 wh[8], wl[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const uint4 a = wfrag[(2 * t) * 64 + lane], b = wfrag[(2 * t + 1) * 64 + lane];
    wh[t] = *reinterpret_cast<const half8*>(&a);
    wl[t] = *reinterpret_cast<const half8*>(&b);
  }
  f32x16 bv;
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * g];

  // per-lane LDS offset (halves) of K-step t: row rho = 2t + g  ->  (ci, ky); rho = 15 re-reads row 0 (zero weights)
  int koff[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    int rho = 2 * t + g;
    rho = rho < 15 ? rho : 0;
    koff[t] = ((rho / 5) * T::ROWS + (rho % 5)) * T::PITCH;
  }

  const int total = tiles_x * tiles_y * nimg;
  uint32_t pre[T::NLOAD];
  auto fetch = [&](int tile) {
    const int tx = tile % tiles_x, t2 = tile / tiles_x;
    const int ty = t2 % tiles_y, img = t2 / tiles_y;
    const int n = img >> 1, eye = img & 1;
    const int iy0 = 2 * ty * T::TR - 2, xs = 2 * tx * TC - 4;          // staged window origin (xs % 4 == 0)
    int tq = tid;
    asm volatile("" : "+v"(tq));
#pragma unroll
    for (int e = 0; e < T::NLOAD; ++e) {
      const int idx = e * 256 + tq;
      const int c = idx / (T::ROWS * T::NDW);
      const int rem = idx - c * (T::ROWS * T::NDW);
      const int r = rem / T::NDW, q = rem - r * T::NDW;
      const int y = iy0 + r, x = xs + 4 * q;
      uint32_t v = 0;
      if (idx < 3 * T::ROWS * T::NDW && (unsigned)y < (unsigned)H) {
        const int8_t* row = in6 + (((size_t)n * 6 + eye * 3 + c) * H + y) * (size_t)W;
        if (al4) {
          if (x >= 0 && x + 3 < W) v = *reinterpret_cast<const uint32_t*>(row + x);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if ((unsigned)(x + k) < (unsigned)W) v |= (uint32_t)(uint8_t)row[x + k] << (8 * k);
        }
      }
      pre[e] = v;
    }
  };
  auto commit = [&](_Float16* buf) {
    int tq = tid;
    asm volatile("" : "+v"(tq));
#pragma unroll
    for (int e = 0; e < T::NLOAD; ++e) {
      const int idx = e * 256 + tq;
      const int c = idx / (T::ROWS * T::NDW);
      const int rem = idx - c * (T::ROWS * T::NDW);
      const int r = rem / T::NDW, q = rem - r * T::NDW;
      if (idx < 3 * T::ROWS * T::NDW) {
        half4 hv;
#pragma unroll
        for (int k = 0; k < 4; ++k) hv[k] = (_Float16)((float)(int8_t)(pre[e] >> (8 * k)) * (1.0f / 128.0f));
        *reinterpret_cast<half4*>(buf + (c * T::ROWS + r) * T::PITCH + 4 * q) = hv;
      }
    }
  };

  int tile = blockIdx.x;
  if (tile >= total) return;
  fetch(tile);
  commit(s_x);
  __syncthreads();
  int cur = 0;
  // outputs: uniform 64-bit address of (image, block q, part, tile origin) + a fixed 32-bit offset per lane
  unsigned io_voff[T::SPW];
#pragma unroll
  for (int s = 0; s < T::SPW; ++s) {
    const int seg = wave * T::SPW + s;
    io_voff[s] = ((unsigned)(seg / T::CSEG) * (unsigned)oPW + (unsigned)((seg % T::CSEG) * 32 + j)) * 16u + g * 8u;
  }
  const size_t plane_b = (size_t)oPH * oPW * 16;               // bytes of one (block, part) plane
  const float slope = lrelu ? kSlope : 1.0f;                  // max(v, v) = v: one code path
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  for (; tile < total; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    if (nxt < total) fetch(nxt);
    const _Float16* buf = s_x + cur * T::BUF;
    f32x16 acc0[T::SPW], acc1[T::SPW];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int s = 0; s < T::SPW; ++s) {
        const int seg = wave * T::SPW + s;
        const int srow = seg / T::CSEG, scol = (seg % T::CSEG) * 32;
        // input row 2*srow + ky, first column 2*(scol + j) + 2 of the staged window (= ix0 + 2*ox_local)
        const uint32_t* px = reinterpret_cast<const uint32_t*>(buf + koff[t] + 2 * srow * T::PITCH + 2 * (scol + j) + 2);
        uint4 xv;
        xv.x = px[0];
        xv.y = px[1];
        xv.z = px[2];
        xv.w = px[3];
        const half8 xb = *reinterpret_cast<const half8*>(&xv);
        // first K-step: accumulators start at the bias / at zero through the C operand (no seeding moves)
        acc0[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], xb, t == 0 ? bv : acc0[s], 0, 0, 0);
        acc1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t], xb, t == 0 ? zero : acc1[s], 0, 0, 0);
      }
    }
    {
      const int tx = tile % tiles_x, t2 = tile / tiles_x;
      const int ty = t2 % tiles_y, img = t2 / tiles_y;
      const int y0 = ty * T::TR, x0 = tx * TC;
      const bool interior = y0 + T::TR <= Ho && x0 + TC <= Wo;             // wave-uniform
      // split-slot tensor for the next down-conv (see SlotIn / low_slot_index): [img][4 blocks][hi | lo][Ho][Wo]
      char* const tbase = reinterpret_cast<char*>(out) + (size_t)img * 8 * plane_b + ((size_t)(y0 + opy) * oPW + x0 + opx) * 16;
#pragma unroll
      for (int s = 0; s < T::SPW; ++s) {
        bool ok = true;
        if (!interior) {
          const int seg = wave * T::SPW + s;
          ok = y0 + seg / T::CSEG < Ho && x0 + (seg % T::CSEG) * 32 + j < Wo;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          char* oq = tbase + (size_t)(2 * q) * plane_b;                    // uniform
          half4 hh, hl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc0[s][4 * q + e] + acc1[s][4 * q + e] * kSplitInv;
            v = fmaxf(v, v * slope);
            const _Float16 hi = (_Float16)v;
            hh[e] = hi;
            hl[e] = (_Float16)((v - (float)hi) * kSplitScale);
          }
          if (ok) {       // non-temporal: 944 MB per 16-pair piece that only the next down-conv reads, once
            __builtin_nontemporal_store(hh, reinterpret_cast<half4*>(oq + io_voff[s]));
            __builtin_nontemporal_store(hl, reinterpret_cast<half4*>(oq + plane_b + io_voff[s]));
          }
        }
      }
    }
    if (nxt < total) commit(s_x + (cur ^ 1) * T::BUF);
    lds_barrier();
    cur ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// K6: final 3x3x3 conv 32->1 fused with soft-argmin.
//   cost[d] = b + sum_{ci,dz,ky,kx} w[ci][dz][ky][kx] * vol[n][d+dz-1][ci][y+ky-1][x+kx-1]
//   disp    = sum_d d * softmax_d(-cost)
// Workgroup = 16 waves x 64 consecutive pixels: wave g accumulates the partial costs of all Dl planes over its 2
// input channels (every load is a 256-byte run of one channel plane, weights are wave-uniform scalar loads), the 16
// partials meet in LDS and wave 0 finishes (softmax / expectation in registers).  The kernel is latency bound
// (~37 MFLOP per pair), so what matters is waves in flight and loads in flight per wave: the plane loop is fully
// unrolled (cost[] stays in registers) but fenced every second plane, which keeps 18 loads per lane in flight and the
// register count near 64 — unfenced, hipcc hoisted all 16 x 9 loads of a channel and spilled 300 B per lane
// (68 MB of scratch writes per launch against 1.4 MB of output).
// ------------------------------------------------------------------------------------------
constexpr int kSamWaves = 16;
template <int DLMAX>
__global__ __launch_bounds__(64 * kSamWaves) void k_head_softargmin(const float* __restrict__ vol,   // [n][Dl][32][H][W]
                                                                    const float* __restrict__ w,     // [32][27] (ci, dz*9+ky*3+kx)
                                                                    float bias, int Dl, int H, int W, int npix_total,
                                                                    float* __restrict__ disp_low,    // [n][H][W]
                                                                    float* __restrict__ cost_out) {  // nullable [n][Dl][H][W]
  __shared__ float s_part[kSamWaves][DLMAX][64];
  constexpr int CPW = kC / kSamWaves;           // channels per wave
  const int lane = threadIdx.x & 63;
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // channel group
  const int gp = blockIdx.x * 64 + lane;        // global pixel index over n*H*W
  const int plane = H * W;
  const bool live = gp < npix_total;
  const int n = live ? gp / plane : 0;
  const int pix = live ? gp - n * plane : 0;
  const int y = pix / W, x = pix - y * W;
  // the nine taps of this pixel: element offset inside a channel plane, or -1 outside the image (zero padding)
  int toff[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xx = x + kx - 1;
      toff[ky * 3 + kx] = (live && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? yy * W + xx : -1;
    }

  float cost[DLMAX];
#pragma unroll
  for (int d = 0; d < DLMAX; ++d) cost[d] = 0.f;

#pragma unroll 1
  for (int cg = 0; cg < CPW; ++cg) {
    const int ci = grp * CPW + cg;
    const float* wc = w + ci * 27;
#pragma unroll
    for (int p = 0; p < DLMAX; ++p) {
      if (p < Dl) {
        const float* src = vol + (((size_t)n * Dl + p) * kC + ci) * plane;
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = toff[t] >= 0 ? src[toff[t]] : 0.f;
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
          const int d = p - dz + 1;           // output plane fed by input plane p through tap dz
          if (d >= 0 && d < DLMAX) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) s = fmaf(wc[dz * 9 + t], v[t], s);
            cost[d] += s;
          }
        }
      }
      if (p & 1) asm volatile("" ::: "memory");       // bound the load hoisting: two planes (18 loads) at a time
    }
  }
#pragma unroll
  for (int d = 0; d < DLMAX; ++d) s_part[grp][d][lane] = cost[d];
  __syncthreads();
  if (grp != 0) return;
#pragma unroll
  for (int d = 0; d < DLMAX; ++d) {
    float c = bias;
#pragma unroll
    for (int g = 0; g < kSamWaves; ++g) c += s_part[g][d][lane];
    cost[d] = c;
  }
  // soft-argmin over the Dl planes (max-subtracted), in registers
  float m = -cost[0];
#pragma unroll
  for (int d = 1; d < DLMAX; ++d)
    if (d < Dl) m = fmaxf(m, -cost[d]);
  float se = 0.f, sd = 0.f;
#pragma unroll
  for (int d = 0; d < DLMAX; ++d)
    if (d < Dl) {
      const float e = expf(-cost[d] - m);
      se += e;
      sd = fmaf((float)d, e, sd);
    }
  if (live) {
    disp_low[gp] = sd / se;
    if (cost_out) {
#pragma unroll
      for (int d = 0; d < DLMAX; ++d)
        if (d < Dl) cost_out[((size_t)n * Dl + d) * plane + pix] = cost[d];
    }
  }
}

// ------------------------------------------------------------------------------------------
// K6 on the partial sums of the last aggregation layer (k_agg_x3s_dma<false, true>, sn_agg_dma.hpp):
//   P[n d'][tap][y'][x'] = sum_c w[c][tap] * vol[n][d'][c][y'][x'],  tap = dz*9 + ky*3 + kx
//   cost[d][y][x]        = b + sum_{dz, ky, kx} P[n (d + dz - 1)][tap][y + ky - 1][x + kx - 1]      (0 outside the volume)
//   disp                 = sum_d d * softmax_d(-cost)
// Workgroup = Dl waves x 64 consecutive pixels: wave d sums the 27 shifted values of plane d (every load a 256-byte run of
// one P image; each element of P is read by exactly one (d, pixel)), the Dl costs of a pixel meet in LDS and wave 0 does
// the soft-argmin exactly as k_head_softargmin does.
// ------------------------------------------------------------------------------------------
template <int DLMAX>
__global__ __launch_bounds__(64 * DLMAX) void k_softargmin_p(const float* __restrict__ P,      // [n][Dl][27][H][W]
                                                          float bias, int Dl, int H, int W, int npix_total,
                                                          float* __restrict__ disp_low,     // [n][H][W]
                                                          float* __restrict__ cost_out) {   // nullable [n][Dl][H][W]
  __shared__ float s_cost[DLMAX][64];
  const int lane = threadIdx.x & 63;
  const int d = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // this wave's plane (blockDim = 64 * Dl)
  const int gp = blockIdx.x * 64 + lane;
  const int plane = H * W;
  const bool live = gp < npix_total;
  const int n = live ? gp / plane : 0;
  const int pix = live ? gp - n * plane : 0;
  const int y = pix / W, x = pix - y * W;
  float c = bias;
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) {
    const int dp = d + dz - 1;
    if (dp < 0 || dp >= Dl) continue;             // wave-uniform
    const float* src = P + ((size_t)n * Dl + dp) * 27 * plane + (size_t)dz * 9 * plane;
    float v[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        const bool ok = live && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        v[ky * 3 + kx] = ok ? src[(size_t)(ky * 3 + kx) * plane + yy * W + xx] : 0.f;
      }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) s += v[t];
    c += s;
  }
  s_cost[d][lane] = c;
  __syncthreads();
  if (d != 0) return;
  float cost[DLMAX];
#pragma unroll
  for (int k = 0; k < DLMAX; ++k) cost[k] = k < Dl ? s_cost[k][lane] : 0.f;
  float m = -cost[0];
#pragma unroll
  for (int k = 1; k < DLMAX; ++k)
    if (k < Dl) m = fmaxf(m, -cost[k]);
  float se = 0.f, sd = 0.f;
#pragma unroll
  for (int k = 0; k < DLMAX; ++k)
    if (k < Dl) {
      const float e = expf(-cost[k] - m);
      se += e;
      sd = fmaf((float)k, e, sd);
    }
  if (live) {
    disp_low[gp] = sd / se;
    if (cost_out) {
#pragma unroll
      for (int k = 0; k < DLMAX; ++k)
        if (k < Dl) cost_out[((size_t)n * Dl + k) * plane + pix] = cost[k];
    }
  }
}

// ------------------------------------------------------------------------------------------
// K8: refinement head 3x3 conv 32->1, disp = relu(up + D*r), float + wire int32 outputs.
//   raw = rint(disp * inv_q), inv_q = 1/(D*scale)   (stereonet_node.cpp:282-288: the consumer
//   multiplies raw by scale*16*12).  One thread per output pixel, lanes along x (coalesced).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head_final(const float* __restrict__ xin,      // [n][32][Hp][Wp]
                                                    const float* __restrict__ w,        // [32][9]
                                                    float bias, const float* __restrict__ disp_low,
                                                    int hl, int wl, int Hp, int Wp, int H, int W,
                                                    float dmax, float inv_q,
                                                    float* __restrict__ out_disp,       // nullable [n][H][W]
                                                    int32_t* __restrict__ out_raw,      // nullable [n][H][W]
                                                    UpScale ups,
                                                    unsigned long long* __restrict__ stat) {   // nullable: sum |D r|
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.z;
  float moved = 0.f;
  if (x < W && y < H) {
  const size_t plane = (size_t)Hp * Wp;
  const float* src = xin + (size_t)n * kC * plane;
  float acc = bias;
  for (int ci = 0; ci < kC; ++ci) {
    const float* p = src + (size_t)ci * plane;
    const float* wc = w + ci * 9;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      if ((unsigned)yy >= (unsigned)Hp) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x + kx - 1;
        const float v = (unsigned)xx < (unsigned)Wp ? p[(size_t)yy * Wp + xx] : 0.f;
        acc = fmaf(wc[ky * 3 + kx], v, acc);
      }
    }
  }
  const float up = upsample_map(disp_low + (size_t)n * hl * wl, hl, wl, y, x, ups);
  float d = up + dmax * acc;
  moved = fabsf(dmax * acc);
  d = d > 0.f ? d : 0.f;
  const size_t o = ((size_t)n * H + y) * W + x;
  if (out_disp) out_disp[o] = d;
  if (out_raw) out_raw[o] = (int32_t)__float2int_rn(d * inv_q);
  }
  refine_stat_commit_block(stat, moved);
}


// ==========================================================================================
// fp16 refinement tower (SN_PREC_F16): the 12 C->C 3x3 (dilated) convolutions at full resolution.
//
// Tensor layout "NCHW8c" with a zero border: [n][4 channel blocks][Hs][Ws] of 16-byte slots, one
// slot = 8 consecutive fp16 channels of one pixel.  Hs = Ht + 2*kRefPad, Ws = Wt + 2*kRefPad where
// Ht x Wt is the image rounded up to whole 8x64 tiles.  The border and the tile overhang are zeroed
// once and never written, so the kernel has no bounds checks: zero padding is real memory.
//   * loads: a tile row is a contiguous run of slots -> LDS-DMA (global_load_lds_dwordx4, 1 KiB per
//     wave instruction, lane-linear destination) straight into the [block][row][col] LDS image;
//   * MFMA  v_mfma_f32_32x32x16_f16:  D[cout][pixel] += W[cout][k] * X[k][pixel], k = 16 input
//     channels of one tap.  The X fragment of lane (pixel j, k-half g) is ONE 16-byte slot of block
//     2*kk+g -> ds_read_b128 of consecutive slots (conflict-free, immediate offsets, no swizzle);
//     all 18 W fragments (9 taps x 2 k-steps) live in 72 VGPRs for the life of the persistent block;
//   * stores: lane (pixel j, half g) owns couts 8q+4g..+3 of block q -> 8-byte store; the two halves
//     of a wave fill whole 16-byte slots of 32 consecutive pixels = one contiguous 512-byte run.
// K is split in two phases of 16 channels (2 channel blocks each) that ping-pong through two LDS
// buffers: while phase p computes, the DMA for phase p+1 (same tile, or the next tile of this
// persistent block) is in flight; one barrier per phase.
// ==========================================================================================
struct RefGeom {
  int Hs, Ws;          // padded plane dims (pixels)
  int H, W;            // valid image area (the network's Hp x Wp)
  int tiles_x, tiles_y;
  int rev;             // 1: this launch walks its tiles in reverse order (see refine_level)
};

// ------------------------------------------------------------------------------------------
// v2 of the tower kernel: same tile math, deeper memory pipeline.
//   * ring of THREE phase buffers; the DMA group of phase g+2 is issued right after the barrier that
//     opens phase g, so one group is always in flight while another is being waited for;
//   * raw s_barrier + hand-counted s_waitcnt vmcnt(N) (a __syncthreads() would drain the LDS-DMA queue).
//     CDNA4 retires VMEM in order and counts stores on vmcnt, so every wave issues a CONSTANT number of
//     VMEM ops per phase: KW DMA instructions per group (tail instructions re-fetch the last slots) and
//     4*SPW unconditional 8-byte stores per tile (out-of-image pixels store zeros into the overhang,
//     which must stay zero anyway).  At the top of phase g the ops younger than group g are exactly
//     {group g+1} (+ the stores of the tile that just finished when g is even), hence
//     N = KW (+ 4*SPW), and 0 for the very last phase.
//   * TW = 32 for dilation 8 keeps three buffers inside the 160 KiB LDS.
// ------------------------------------------------------------------------------------------
template <int DIL, int TW_, int TH_ = 8, int NB_ = 3>
struct RefTile2 {
  static constexpr int TH = TH_, TW = TW_;
  static constexpr int CSEG = TW / 32, SPW = TH * CSEG / 4;
  static constexpr int ROWS = TH + 2 * DIL, COLS = TW + 2 * DIL;
  static constexpr int PLANE = ROWS * COLS;
  static constexpr int HALF = 2 * PLANE;
  static constexpr int NINST = (HALF + 63) / 64;
  static constexpr int KW = (NINST + 3) / 4;            // DMA instructions per wave per group (constant)
  static constexpr int BUF = NINST * 64;
  static constexpr int NBUF = NB_;
  static constexpr int LDS_BYTES = NBUF * BUF * 16 + 16;   // + the two tile-queue words of the dynamic schedule
  static constexpr int NSTORE = 4 * SPW;
};

// LeakyReLU as max(v, slope*v) with a raw v_max_f32: fmaxf() makes hipcc insert a canonicalising v_max in front
// (sNaN semantics) and the select form costs cmp + cndmask + mul; MFMA results are never signalling NaNs.
__device__ __forceinline__ float lrelu_fast(float v) {
  const float t = v * kSlope;
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(t));
  return r;
}

// Activation of the fp16 tower (SN_PREC_F16), round 4: every value a tower layer stores is
//     h = fp16(v);  out = max(h, fp16(slope * h))            (slope = 0.2, or 1 for "no activation")
// i.e. LeakyReLU is applied AFTER the rounding to fp16, on packed pairs (v_cvt_pk_f16_f32, v_pk_mul_f16, v_pk_max_f16:
// 1.5 instead of 2.5 VALU instructions per value).  For v >= 0 nothing changes; for v < 0 the product is rounded from
// fp16(v) instead of v (one extra fp16 rounding: EPE vs the oracle 3.75e-4 -> 3.80e-4 px at 1280x720, 6.48e-4 -> 6.56e-4
// for the hierarchical model at 1242x375).  Epilogue VALU instructions are on the critical path of the streamed block
// (sn_stream_block.hpp): -3.5 % per block, +1.2 % end to end.  EVERY kernel of the fp16 tower uses this one function, so
// the streamed, per-layer and head-fused forms stay bit-identical to each other.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned act_pack2_f16(float a, float b, _Float16 slope) {
  half2v h;
  h[0] = (_Float16)a;
  h[1] = (_Float16)b;
  const half2v m = h * slope;
  h = __builtin_elementwise_max(h, m);
  return *reinterpret_cast<const unsigned*>(&h);
}

// Workgroup barrier for the hand-synchronised kernels: a bare s_barrier (no vmcnt drain, unlike __syncthreads())
// fenced on both sides so that hipcc cannot move LDS / global accesses across it.
__device__ __forceinline__ void block_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int DIL, int TW, int KK, int TH = 8>
__device__ __forceinline__ void ref2_compute(const uint4* lds_lane, const half8 (&wf)[18],
                                             f32x16 (&acc)[RefTile2<DIL, TW, TH>::SPW]) {
  using T = RefTile2<DIL, TW, TH>;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
    for (int s = 0; s < T::SPW; ++s) {
      const int off = ((s / T::CSEG) + ky * DIL) * T::COLS + (s % T::CSEG) * 32 + kx * DIL;
      const half8 xb = *reinterpret_cast<const half8*>(lds_lane + off);
      acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[tap * 2 + KK], xb, acc[s], 0, 0, 0);
    }
  }
}
// First phase of a tile: the accumulators START at `init` (the bias) through the C operand of the first MFMA of each
// segment — no 16 v_mov per segment to seed them.
// `between(m)` runs after the m-th MFMA (m = 0 .. 9 * SPW - 1): the caller spreads its VMEM instructions (residual
// loads) over the phase instead of issuing them as one burst in front of it — a burst of 16 loads behind a DMA group
// stalled the wave's in-order issue for ~4 k cycles (VMEM queue full), i.e. kept 36 MFMAs waiting behind it.
template <int DIL, int TW, int TH = 8, class Between>
__device__ __forceinline__ void ref2_compute_init(const uint4* lds_lane, const half8 (&wf)[18], const f32x16& init,
                                                  f32x16 (&acc)[RefTile2<DIL, TW, TH>::SPW], Between between) {
  using T = RefTile2<DIL, TW, TH>;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
    for (int s = 0; s < T::SPW; ++s) {
      const int off = ((s / T::CSEG) + ky * DIL) * T::COLS + (s % T::CSEG) * 32 + kx * DIL;
      const half8 xb = *reinterpret_cast<const half8*>(lds_lane + off);
      if (tap == 0) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], xb, init, 0, 0, 0);
      else acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[tap * 2], xb, acc[s], 0, 0, 0);
      between(tap * T::SPW + s);
    }
  }
}

// Tiles beyond the first two rounds are handed out by one device-scope counter per XCD band
// (tile_ctr[16 * xcd], zeroed by the host before the launch) instead of a static stride.  When the
// low-resolution branch runs on the other stream the hardware places the tower's workgroups unevenly (two on one
// CU, none on a CU that is full of other kernels' waves); a static partition then waits for the slowest CU.
// Wave 0 fetches tile ti+2 during tile ti: the returning atomic is issued right after the phase-g0 barrier,
// i.e. it is OLDER than every DMA group / store the counted waits below leave in flight, so the existing
// vmcnt immediates stay valid (they only ever name the youngest ops); the value crosses to the other waves
// through two LDS words behind the ring.
// TH = 16 (dilated layers): a taller tile re-reads less halo through L2 (dilation 8: 3.0x instead of 4.5x the
// tile's own pixels, dilation 4: 1.9x instead of 2.25x).
// NB = 2 (dilation 8): two ring buffers instead of three.  The 110 KB three-buffer ring of the 24x48 halo tile
// allows one workgroup per CU only; with 74 KB two fit, and eight resident waves hide more latency than a DMA
// group in flight across the barrier does (dilation 4 went 76 -> 58 us per launch by the same move to two
// workgroups per CU).  Schedule with NB = 2: group g+1 is issued after the barrier of phase g (its buffer was
// read in phase g-1) and must be complete at the barrier of phase g+1.
template <int DIL, int TW, bool RES, int TH = 8, int MINW = 2, int NB = 3>
__global__ __launch_bounds__(256, MINW) void k_ref_conv_f16_v2(const uint4* __restrict__ in, uint4* out,
                                                            const uint4* res, const uint4* __restrict__ wfrag,
                                                            const float* __restrict__ bias, RefGeom g, int nimg,
                                                            int lrelu, unsigned* tile_ctr) {
  using T = RefTile2<DIL, TW, TH, NB>;
  static_assert(NB == 2 || NB == 3, "ring depth");
  // (Moving residual and output as whole 16-byte slots — lane (j, g) handling block 2i + g and the halves traded with
  // v_permlane32_swap, 8 instead of 16 VMEM instructions per tile and direction — measured +-0 twice, this round with
  // loads too; note for anyone retrying: an inline-asm global_store_dwordx4 needs the wait states of the VMEM-store-
  // data hazard by hand, hipcc cannot see through the asm and dword 0 of lanes 12-15 of every row of 16 came out stale.)
  constexpr int NIO = T::NSTORE;                             // VMEM instructions per tile: residual loads, stores
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, gh = lane >> 5;

  const int per_img = g.tiles_x * g.tiles_y;
  const int total = per_img * nimg;
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
  const int t_begin = (int)((long)xcd * total / 8), t_end = (int)((long)(xcd + 1) * total / 8);
  const int t0 = t_begin + lb;
  if (t0 >= t_end) return;

  const int seg0 = wave * T::SPW;
  const int lane_off = gh * T::PLANE + (seg0 / T::CSEG) * T::COLS + j;

  // Addressing: every global address is  uniform 32-bit byte offset of the tile (SGPRs, recomputed per tile)
  // + a per-lane 32-bit byte offset that never changes (VGPRs, computed once).  This keeps 64-bit VALU
  // address arithmetic (previously ~40 % of the loop's VALU work) out of the tile loop.  The host guarantees
  // that a tensor stays below 4 GiB.
  const unsigned plane_b = (unsigned)g.Hs * (unsigned)g.Ws * 16u;           // bytes per channel block
  unsigned dma_voff[T::KW];                                                    // per DMA instruction of this wave
#pragma unroll
  for (int k = 0; k < T::KW; ++k) {
    int i = wave + 4 * k;
    i = i < T::NINST ? i : T::NINST - 1;                 // constant op count per wave: repeat the last chunk
    int s = i * 64 + lane;
    s = s < T::HALF ? s : T::HALF - 1;
    const int pc = s / T::PLANE;
    const int rem = s - pc * T::PLANE;
    const int r = rem / T::COLS;
    const int c = rem - r * T::COLS;
    dma_voff[k] = (unsigned)pc * plane_b + ((unsigned)r * (unsigned)g.Ws + (unsigned)c) * 16u;
  }
  unsigned io_voff[T::SPW];                                                    // output / residual, per segment
#pragma unroll
  for (int s = 0; s < T::SPW; ++s) {
    const int seg = seg0 + s;
    io_voff[s] = ((unsigned)(seg / T::CSEG) * (unsigned)g.Ws + (unsigned)((seg % T::CSEG) * 32 + j)) * 16u + gh * 8u;
  }

  // tile index -> (image, row, column): FastDiv (one multiply-high + one correction per division); hipcc's general
  // signed division was ~40 scalar instructions per tile in a loop whose instruction count is what bounds it.
  const FastDiv div_img((unsigned)per_img), div_tx((unsigned)g.tiles_x);
  auto tile_xy = [&](int t, int& img, int& y0, int& x0) {      // t = global tile index
    unsigned rem, tx;
    if (g.rev) t = t_begin + (t_end - 1 - t);                  // reverse walk inside the XCD band
    img = (int)div_img.divmod((unsigned)t, rem);
    const int ty = (int)div_tx.divmod(rem, tx);
    y0 = ty * T::TH;
    x0 = (int)tx * T::TW;
  };
  // uniform byte offset of (image, channel block 0, padded row y, padded col x)
  auto tile_base = [&](int img, int y, int x) -> unsigned {
    return (((unsigned)img * 4u * (unsigned)g.Hs + (unsigned)(y + kRefPad)) * (unsigned)g.Ws + (unsigned)(x + kRefPad)) * 16u;
  };
  // DMA group of channel half `half` of tile (img, y0, x0) -> ring buffer `slot`.  The ring position of phase g0 = 2 ti
  // is carried in r0 and advanced by two per tile (g % 3 through a multiply-high, several times per tile, was another
  // ~25 scalar instructions)
  auto issue = [&](int slot, int half, unsigned src_base) {   // src_base = tile_base(img, y0 - DIL, x0 - DIL)
    const char* src = reinterpret_cast<const char*>(in) + (src_base + 2u * (unsigned)half * plane_b);
    uint4* dst = lds + slot * T::BUF;
#pragma unroll
    for (int k = 0; k < T::KW; ++k) {
      int i = wave + 4 * k;
      i = i < T::NINST ? i : T::NINST - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + dma_voff[k]),
                                       (__attribute__((address_space(3))) void*)(dst + i * 64), 16, 0, 0);
    }
  };

  int img, y0, x0, nimg_ = 0, ny0 = 0, nx0 = 0;
  tile_xy(t0, img, y0, x0);
  unsigned cur_base = tile_base(img, y0 - DIL, x0 - DIL), nxt_base = 0;   // DMA source of this / the next tile
  issue(0, 0, cur_base);
  if (NB == 3) issue(1, 1, cur_base);
  // Weights and bias are fetched AFTER the first tile's DMA groups are on their way, so the two latencies of a
  // workgroup's start-up overlap (~1-2 us of every launch).  The loads are consumed HERE (hipcc waits for them with
  // vmcnt(0), which also lands the first groups) so that its own vmcnt bookkeeping is clean before the loop; otherwise
  // it re-inserts s_waitcnt vmcnt(0) at the first MFMA of every iteration (loop-header merge) and drains the ring.
  half8 wf[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const uint4 v = wfrag[i * 64 + lane];
    wf[i] = *reinterpret_cast<const half8*>(&v);
  }
  f32x16 bv;
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * gh];
#pragma unroll
  for (int i = 0; i < 18; ++i) asm volatile("" : "+v"(wf[i]));
  asm volatile("" : "+v"(bv));
  wait_vmcnt<0>();
  int t_next = t0 + nlb;                                      // tile ti+1 (second round is static as well)
  int t_next2 = t0 + 2 * nlb;                                 // tile ti+2: handed out by the queue from here on
  unsigned* tile_slot = reinterpret_cast<unsigned*>(lds + NB * T::BUF);  // not volatile: that would drain vmcnt
  unsigned* const my_ctr = tile_ctr + 16 * xcd;

  f32x16 acc[T::SPW];
  int r0 = 0;                                                 // ring buffer of phase g0 = 2 ti: g0 % NB
  for (int ti = 0;; ++ti) {
    const int r1 = NB == 2 ? 1 : (r0 + 1 >= 3 ? r0 - 2 : r0 + 1);          // (g0 + 1) % NB
    const int r2 = NB == 2 ? 0 : (r0 + 2 >= 3 ? r0 - 1 : r0 + 2);          // (g0 + 2) % NB; (g0 + 3) % 3 = r0
    // an SGPR integer, not an i1: hipcc otherwise carries the flag as a lane mask and re-derives its negation through
    // v_cndmask / v_cmp at every use
    const int has_next = __builtin_amdgcn_readfirstlane(t_next < t_end ? 1 : 0);
    if (has_next) {                                           // one coordinate decode per tile
      tile_xy(t_next, nimg_, ny0, nx0);
      nxt_base = tile_base(nimg_, ny0 - DIL, nx0 - DIL);
    }
    // ---- phase g0 (channels 0..15) ----
    if (NB == 3) {
      if (ti == 0) wait_vmcnt<T::KW>();                       // younger than group 0: group 1
      else wait_vmcnt<T::KW + NIO>();                         // ... plus the previous tile's stores
    } else {
      if (ti == 0) wait_vmcnt<0>();                           // two buffers: nothing else is in flight yet
      else wait_vmcnt<NIO>();                                 // only the previous tile's stores are younger
    }
    block_barrier();
    // `fetched` is written asynchronously by the returning atomic: it is defined opaquely up front and tied
    // read-write into the asm so that hipcc keeps it in one register and never copies it before the wait below.
    unsigned fetched;
    asm volatile("" : "=v"(fetched));
    if (has_next && wave == 0) {                              // wave-uniform branch; lane 0 only inside the asm
      unsigned one = 1;
      unsigned long long saved_exec;
      asm volatile(
          "s_mov_b64 %1, exec\n\t"
          "s_mov_b64 exec, 1\n\t"
          "v_mov_b32 %0, -1\n\t"                      // sentinel: overwritten when the atomic returns
          "global_atomic_add %0, %2, %3, off sc0\n\t"
          "s_mov_b64 exec, %1"
          : "+v"(fetched), "=&s"(saved_exec)
          : "v"(my_ctr), "v"(one)
          : "memory");
    }
    if (NB == 3) {
      if (has_next) issue(r2, 0, nxt_base);                   // phase g0 + 2: first channel half of the next tile
    } else {
      issue(1, 1, cur_base);                                  // second channel half of THIS tile
    }
    // residual: NSTORE 8-byte loads as inline asm, spread over THIS phase's MFMAs (one after every second MFMA), a
    // whole tile before the epilogue needs them.  They sit between two DMA groups in the in-order VMEM queue, so the
    // counted waits below name them explicitly.  hipcc does not track asm loads: the "+v" statement behind the wait
    // is what orders their use.  SGPR base + 32-bit VGPR offset: no 64-bit VALU add per load.
    const unsigned tb = tile_base(img, y0, x0);
    uint2 rres[RES ? T::NSTORE : 1];
    const char* const rbase = reinterpret_cast<const char*>(res) + tb;                         // uniform
    auto res_load = [&](int m) {
      if (RES && (m & 1) && (m >> 1) < T::NSTORE) {
        const int i = m >> 1, q = i / T::SPW, sg = i - q * T::SPW;
        const char* rq = rbase + (unsigned)q * plane_b;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(rres[RES ? sg * 4 + q : 0]) : "v"(io_voff[sg]), "s"(rq) : "memory");
      }
    };
    // accumulators start at the bias (C operand of each segment's first MFMA): no add in the epilogue, no seeding moves
    ref2_compute_init<DIL, TW, TH>(lds + r0 * T::BUF + lane_off, wf, bv, acc, res_load);

    // ---- phase g0+1 (channels 16..31) ----
    // younger than group g0+1: (NB = 3, another tile follows) group g0+2, and the residual loads behind it
    constexpr int NRES = RES ? NIO : 0;
    if (NB == 3 && has_next) wait_vmcnt<T::KW + NRES>();
    else wait_vmcnt<NRES>();                                  // last tile of this block / two-buffer ring
    if (has_next && wave == 0) {                              // the atomic is older than group g0+2: it has returned
      asm volatile("" : "+v"(fetched));
      // belt and braces: should a returning atomic ever be retired out of order with the DMA groups, the sentinel
      // is still there -> drain and read again (never taken in practice; costs one readfirstlane + compare per tile)
      if (__builtin_amdgcn_readfirstlane(fetched) == 0xFFFFFFFFu) {
        wait_vmcnt<0>();
        asm volatile("" : "+v"(fetched));
      }
      if (lane == 0) tile_slot[ti & 1] = fetched;
      // block_barrier() is a bare s_barrier: an ordinary LDS write has to be retired by its own wave first
      // (the ring itself is filled by LDS-DMA and ordered by vmcnt)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    block_barrier();
    if (has_next) t_next2 = t_begin + 2 * nlb + (int)__builtin_amdgcn_readfirstlane(tile_slot[ti & 1]);
    else t_next2 = t_end;

    const int more = has_next;
    if (more) issue(NB == 3 ? r0 : 0, NB == 3 ? 1 : 0, nxt_base);          // phase g0 + 3 (NB = 2: g0 + 2, first half of the next tile)
    ref2_compute<DIL, TW, 1, TH>(lds + r1 * T::BUF + lane_off, wf, acc);
    if (RES) {
      if (more) wait_vmcnt<T::KW>(); else wait_vmcnt<0>();
#pragma unroll
      for (int i = 0; i < T::NSTORE; ++i) asm volatile("" : "+v"(rres[i]));
    }

    // ---- epilogue: exactly NSTORE stores per wave; out-of-image pixels store zeros.  Tiles that lie
    // completely inside the image (all of them at 1280x720) skip the per-element masking. ----
    const bool interior = y0 + T::TH <= g.H && x0 + T::TW <= g.W;            // wave-uniform
    const _Float16 slope_h = lrelu ? (_Float16)kSlope : (_Float16)1.0f;         // max(h, h) = h: one code path with and without the activation
    float one = 1.0f;
    asm volatile("" : "+v"(one));                      // opaque: keeps the multiply so that hipcc selects v_fma_mix_f32
    // (the epilogue body exists twice, selected by ONE uniform branch per tile: with the test inside, hipcc put two
    // v_cndmask and a scalar mask sequence in front of every store of the interior path as well)
    auto finish = [&](int s, int q, auto inside) -> uint2 {   // bias is in the accumulator; + residual, activation, fp16
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[s][4 * q + e];
      if (RES) {
        const uint2 rw = rres[RES ? s * 4 + q : 0];
        const half4 rv = *reinterpret_cast<const half4*>(&rw);
        // v += (float)residual as ONE v_fma_mix_f32 per value (fp16 source read in place) instead of cvt + add
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf((float)rv[e], one, v[e]);
      }
      uint2 hv{act_pack2_f16(v[0], v[1], slope_h), act_pack2_f16(v[2], v[3], slope_h)};      // round, then activate (act_pack2_f16)
      if (!decltype(inside)::value) {
        const int seg = seg0 + s;
        const int y = y0 + seg / T::CSEG, x = x0 + (seg % T::CSEG) * 32 + j;
        if (!(y < g.H && x < g.W)) hv = uint2{0u, 0u};
      }
      return hv;
    };
    auto store_tile = [&](auto inside) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        char* oq = reinterpret_cast<char*>(out) + (tb + (unsigned)q * plane_b);                  // uniform
#pragma unroll
        for (int s = 0; s < T::SPW; ++s) {
          const uint2 hv = finish(s, q, inside);  // SGPR base + 32-bit VGPR offset: no 64-bit VALU add per store
          const unsigned voff = io_voff[s];       // (a local: an asm operand inside a generic lambda cannot name the capture)
          asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(voff), "v"(hv), "s"(oq) : "memory");
        }
      }
    };
    if (interior) store_tile(std::true_type{});
    else store_tile(std::false_type{});
    if (!has_next) break;
    img = nimg_;
    y0 = ny0;
    x0 = nx0;
    t_next = t_next2;
    t_next2 = t_end;
    r0 = r2;
    cur_base = nxt_base;
  }
}

// ------------------------------------------------------------------------------------------
// SN_PREC_F16X3: the v2 tower kernel on split operands.  Every activation / weight is a pair of fp16 numbers
// (hi = fp16(v), lo = fp16((v - hi) * 2^11)), i.e. 22 significant bits, and a product is evaluated with three
// fp16 MFMAs:  x*w ~= xhi*whi + (xhi*wlo + xlo*whi) * 2^-11   (the dropped xlo*wlo term is 2^-22 relative).
// Two accumulators per segment (acc0 for the hi*hi products, acc1 for the cross terms) keep the scaled term
// exact until the epilogue.  Result: fp32-class accuracy (EPE ~3e-7 px vs the oracle in emulation) at 3/16 of
// the fp32-MFMA cost.  Tensors are two NCHW8c fp16 tensors (hi at `in`, lo at `in + lo_slots`), so the HBM
// traffic equals an fp32 tensor.  One workgroup per CU (LDS holds hi and lo tiles), NBUF-deep DMA ring.
// ------------------------------------------------------------------------------------------
template <int DIL, int TW, int NBUF, bool RES>
__global__ __launch_bounds__(256, 1) void k_ref_conv_f16x3(const uint4* __restrict__ in, uint4* out, const uint4* res,
                                                           size_t lo_slots,                 // hi -> lo tensor offset
                                                           const uint4* __restrict__ wfrag, // [hi 18][lo 18] x 64 slots
                                                           const float* __restrict__ bias, RefGeom g, int nimg, int lrelu) {
  using T = RefTile2<DIL, TW>;
  constexpr int KW2 = 2 * T::KW;               // DMA instructions per wave per group (hi + lo)
  constexpr int NST = 2 * T::NSTORE;           // stores per wave per tile (hi + lo)
  constexpr int RING = 2 * T::BUF;             // slots per ring entry
  static_assert(NBUF * RING * 16 <= 160 * 1024, "ring does not fit the LDS");
  static_assert(NBUF == 2 || NBUF == 3, "ring depth");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, gh = lane >> 5;

  half8 wh[18], wl[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const uint4 a = wfrag[i * 64 + lane], b = wfrag[(18 + i) * 64 + lane];
    wh[i] = *reinterpret_cast<const half8*>(&a);
    wl[i] = *reinterpret_cast<const half8*>(&b);
  }
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * gh];
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    asm volatile("" : "+v"(wh[i]));
    asm volatile("" : "+v"(wl[i]));
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(bv[r]));

  const int per_img = g.tiles_x * g.tiles_y;
  const int total = per_img * nimg;
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
  const int t_begin = (int)((long)xcd * total / 8), t_end = (int)((long)(xcd + 1) * total / 8);
  const int t0 = t_begin + lb;
  if (t0 >= t_end) return;
  const int ntiles = (t_end - t0 + nlb - 1) / nlb;
  const int G = 2 * ntiles;

  const int seg0 = wave * T::SPW;
  // addressing as in k_ref_conv_f16_v2: uniform 32-bit tile offset + fixed per-lane 32-bit offsets; the lo tensor
  // sits lo_slots*16 bytes behind the hi tensor (folded into the uniform base pointers)
  const unsigned plane_b = (unsigned)g.Hs * (unsigned)g.Ws * 16u;
  unsigned dma_voff[T::KW];
#pragma unroll
  for (int k = 0; k < T::KW; ++k) {
    int i = wave + 4 * k;
    i = i < T::NINST ? i : T::NINST - 1;
    int s = i * 64 + lane;
    s = s < T::HALF ? s : T::HALF - 1;
    const int pc = s / T::PLANE;
    const int rem = s - pc * T::PLANE;
    const int r = rem / T::COLS;
    const int c = rem - r * T::COLS;
    dma_voff[k] = (unsigned)pc * plane_b + ((unsigned)r * (unsigned)g.Ws + (unsigned)c) * 16u;
  }
  unsigned io_voff[T::SPW];
#pragma unroll
  for (int s = 0; s < T::SPW; ++s) {
    const int seg = seg0 + s;
    io_voff[s] = ((unsigned)(seg / T::CSEG) * (unsigned)g.Ws + (unsigned)((seg % T::CSEG) * 32 + j)) * 16u + gh * 8u;
  }
  const char* in_lo = reinterpret_cast<const char*>(in) + lo_slots * 16;
  const char* res_lo = reinterpret_cast<const char*>(res) + lo_slots * 16;
  char* out_lo = reinterpret_cast<char*>(out) + lo_slots * 16;

  auto tile_xy = [&](int ti, int& img, int& y0, int& x0) {
    const int t = t0 + ti * nlb;
    img = t / per_img;
    const int rem = t - img * per_img;
    const int ty = rem / g.tiles_x;
    y0 = ty * T::TH;
    x0 = (rem - ty * g.tiles_x) * T::TW;
  };
  auto tile_base = [&](int img, int y, int x) -> unsigned {
    return (((unsigned)img * 4u * (unsigned)g.Hs + (unsigned)(y + kRefPad)) * (unsigned)g.Ws + (unsigned)(x + kRefPad)) * 16u;
  };
  auto issue = [&](int gp, int img, int y0, int x0) {        // hi and lo half-tiles of phase gp -> ring entry gp % NBUF
    const unsigned sb = tile_base(img, y0 - DIL, x0 - DIL) + 2u * (gp & 1) * plane_b;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      uint4* dst = lds + (gp % NBUF) * RING + part * T::BUF;
      const char* src = (part ? in_lo : reinterpret_cast<const char*>(in)) + sb;
#pragma unroll
      for (int k = 0; k < T::KW; ++k) {
        int i = wave + 4 * k;
        i = i < T::NINST ? i : T::NINST - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + dma_voff[k]),
                                         (__attribute__((address_space(3))) void*)(dst + i * 64), 16, 0, 0);
      }
    }
  };
  // three MFMAs per (tap, segment): hi*hi -> acc0; hi*lo and lo*hi -> acc1
  auto compute = [&](const uint4* ring, auto kkc, f32x16 (&a0)[T::SPW], f32x16 (&a1)[T::SPW]) {
    constexpr int kk = decltype(kkc)::value;
    const uint4* bh = ring + gh * T::PLANE + (seg0 / T::CSEG) * T::COLS + j;
    const uint4* bl = bh + T::BUF;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int s = 0; s < T::SPW; ++s) {
        const int off = ((s / T::CSEG) + ky * DIL) * T::COLS + (s % T::CSEG) * 32 + kx * DIL;
        const half8 xh = *reinterpret_cast<const half8*>(bh + off);
        const half8 xl = *reinterpret_cast<const half8*>(bl + off);
        a0[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[tap * 2 + kk], xh, a0[s], 0, 0, 0);
        a1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[tap * 2 + kk], xh, a1[s], 0, 0, 0);
        a1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[tap * 2 + kk], xl, a1[s], 0, 0, 0);
      }
    }
  };

  wait_vmcnt<0>();
  int img, y0, x0, nimg_ = 0, ny0 = 0, nx0 = 0;
  tile_xy(0, img, y0, x0);
  // prologue: groups 0 .. NBUF-2 (group 1 belongs to tile 0 as well)
  issue(0, img, y0, x0);
  if (NBUF == 3) issue(1, img, y0, x0);

  f32x16 acc0[T::SPW], acc1[T::SPW];
  for (int ti = 0; ti < ntiles; ++ti) {
    const int g0 = 2 * ti;
    if (ti + 1 < ntiles) tile_xy(ti + 1, nimg_, ny0, nx0);
    // ---- phase g0: ops younger than group g0 = groups g0+1 .. g0+NBUF-2 (+ the previous tile's stores) ----
    if (NBUF == 3) {
      if (ti == 0) wait_vmcnt<KW2>(); else wait_vmcnt<KW2 + NST>();     // group g0+1 always exists
    } else {
      if (ti == 0) wait_vmcnt<0>(); else wait_vmcnt<NST>();
    }
    block_barrier();
    if (NBUF == 3) {
      if (g0 + 2 < G) issue(g0 + 2, nimg_, ny0, nx0);
    } else {
      issue(g0 + 1, img, y0, x0);
    }
#pragma unroll
    for (int s = 0; s < T::SPW; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc0[s][r] = bv[r];
        acc1[s][r] = 0.f;
      }
    compute(lds + (g0 % NBUF) * RING, std::integral_constant<int, 0>{}, acc0, acc1);

    // ---- phase g0+1 ----
    if (NBUF == 3) {
      if (g0 + 2 < G) wait_vmcnt<KW2>(); else wait_vmcnt<0>();
    } else {
      wait_vmcnt<0>();
    }
    block_barrier();
    const unsigned tb = tile_base(img, y0, x0);
    uint2 rres[RES ? 2 * T::NSTORE : 1];
    if (RES) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const char* rq = reinterpret_cast<const char*>(res) + (tb + (unsigned)q * plane_b);
        const char* rql = res_lo + (tb + (unsigned)q * plane_b);
#pragma unroll
        for (int s = 0; s < T::SPW; ++s) {
          const char* rp = rq + io_voff[s];
          const char* rl = rql + io_voff[s];
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rres[RES ? 2 * (s * 4 + q) : 0]) : "v"(rp) : "memory");
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rres[RES ? 2 * (s * 4 + q) + 1 : 0]) : "v"(rl) : "memory");
        }
      }
    }
    bool more;
    if (NBUF == 3) {
      more = g0 + 3 < G;
      if (more) issue(g0 + 3, nimg_, ny0, nx0);
    } else {
      more = g0 + 2 < G;
      if (more) issue(g0 + 2, nimg_, ny0, nx0);
    }
    compute(lds + ((g0 + 1) % NBUF) * RING, std::integral_constant<int, 1>{}, acc0, acc1);
    if (RES) {
      if (more) wait_vmcnt<KW2>(); else wait_vmcnt<0>();
#pragma unroll
      for (int i = 0; i < 2 * T::NSTORE; ++i) asm volatile("" : "+v"(rres[i]));
    }
    // ---- epilogue: v = acc0 + acc1 * 2^-11 (+ residual), LeakyReLU, split, 2 * NSTORE stores ----
    const bool interior = y0 + T::TH <= g.H && x0 + T::TW <= g.W;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      char* oq = reinterpret_cast<char*>(out) + (tb + (unsigned)q * plane_b);
      char* oql = out_lo + (tb + (unsigned)q * plane_b);
#pragma unroll
      for (int s = 0; s < T::SPW; ++s) {
        half4 hh, hl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc0[s][4 * q + e] + acc1[s][4 * q + e] * kSplitInv;
          if (RES) {
            const uint2 rh = rres[RES ? 2 * (s * 4 + q) : 0], rl = rres[RES ? 2 * (s * 4 + q) + 1 : 0];
            const half4 vh = *reinterpret_cast<const half4*>(&rh);
            const half4 vl = *reinterpret_cast<const half4*>(&rl);
            v += (float)vh[e] + (float)vl[e] * kSplitInv;
          }
          if (lrelu) v = lrelu_fast(v);
          const _Float16 hi = (_Float16)v;
          hh[e] = hi;
          hl[e] = (_Float16)((v - (float)hi) * kSplitScale);
        }
        if (!interior) {
          const int seg = seg0 + s;
          const int y = y0 + seg / T::CSEG, x = x0 + (seg % T::CSEG) * 32 + j;
          if (!(y < g.H && x < g.W)) {
            hh = half4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            hl = hh;
          }
        }
        *reinterpret_cast<half4*>(oq + io_voff[s]) = hh;
        *reinterpret_cast<half4*>(oql + io_voff[s]) = hl;
      }
    }
    img = nimg_;
    y0 = ny0;
    x0 = nx0;
  }
}

// ------------------------------------------------------------------------------------------
// K7a: refinement input conv (4 -> 32, 3x3, LeakyReLU) for the fp16 modes, writing the tower's NCHW8c tensor.
// Inputs are produced on the fly: channel 0 = bilinear x16 upsample of the low-resolution disparity / D (a float,
// carried as an fp16 hi/lo pair), channels 1..3 = left-eye int8 planes / 128 (exact in fp16).  The generic
// implicit GEMM ran this K = 36 layer on the exact-fp32 MFMA (18 x 64 cycles per 32 pixels) behind a scalar
// per-element loader.  Here every staged pixel is ONE 16-byte LDS slot
//     [d_hi, Y, U, V, d_lo, 0, 0, 0]                       (K packing: k = 8 * tap + slot entry)
// so the B fragment of lane (pixel j, k-group g) at K-step t is a single aligned ds_read_b128 of the slot of tap
// 2t + g (tap 9 = zero weights), and the layer is 5 K-steps x 2 fp16 MFMAs = 320 matrix cycles per 32 pixels:
//     acc0 += [wh_d, wh_Y, wh_U, wh_V, 0, ...] . slot        acc1 += [wl_d, wl_Y, wl_U, wl_V, wh_d, 0, ...] . slot
//     out   = lrelu(acc0 + acc1 / 2048 + bias)               (w = wh + wl / 2048, d = d_hi + d_lo / 2048)
// Staging: one thread builds 4 consecutive pixels (three aligned dword loads of int8 + four upsample evaluations)
// for the next tile while the MFMAs of the current one run; persistent workgroups, LDS double buffered.
// SPLIT = true writes the hi/lo pair of tensors of SN_PREC_F16X3.
// PYR = true (coarser levels of the hierarchical refinement): the image planes come from the float pyramid
// [n][3][g.H][g.W]; their values are not exact in fp16, so the slot carries them as hi/lo pairs too,
//     [d_hi, Y_hi, U_hi, V_hi, d_lo, Y_lo, U_lo, V_lo]
// and fragment b of the weights holds the hi weights at entries 4..7 (for the int8 source entries 5..7 are zero).
// The disparity comes from `disp_low` upsampled by `ups` (x16 from the soft-argmin map, x2 from the level below).
// ------------------------------------------------------------------------------------------
// VMEM instructions are what a tile's time is made of (as in the tower, §5 of DESIGN.md): the four pixels of a staging
// unit share their four bilinear taps when the upsample is x16 (an aligned group of four never straddles a cell
// boundary, which lies between columns 16n+7 and 16n+8): 4 instead of 16 tap loads per thread; outputs go out with
// SGPR-base stores on precomputed per-lane offsets; the accumulators start at the bias / at an inline zero through the
// first MFMA's C operand (50.5 -> 41.5 us per two pairs at 1280x720; __launch_bounds__(256, 2): left alone hipcc spent
// 335 registers on it and halved the occupancy).
template <int TW_>
struct RefInTile {
  static constexpr int TH = 8, TW = TW_, CSEG = TW / 32;
  static constexpr int ROWS = TH + 2, COLS = TW + 8;        // staged window starts 4 px left of the tile (dword aligned)
  static constexpr int BUF = ROWS * COLS;                   // slots per buffer
  static constexpr int LDS_BYTES = 2 * BUF * 16;
  static constexpr int NUNIT = ROWS * (COLS / 4);           // 4-pixel staging units per tile (180 / 100 <= 256 threads)
  static constexpr int SPW = TH * CSEG / 4;
};

template <bool SPLIT, bool PYR, int TW>
__global__ __launch_bounds__(256, 2) void k_refin_f16(const float* __restrict__ disp_low,   // [n][hl][wl]
                                                   const void* __restrict__ img_src,     // int8 [n][6][H][W] / PYR: float [n][3][g.H][g.W]
                                                   int hl, int wl, int H, int W, float inv_d, UpScale ups,
                                                   const uint4* __restrict__ wfrag,      // [5][a|b][64]
                                                   const float* __restrict__ bias, uint4* __restrict__ out,
                                                   size_t lo_off_bytes, RefGeom g, int nimg, int al4) {
  using T = RefInTile<TW>;
  static_assert(T::NUNIT <= 256, "one staging unit per thread");
  extern __shared__ __attribute__((aligned(16))) uint4 s_px[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, gk = lane >> 5;

  half8 wa[5], wb[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const uint4 a = wfrag[(2 * t) * 64 + lane], b = wfrag[(2 * t + 1) * 64 + lane];
    wa[t] = *reinterpret_cast<const half8*>(&a);
    wb[t] = *reinterpret_cast<const half8*>(&b);
  }
  f32x16 bv;
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * gk];
  int koff[5];                               // slot offset of tap 2t + g inside the staged window
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    int tap = 2 * t + gk;
    tap = tap < 9 ? tap : 0;
    koff[t] = (tap / 3) * T::COLS + (tap % 3) + 3;
  }

  const int per_img = g.tiles_x * g.tiles_y;
  const int total = per_img * nimg;
  // staging unit of this thread: row ur of the window, pixels 4*uq .. 4*uq+3
  const int ur = tid / (T::COLS / 4), uq = tid - ur * (T::COLS / 4);
  const bool unit = tid < T::NUNIT;
  const int8_t* const in6 = reinterpret_cast<const int8_t*>(img_src);
  const float* const pyr = reinterpret_cast<const float*>(img_src);
  const bool x16 = ups.rs == 1.0f / 16.0f;                  // uniform
  uint32_t pimg[3];
  float pf[PYR ? 3 : 1][4];
  float pd[4];
  bool pval[4];
  auto fetch = [&](int tile) {
    if (g.rev) tile = total - 1 - tile;
    const int img = tile / per_img, rem = tile - img * per_img;
    const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
    const int y = ty * T::TH - 1 + ur, x = tx * T::TW - 4 + 4 * uq;
    pimg[0] = pimg[1] = pimg[2] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pd[k] = 0.f;
      pval[k] = false;
    }
    if (!unit || (unsigned)y >= (unsigned)g.H) return;
    const float* dl = disp_low + (size_t)img * hl * wl;
#pragma unroll
    for (int k = 0; k < 4; ++k) pval[k] = (unsigned)(x + k) < (unsigned)g.W;
    if (x16 && x >= 0) {
      // the four pixels of an aligned group lie in one cell of the x16 grid: one set of taps (upsample_map's arithmetic)
      float sy = ((float)y + 0.5f) * ups.rs - 0.5f;
      sy = sy < 0.f ? 0.f : sy;
      float sx0 = ((float)x + 0.5f) * ups.rs - 0.5f;
      sx0 = sx0 < 0.f ? 0.f : sx0;
      const int y0 = (int)sy, x0 = (int)sx0;
      const int y1 = y0 < hl - 1 ? y0 + 1 : y0, x1 = x0 < wl - 1 ? x0 + 1 : x0;
      const float ly = sy - (float)y0, hy = 1.0f - ly;
      const float a00 = dl[y0 * wl + x0], a01 = dl[y0 * wl + x1], a10 = dl[y1 * wl + x0], a11 = dl[y1 * wl + x1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float sx = ((float)(x + k) + 0.5f) * ups.rs - 0.5f;
        sx = sx < 0.f ? 0.f : sx;
        const float lx = sx - (float)x0, hx = 1.0f - lx;
        const float v = hy * (hx * a00 + lx * a01) + ly * (hx * a10 + lx * a11);
        pd[k] = pval[k] ? v * ups.mul * inv_d : 0.f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (pval[k]) pd[k] = upsample_map(dl, hl, wl, y, x + k, ups) * inv_d;
    }
    if (PYR) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* row = pyr + (((size_t)img * 3 + c) * g.H + y) * (size_t)g.W;
#pragma unroll
        for (int k = 0; k < 4; ++k) pf[PYR ? c : 0][k] = pval[k] ? row[x + k] : 0.f;
      }
    } else if (y < H) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int8_t* row = in6 + (((size_t)img * 6 + c) * H + y) * (size_t)W;
        uint32_t v = 0;
        if (al4) {
          if (x >= 0 && x + 3 < W) v = *reinterpret_cast<const uint32_t*>(row + x);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if ((unsigned)(x + k) < (unsigned)W) v |= (uint32_t)(uint8_t)row[x + k] << (8 * k);
        }
        pimg[c] = v;
      }
    }
  };
  auto commit = [&](uint4* buf) {
    if (!unit) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      half8 sl;
#pragma unroll
      for (int e = 0; e < 8; ++e) sl[e] = (_Float16)0.f;
      if (pval[k]) {
        const _Float16 dh = (_Float16)pd[k];
        sl[0] = dh;
        sl[4] = (_Float16)((pd[k] - (float)dh) * kSplitScale);
        if (PYR) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float v = pf[PYR ? c : 0][k];
            const _Float16 vh = (_Float16)v;
            sl[1 + c] = vh;
            sl[5 + c] = (_Float16)((v - (float)vh) * kSplitScale);
          }
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) sl[1 + c] = (_Float16)((float)(int8_t)(pimg[c] >> (8 * k)) * (1.0f / 128.0f));
        }
      }
      buf[ur * T::COLS + 4 * uq + k] = *reinterpret_cast<const uint4*>(&sl);
    }
  };

  int tile = blockIdx.x;
  if (tile >= total) return;
  fetch(tile);
  commit(s_px);
  __syncthreads();
  int cur = 0;
  const unsigned plane_b = (unsigned)g.Hs * (unsigned)g.Ws * 16u;
  unsigned io_voff[T::SPW];                  // per-lane byte offset of this lane's 8 output bytes inside the tile
#pragma unroll
  for (int s = 0; s < T::SPW; ++s) {
    const int seg = wave * T::SPW + s;
    io_voff[s] = ((unsigned)(seg / T::CSEG) * (unsigned)g.Ws + (unsigned)((seg % T::CSEG) * 32 + j)) * 16u + gk * 8u;
  }
  for (; tile < total; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    if (nxt < total) fetch(nxt);
    const uint4* buf = s_px + cur * T::BUF;
    f32x16 acc0[T::SPW], acc1[T::SPW];
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
#pragma unroll
      for (int s = 0; s < T::SPW; ++s) {
        const int seg = wave * T::SPW + s;
        const int srow = seg / T::CSEG, scol = (seg % T::CSEG) * 32;
        const uint4 xv = buf[koff[t] + srow * T::COLS + scol + j];
        const half8 xb = *reinterpret_cast<const half8*>(&xv);
        // the first K-step starts the accumulators at the bias / at zero through the C operand (no seeding moves)
        acc0[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[t], xb, t == 0 ? bv : acc0[s], 0, 0, 0);
        acc1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[t], xb, t == 0 ? zero : acc1[s], 0, 0, 0);
      }
    }
    {
      const int tile_m = g.rev ? total - 1 - tile : tile;
      const int img = tile_m / per_img, rem = tile_m - img * per_img;
      const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
      const int y0 = ty * T::TH, x0 = tx * T::TW;
      const bool interior = y0 + T::TH <= g.H && x0 + T::TW <= g.W;            // wave-uniform
      // uniform byte offset of (image, block 0, padded row y0, padded col x0); the host keeps a tensor below 4 GiB
      const unsigned tb = (((unsigned)img * 4u * (unsigned)g.Hs + (unsigned)(y0 + kRefPad)) * (unsigned)g.Ws +
                           (unsigned)(x0 + kRefPad)) * 16u;
#pragma unroll
      for (int s = 0; s < T::SPW; ++s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          char* oq = reinterpret_cast<char*>(out) + (tb + (unsigned)q * plane_b);                  // uniform
          char* oql = oq + lo_off_bytes;
          half4 hh, hl4;
          if (SPLIT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc0[s][4 * q + e] + acc1[s][4 * q + e] * kSplitInv;
              v = lrelu_fast(v);
              const _Float16 hi = (_Float16)v;
              hh[e] = hi;
              hl4[e] = (_Float16)((v - (float)hi) * kSplitScale);
            }
          } else {      // SN_PREC_F16: the tower's activation rule (act_pack2_f16: round, then LeakyReLU on the packed pair)
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc0[s][4 * q + e] + acc1[s][4 * q + e] * kSplitInv;
            const uint2 pk{act_pack2_f16(v[0], v[1], (_Float16)kSlope), act_pack2_f16(v[2], v[3], (_Float16)kSlope)};
            hh = *reinterpret_cast<const half4*>(&pk);
          }
          bool ok = true;
          if (!interior) {
            const int seg = wave * T::SPW + s;
            ok = y0 + seg / T::CSEG < g.H && x0 + (seg % T::CSEG) * 32 + j < g.W;   // never write the zero border
          }
          if (ok) {
            *reinterpret_cast<half4*>(oq + io_voff[s]) = hh;
            if (SPLIT) *reinterpret_cast<half4*>(oql + io_voff[s]) = hl4;
          }
        }
      }
    }
    if (nxt < total) commit(s_px + (cur ^ 1) * T::BUF);
    lds_barrier();
    cur ^= 1;
  }
}

// K8 for the fp16 tower: 3x3 conv 32->1 on the NCHW8c tensor, disp = relu(up + D*r), outputs as k_head_final.
// A per-pixel VALU kernel moves every input slot through L1 nine times (36 16-byte loads and 576 cvt / fma per pixel:
// 40 us per two pairs, 1.78x the tensor's bytes from HBM).  Here the contraction over the 32 channels runs on the
// matrix core with the nine TAPS as the M dimension:
//     P[tap][pixel] = sum_c w[c][tap] * x[c][pixel]            one 32x32x16 MFMA per 16 channels and 32 pixels
//     r(y, x)       = bias + sum_tap P[tap][y + ky - 1][x + kx - 1]
// x is read ONCE, straight from global memory as the MFMA's B operand (lane (j, g) of K-step kk needs the 8 fp16
// channels of block 2 kk + g of pixel j = one 16-byte slot of the NCHW8c tensor: no LDS staging, no conversion);
// the fp32 weights are split hi / lo (22 bits) into two A fragments built in registers, so the products are exact
// and only the fp32 summation order differs from the scalar form.  P of a (TH + 2) x 64 pixel window goes to LDS
// (rows 0..8 of the accumulator tile are the nine taps), then every thread sums nine shifted P values per output
// pixel of the TH x 62 tile, adds the upsampled disparity and writes both outputs.
// SPLIT (SN_PREC_F16X3): x = hi + lo / 2048 from the two tensors, three MFMAs per K-step.
// ------------------------------------------------------------------------------------------
template <int TH_>
struct HeadTile {
  static constexpr int TH = TH_, TWO = 62;                   // output tile
  static constexpr int RP = TH + 2, CP = 64;                 // P window
  static constexpr int NSEG = RP * 2, SPW = (NSEG + 3) / 4;  // 32-pixel segments, per wave
  static constexpr int PLANE = RP * CP;
  static constexpr int LDS_BYTES = 9 * PLANE * 4;
};

template <bool SPLIT, int TH>
__global__ __launch_bounds__(256) void k_head_final_f16(const uint4* __restrict__ xin, size_t lo_slots, RefGeom g,
                                                        const float* __restrict__ w,      // [32][9]
                                                        float bias, const float* __restrict__ disp_low, int hl,
                                                        int wl, int H, int W, float dmax, float inv_q,
                                                        float* __restrict__ out_disp, int32_t* __restrict__ out_raw,
                                                        int tiles_x, int tiles_y, UpScale ups,
                                                        unsigned long long* __restrict__ stat) {   // nullable: sum |D r|
  using T = HeadTile<TH>;
  extern __shared__ __attribute__((aligned(16))) float s_p[];           // [9][RP][CP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, gh = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, n = t / tiles_y;
  const int y0 = ty * T::TH, x0 = tx * T::TWO;

  // B operands of this wave's segments: issued first, all of them in flight together
  const size_t plane_s = (size_t)g.Hs * g.Ws;
  const uint4* img = xin + (size_t)n * 4 * plane_s;
  uint4 xb[T::SPW][2], xl[SPLIT ? T::SPW : 1][2];
#pragma unroll
  for (int s = 0; s < T::SPW; ++s) {
    int seg = wave * T::SPW + s;
    seg = seg < T::NSEG ? seg : T::NSEG - 1;                 // tail waves repeat the last segment (same P values)
    const int prow = seg >> 1, pcol = (seg & 1) * 32 + j;
    const size_t slot = (size_t)(y0 - 1 + prow + kRefPad) * g.Ws + (x0 - 1 + pcol + kRefPad);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      xb[s][kk] = img[(size_t)(2 * kk + gh) * plane_s + slot];
      if (SPLIT) xl[s][kk] = img[(size_t)(2 * kk + gh) * plane_s + slot + lo_slots];
    }
  }
  // A fragments: row i = tap (rows 9.. are zero), k = 8 gh + e <-> channel 16 kk + 8 gh + e
  half8 ah[2], al[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float wv = j < 9 ? w[(16 * kk + 8 * gh + e) * 9 + j] : 0.f;
      const _Float16 hi = (_Float16)wv;
      ah[kk][e] = hi;
      al[kk][e] = (_Float16)((wv - (float)hi) * kSplitScale);
    }
#pragma unroll
  for (int s = 0; s < T::SPW; ++s) {
    int seg = wave * T::SPW + s;
    seg = seg < T::NSEG ? seg : T::NSEG - 1;
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a0[r] = 0.f;
      a1[r] = 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const half8 xh = *reinterpret_cast<const half8*>(&xb[s][kk]);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], xh, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk], xh, a1, 0, 0, 0);
      if (SPLIT) {
        const half8 xlo = *reinterpret_cast<const half8*>(&xl[SPLIT ? s : 0][kk]);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], xlo, a1, 0, 0, 0);
      }
    }
    // accumulator row (r & 3) + 8 (r >> 2) + 4 gh: lanes gh = 0 hold taps 0..3 (r 0..3) and 8 (r 4), gh = 1 taps 4..7
    float* dst = s_p + (seg >> 1) * T::CP + (seg & 1) * 32 + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(4 * gh + r) * T::PLANE] = a0[r] + a1[r] * kSplitInv;
    if (gh == 0) dst[8 * T::PLANE] = a0[4] + a1[4] * kSplitInv;
  }
  __syncthreads();
  const float* dl = disp_low + (size_t)n * hl * wl;
  float moved = 0.f;
  for (int p = tid; p < T::TH * T::TWO; p += 256) {
    const int oy = p / T::TWO, ox = p - oy * T::TWO;
    const int y = y0 + oy, x = x0 + ox;
    if (y >= H || x >= W) continue;
    float acc = bias;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc += s_p[(ky * 3 + kx) * T::PLANE + (oy + ky) * T::CP + ox + kx];
    const float up = upsample_map(dl, hl, wl, y, x, ups);
    float d = up + dmax * acc;
    moved += fabsf(dmax * acc);
    d = d > 0.f ? d : 0.f;
    const size_t o = ((size_t)n * H + y) * W + x;
    if (out_disp) out_disp[o] = d;
    if (out_raw) out_raw[o] = (int32_t)__float2int_rn(d * inv_q);
  }
  refine_stat_commit_block(stat, moved);
}

// K8 in SN_PREC_FP32 on the matrix core: the same taps-as-M contraction on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32):
//     P[tap][pixel] = sum_c w[c][tap] * x[c][pixel]            sixteen 32x32x2 MFMAs per 32 pixels (K = 2 channels each)
// x is the plain fp32 NCHW tensor (no border: out-of-image pixels of the P window are zero operands); lane (j, kh) of K-step
// kk reads channel 2 kk + kh of pixel j — 128-byte runs per half-wave, every element ONCE per P window instead of nine
// times through L1 (k_head_final: 288 loads + 288 fma per pixel, 81 us per 1280x720 map; BASELINE configs[1]'s single
// pair spends 2.6 % of its time there).  Products and sums are exact fp32 fmas; only the summation order differs from
// k_head_final (channels first, then the nine taps).
template <int TH>
__global__ __launch_bounds__(256) void k_head_final_mfma32(const float* __restrict__ xin,     // [n][32][Hp][Wp]
                                                           const float* __restrict__ w,       // [32][9]
                                                           float bias, const float* __restrict__ disp_low, int hl, int wl,
                                                           int Hp, int Wp, int H, int W, float dmax, float inv_q,
                                                           float* __restrict__ out_disp, int32_t* __restrict__ out_raw,
                                                           int tiles_x, int tiles_y, UpScale ups,
                                                           unsigned long long* __restrict__ stat) {   // nullable: sum |D r|
  using T = HeadTile<TH>;
  static_assert(T::NSEG % 4 == 0, "whole segments per wave");
  extern __shared__ __attribute__((aligned(16))) float s_p[];           // [9][RP][CP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, kh = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, n = t / tiles_y;
  const int y0 = ty * T::TH, x0 = tx * T::TWO;
  const size_t plane = (size_t)Hp * Wp;
  const float* src = xin + ((size_t)n * kC + kh) * plane;

  // A operands: row i = tap (rows 9.. are zero), K-step kk <-> channels 2 kk, 2 kk + 1
  float a[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) a[kk] = j < 9 ? w[(2 * kk + kh) * 9 + j] : 0.f;
  constexpr int HALF = T::SPW > 4 ? (T::SPW + 1) / 2 : T::SPW;         // segments whose loads are in flight together
#pragma unroll
  for (int s0 = 0; s0 < T::SPW; s0 += HALF) {
    float b[HALF][16];
#pragma unroll
    for (int s = 0; s < HALF; ++s) {
      const int seg = wave * T::SPW + s0 + s;
      const int yy = y0 - 1 + (seg >> 1), xx = x0 - 1 + (seg & 1) * 32 + j;
      const bool ok = s0 + s < T::SPW && (unsigned)yy < (unsigned)Hp && (unsigned)xx < (unsigned)Wp;
      const float* p = src + (size_t)(ok ? yy : 0) * Wp + (ok ? xx : 0);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float v = p[(size_t)(2 * kk) * plane];
        b[s][kk] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < HALF; ++s) {
      if (s0 + s >= T::SPW) break;
      const int seg = wave * T::SPW + s0 + s;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[s][kk], acc, 0, 0, 0);
      // accumulator row (r & 3) + 8 (r >> 2) + 4 kh: lanes kh = 0 hold taps 0..3 (r 0..3) and 8 (r 4), kh = 1 taps 4..7
      float* dst = s_p + (seg >> 1) * T::CP + (seg & 1) * 32 + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(4 * kh + r) * T::PLANE] = acc[r];
      if (kh == 0) dst[8 * T::PLANE] = acc[4];
    }
  }
  __syncthreads();
  const float* dl = disp_low + (size_t)n * hl * wl;
  float moved = 0.f;
  for (int p = tid; p < T::TH * T::TWO; p += 256) {
    const int oy = p / T::TWO, ox = p - oy * T::TWO;
    const int y = y0 + oy, x = x0 + ox;
    if (y >= H || x >= W) continue;
    float acc = bias;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc += s_p[(ky * 3 + kx) * T::PLANE + (oy + ky) * T::CP + ox + kx];
    const float up = upsample_map(dl, hl, wl, y, x, ups);
    float d = up + dmax * acc;
    moved += fabsf(dmax * acc);
    d = d > 0.f ? d : 0.f;
    const size_t o = ((size_t)n * H + y) * W + x;
    if (out_disp) out_disp[o] = d;
    if (out_raw) out_raw[o] = (int32_t)__float2int_rn(d * inv_q);
  }
  refine_stat_commit_block(stat, moved);
}

}  // namespace sn

#include "sn_stream_block.hpp"
#include "sn_stream_block_x3.hpp"
#include "sn_tower_f32.hpp"
#include "sn_agg_dma.hpp"
#include "sn_down01.hpp"
#include "sn_feat_dma.hpp"
