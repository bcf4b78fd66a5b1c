// sn_feat_dma.hpp — the thirteen 3x3 32->32 layers of the feature tower (six residual blocks + the output conv) of the
// fp16 modes on ZERO-BORDERED split-slot tensors (gfx950).
//
// Same arithmetic as k_conv_x3s<3, 1, 32, 8, 16, 16, ...> (sn_kernels.hpp; the network behind DnnNode::Run,
// stereonet_infer/src/stereonet_node.cpp:812; layer semantics DESIGN.md §2): split operands, three fp16 MFMAs per
// product, the nine taps of channel chunk 0 and of chunk 1 accumulated separately and added — bit-identical results.
// What changes is everything around the 54 MFMAs of a tile, which is where a 45 x 80 map's launch spent its time
// (12.7-16.4 us for ~2.5 us of matrix work, profiles/r04_kernel_summary_serialised.txt):
//   * 32 channels need only 36 A fragments (144 registers), so every wave keeps BOTH channel chunks and owns one
//     32-pixel segment of the 8 x 16 tile: no K split across wave pairs, no partial sums through LDS, ONE barrier
//     per tile instead of two;
//   * the tensors live in a padded layout (one zero pixel around every image, the grid rounded up to whole tiles:
//     FeatPad), so a halo tile is always inside the tensor and already contains its zero padding: staging is six
//     LDS-DMA instructions per wave (1 KiB each), no validity arithmetic, no staging registers, no commit pass;
//   * the halo tile is double buffered (2 x 24 KiB; two workgroups per CU), the DMA of tile i + 1 lands under tile i's
//     MFMAs.
#pragma once
#include "sn_kernels.hpp"

namespace sn {

struct FeatPad {      // [img][4 channel blocks][hi | lo][PH][PW] 16-byte slots, pixel (y, x) at row y + 1, column x + 1
  int H, W, PH, PW;
  __host__ __device__ static int ph(int H) { return (H + 7) / 8 * 8 + 2; }
  __host__ __device__ static int pw(int W) { return (W + 15) / 16 * 16 + 2; }
  __host__ __device__ size_t img_slots() const { return (size_t)8 * PH * PW; }
};

struct FeatDma {
  static constexpr int TR = 8, TC = 16, ROWS_IN = 10, PITCH = 18, PLANE = ROWS_IN * PITCH;
  static constexpr int NCB = 4, TAPS = 9, NK = 2 * TAPS;                 // K-steps: chunk-major, tap-minor
  static constexpr int NSL = 2 * NCB * PLANE;                           // slots of a halo tile (hi and lo): 1440
  static constexpr int KW = ((NSL + 63) / 64 + 3) / 4;                  // DMA instructions per wave: 6
  static constexpr int BUF = KW * 4 * 64;                               // slots per LDS buffer (the last instructions overshoot)
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF * 16 + 32 * 4 + 16;
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

// a.wpk = split A fragments [chunk][tap][hi | lo][lane] (upload_x3), a.out / a.res: FeatPad tensors (OUTSLOT) or fp32
// NCHW (the feature map the cost-volume builder reads); vin: FeatPad tensor.
template <bool OUTSLOT, bool HASRES>
__global__ __launch_bounds__(256, OUTSLOT ? 2 : 1) void k_feat_x3s_dma(ConvArgs a, const uint4* __restrict__ vin, FeatPad g) {
  using T = FeatDma;
  constexpr int BUF = T::BUF, KW = T::KW;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  float* s_bias = reinterpret_cast<float*>(smem4 + 2 * BUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gh = lane >> 5, j = lane & 31;

  half8 wh[T::NK], wl[T::NK];
  {
    const uint4* wsrc = reinterpret_cast<const uint4*>(a.wpk) + lane;
#pragma unroll
    for (int k = 0; k < T::NK; ++k) {
      const uint4 x = wsrc[(2 * k) * 64], y = wsrc[(2 * k + 1) * 64];
      wh[k] = *reinterpret_cast<const half8*>(&x);
      wl[k] = *reinterpret_cast<const half8*>(&y);
    }
  }
  // pixel of the segment (tile rows 2 wave, 2 wave + 1) that lane j computes; the columns of the second row are rotated
  // so that a ds_read_b128 of the wave touches 64 distinct banks (k_conv_x3s)
  constexpr int ROT = T::PITCH % 16;
  const int pr = j / 16, pc = ((j % 16) - pr * ROT) & 15;
  const int srow = 2 * wave + pr;
  const int lane_base = gh * T::PLANE + srow * T::PITCH + pc;
  if (tid < kC) s_bias[tid] = a.bias[tid];

  const unsigned phw = (unsigned)(g.PH * g.PW);
  unsigned srel[KW];
#pragma unroll
  for (int e = 0; e < KW; ++e) {
    const int L = (4 * e + wave) * 64 + lane;
    const int part = L / (T::NCB * T::PLANE);
    const int rem = L - part * (T::NCB * T::PLANE);
    const int cb = rem / T::PLANE;
    const int rc = rem - cb * T::PLANE;
    const int r = rc / T::PITCH, cc = rc - r * T::PITCH;
    srel[e] = L < T::NSL ? (unsigned)(((cb * 2 + part) * (int)phw + r * g.PW + cc) * 16) : 0u;
  }
  const FastDiv div_tx((unsigned)a.tiles_x), div_ty((unsigned)a.tiles_y);
  auto tile_off = [&](int tile, int& img, int& ty, int& tx) {     // byte offset of the tile's halo origin
    unsigned txu, tyu;
    const unsigned t2 = div_tx.divmod((unsigned)tile, txu);
    img = (int)div_ty.divmod(t2, tyu);
    ty = (int)tyu;
    tx = (int)txu;
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)img * 8u * phw + (unsigned)(ty * T::TR * g.PW + tx * T::TC)) * 16u));
  };
  const unsigned lds0 = lds_addr(smem4);
  auto dma = [&](int e, int buf, unsigned toff) {
    glds16(lds0 + (unsigned)(buf * BUF * 16) + (unsigned)((4 * e + wave) * 1024), srel[e] + toff, vin);
  };

  const int total = a.tiles_x * a.tiles_y * a.nimg;
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
  const int t_end = (int)((long)(xcd + 1) * total / 8);
  int tile = (int)((long)xcd * total / 8) + lb;
  if (tile >= t_end) return;
  int c_img, c_ty, c_tx;
  unsigned toff = tile_off(tile, c_img, c_ty, c_tx);
#pragma unroll
  for (int e = 0; e < KW; ++e) dma(e, 0, toff);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const size_t plane_o = (size_t)a.Ho * a.Wo;
  const float slope = a.lrelu ? kSlope : 1.0f;
  auto koff_of = [&](int k) {
    const int kc = k / T::TAPS, tap = k - kc * T::TAPS;
    const int ky = tap / 3, kx = tap - ky * 3;
    return 2 * kc * T::PLANE + ky * T::PITCH + kx;
  };
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;

  int cur = 0;
  for (; tile < t_end; tile += nlb) {
    const int nxt = tile + nlb;
    const int more = __builtin_amdgcn_readfirstlane(nxt < t_end ? 1 : 0);
    int n_img = c_img, n_ty = c_ty, n_tx = c_tx;
    const unsigned ntoff = more ? tile_off(nxt, n_img, n_ty, n_tx) : toff;
    const int e_y = c_ty * T::TR + srow, e_x = c_tx * T::TC + pc;
    const bool e_in = e_y < a.Ho && e_x < a.Wo;
    const unsigned pad_off = ((unsigned)(e_y + 1) * (unsigned)g.PW + (unsigned)(e_x + 1)) * 16u + gh * 8u;
    // residual of this lane's pixel: requested before the MFMAs, consumed (and only then waited for) in the epilogue —
    // converted here, hipcc waits for the loads, and with them for the DMA issued just above, in front of the first MFMA
    uint2 rraw[HASRES && OUTSLOT ? 8 : 1];
    float rv[HASRES && !OUTSLOT ? 16 : 1];
    if (HASRES && OUTSLOT) {
      const char* rs = reinterpret_cast<const char*>(a.res) + (size_t)c_img * 8 * phw * 16;
#pragma unroll
      for (int q = 0; q < 8; ++q)       // (block q >> 1, part q & 1); lanes without a pixel read their (valid) padded address
        rraw[q] = *reinterpret_cast<const uint2*>(rs + (size_t)q * phw * 16 + (e_in ? pad_off : 0u));
    }
    if (HASRES && !OUTSLOT) {
      const size_t base = (size_t)c_img * kC * plane_o + (size_t)e_y * a.Wo + e_x;
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = e_in ? a.res[base + (size_t)((r & 3) + 8 * (r >> 2) + 4 * gh) * plane_o] : 0.f;
    }

    const uint4* s_xh = smem4 + cur * BUF + lane_base;
    const uint4* s_xl = s_xh + T::NCB * T::PLANE;
    f32x16 acc0, acc1;
    float p0[16];
    uint4 bh[2], bl[2];
    bh[0] = s_xh[koff_of(0)];
    bl[0] = s_xl[koff_of(0)];
#pragma unroll
    for (int k = 0; k < T::NK; ++k) {
      const int cb_ = k & 1, nx = cb_ ^ 1;
      if (k + 1 < T::NK) {
        bh[nx] = s_xh[koff_of(k + 1)];
        bl[nx] = s_xl[koff_of(k + 1)];
      }
      __builtin_amdgcn_sched_barrier(0);
      const half8 xh = *reinterpret_cast<const half8*>(&bh[cb_]);
      const half8 xl = *reinterpret_cast<const half8*>(&bl[cb_]);
      const bool first = k % T::TAPS == 0;            // each channel chunk starts its own sums (the K halves of k_conv_x3s)
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xh, first ? zero : acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xh, first ? zero : acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xl, acc1, 0, 0, 0);
      // the next tile's DMA starts behind the first MFMAs: in front of them it would sit between the previous tile's
      // stores and hipcc's (partial) vmcnt waits for those, which count it as one of theirs
      // (no next tile: this one again, into the other buffer — no branch inside the loop)
      if (k >= 1 && 2 * (k - 1) < KW) {
        dma(2 * (k - 1), cur ^ 1, ntoff);
        if (2 * (k - 1) + 1 < KW) dma(2 * (k - 1) + 1, cur ^ 1, ntoff);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (k == T::TAPS - 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) p0[r] = acc0[r] + acc1[r] * kSplitInv;
      }
    }
    // the segment of wave w was finished by K half (w & 1) in k_conv_x3s: own half first, then the partner's
    float fin[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p1 = acc0[r] + acc1[r] * kSplitInv;
      fin[r] = p0[r] + p1;
    }
    if (HASRES && OUTSLOT) {
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(rraw[q].x), "+v"(rraw[q].y));      // first use: after the MFMAs
    }
    // the next tile has landed (the DMA is invisible to hipcc) and every wave is done with buffer `cur`: ONE barrier per
    // tile, in front of the epilogue, so that the stores drain under the next tile's MFMAs instead of in front of it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (e_in) {
      if (OUTSLOT) {
        char* const o = reinterpret_cast<char*>(a.out) + (size_t)c_img * 8 * phw * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          char* oq = o + (size_t)(2 * q) * phw * 16;
          half4 hh, hl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            float v = fin[r] + s_bias[8 * q + 4 * gh + e];
            if (HASRES) {
              const half4 rh = *reinterpret_cast<const half4*>(&rraw[HASRES ? 2 * q : 0]);
              const half4 rl = *reinterpret_cast<const half4*>(&rraw[HASRES ? 2 * q + 1 : 0]);
              v += (float)rh[e] + (float)rl[e] * kSplitInv;
            }
            v = fmaxf(v, v * slope);
            const _Float16 hi = (_Float16)v;
            hh[e] = hi;
            hl[e] = (_Float16)((v - (float)hi) * kSplitScale);
          }
          __builtin_nontemporal_store(hh, reinterpret_cast<half4*>(oq + pad_off));
          __builtin_nontemporal_store(hl, reinterpret_cast<half4*>(oq + (size_t)phw * 16 + pad_off));
        }
      } else {
        const size_t base = (size_t)c_img * kC * plane_o + (size_t)e_y * a.Wo + e_x;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (r & 3) + 8 * (r >> 2) + 4 * gh;
          float v = fin[r] + s_bias[co];
          if (HASRES) v += rv[HASRES ? r : 0];      // (no fp32-output layer has a residual in SN-K4)
          if (a.lrelu) v = v > 0.f ? v : v * kSlope;
          a.out[base + (size_t)co * plane_o] = v;
        }
      }
    }
    cur ^= 1;
    toff = ntoff;
    c_img = n_img;
    c_ty = n_ty;
    c_tx = n_tx;
  }
}

}  // namespace sn
