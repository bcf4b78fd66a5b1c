// sn_agg_dma.hpp — the 3x3x3 aggregation layers of the fp16 modes on ZERO-BORDERED split-slot volumes (gfx950).
//
// Same arithmetic as k_conv_x3s<3, 1, 96, 8, 16, 16, ...> (sn_kernels.hpp; DnnNode::Run's network, the one call
// site stereonet_infer/src/stereonet_node.cpp:812; layer semantics DESIGN.md §2): weights-stationary split-operand
// implicit GEMM, 96 virtual channels = depth planes d-1, d, d+1, K split over the two wave pairs, partials summed
// through LDS — the same MFMAs in the same order, bit-identical results.  What changes is everything AROUND the
// MFMAs, which at one wave per SIMD (the 216 weight registers) is all exposed time:
//   * the volume lives in a padded layout — one zero pixel around every plane, the plane grid rounded up to whole
//     tiles, one zero plane in front of every image's planes and one behind the last — so a halo tile is ALWAYS
//     inside the tensor and the padding it needs is already there: staging is 17 LDS-DMA instructions per wave
//     (1 KiB each, address = uniform tile offset + one fixed per-lane register), no validity arithmetic, no
//     registers for the data in flight, no commit pass (k_conv_x3s: 68 VGPRs of staged data, ~12 VALU per copy on
//     the 60 % of the tiles of a 45 x 80 x 12 volume that touch a border, and 1.2 k cycles per tile to write the
//     registers to LDS behind a barrier);
//   * the LDS tile is double buffered (2 x 68 KiB + 16 KiB of partials = 152 KiB, one workgroup per CU as before):
//     the DMA of tile i+1 lands while tile i's MFMAs run;
//   * the epilogue is deferred: after the partial hand-off a wave keeps its 16 finished sums in registers and
//     converts / stores them inside the NEXT tile's MFMA loop.
//
// Padded volume (VolPad): plane P = n (Dl + 1) + d + 1 of [4 channel blocks][hi | lo][PH][PW] 16-byte slots, pixel
// (y, x) at row y + 1, column x + 1; PH = 8 tiles_y + 2, PW = 16 tiles_x + 2.  Kernels only ever write pixels of
// the image, so the borders keep the zeros of the allocation.
#pragma once

// Development only (scripts/agg_dma_probe.hip -DSN_AGG_TIMING): s_memtime stamps of rounds 4..11 of workgroup 17 through
// a.res (unused by these layers): wave w, stamp k of round it at u64 index ((w * 8 + it - 4) * 8 + k); whole-kernel
// clock64 / wall_clock64 of every workgroup at 256 + 4 * blockIdx.x.
#ifdef SN_AGG_TIMING
#define SN_AGG_STAMP(k)                                                                                   \
  do {                                                                                                    \
    if (blockIdx.x == 17 && it >= 4 && it < 12) {                                                         \
      const unsigned long long t_ = __builtin_readcyclecounter();                                         \
      if (lane == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res))[(wave * 8 + it - 4) * 8 + (k)] = t_; \
    }                                                                                                     \
  } while (0)
#define SN_AGG_STAMP_WG(k)                                                                                \
  do {                                                                                                    \
    if (threadIdx.x == 0) {                                                                               \
      unsigned long long* d_ = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res)) + 256 + 4 * blockIdx.x + (k); \
      d_[0] = __builtin_readcyclecounter();                                                               \
      d_[2] = wall_clock64();                                                                             \
    }                                                                                                     \
  } while (0)
#else
#define SN_AGG_STAMP(k) do { } while (0)
#define SN_AGG_STAMP_WG(k) do { } while (0)
#endif

namespace sn {

struct VolPad {
  int Dl, H, W, PH, PW;
  __host__ __device__ static int ph(int H) { return (H + 7) / 8 * 8 + 2; }
  __host__ __device__ static int pw(int W) { return (W + 15) / 16 * 16 + 2; }
  __host__ __device__ size_t plane_slots() const { return (size_t)8 * PH * PW; }
  __host__ __device__ size_t planes(int npairs) const { return (size_t)npairs * (Dl + 1) + 1; }
};

// Cost volume straight into the padded layout (k_cost_slots writes the plain one).
__global__ __launch_bounds__(256) void k_cost_slots_pad(const float* __restrict__ feat, uint4* __restrict__ vol, VolPad g,
                                                        int npairs) {
  const int H = g.H, W = g.W, Dl = g.Dl;
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
  const size_t phw = (size_t)g.PH * g.PW;
  const size_t P = (size_t)n * (Dl + 1) + d + 1;
  uint4* o = vol + (P * 8 + cb * 2) * phw + (size_t)(y + 1) * g.PW + x + 1;
  o[0] = *reinterpret_cast<const uint4*>(&hi);
  o[phw] = *reinterpret_cast<const uint4*>(&lo);
}

struct AggDma {
  using T = X3sTile<3, 1, 96, 8, 16, 16>;
  static constexpr int NSL = 2 * T::NCB * T::PLANE;             // slots of a halo tile (hi and lo): 4320
  static constexpr int NINST = (NSL + 63) / 64;                 // 1 KiB DMA instructions per tile: 68
  static constexpr int KW = (NINST + 3) / 4;                    // ... per wave: 17
  static constexpr int BUF = NINST * 64;                        // slots per LDS buffer (the last instruction overshoots)
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF * 16 + (size_t)T::RED_FLOATS * 4 + 32 * 4 + 16;
  static_assert(T::PITCH == T::COLS_IN, "dense tile");
  static_assert(NINST % 4 == 0, "every wave issues KW instructions");
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

// HEADP (the LAST aggregation layer, with OUTSLOT = false): the 3x3x3 32 -> 1 output conv of the aggregation network
// starts here.  Its contraction over the 32 channels runs on the matrix core with the 27 TAPS as the M dimension, straight
// from the finished sums of a segment:  P[tap][pixel] = sum_c w[c][tap] * y[c][pixel],  y = lrelu(layer output) split
// hi / lo (the B operand, formed from the accumulator layout by one half exchange as in the tower's tail form), w split
// hi / lo (a.res = the four A fragments [K-step][hi | lo][lane], upload_agg_head_frag) — three MFMAs per K-step, six per
// segment.  The kernel then writes P [n Dl][27][H][W] fp32 instead of y [n Dl][32][H][W]; k_softargmin_p sums the 27
// shifted values per (d, pixel) — every P element is read exactly once — and does the soft-argmin.  k_head_softargmin
// re-read the 32-channel volume at 2.7x its size in fabric traffic for 864 FMAs per (d, pixel).
template <bool OUTSLOT, bool HEADP = false>
__global__ __launch_bounds__(256, 1) void k_agg_x3s_dma(ConvArgs a, const uint4* __restrict__ vin, VolPad g) {
  static_assert(!(OUTSLOT && HEADP), "the head's partial sums are an fp32 tensor");
  constexpr int OCH = HEADP ? 27 : kC;          // planes per (n, d) image of the fp32 output
  using T = AggDma::T;
  constexpr int BUF = AggDma::BUF, KW = AggDma::KW;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  float* s_red = reinterpret_cast<float*>(smem4 + 2 * BUF);
  float* s_bias = s_red + T::RED_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = wave >> 1, pset = wave & 1;
  const int gh = lane >> 5, j = lane & 31;

  // this wave's A fragments (see k_conv_x3s): chunks [khalf * NKC, (khalf + 1) * NKC) of [chunk][tap][hi|lo][lane]
  half8 wh[T::NK], wl[T::NK];
  {
    const uint4* wsrc = reinterpret_cast<const uint4*>(a.wpk) + (size_t)khalf * T::NKC * T::TAPS * 2 * 64 + lane;
#pragma unroll
    for (int k = 0; k < T::NK; ++k) {
      const uint4 x = wsrc[(2 * k) * 64], y = wsrc[(2 * k + 1) * 64];
      wh[k] = *reinterpret_cast<const half8*>(&x);
      wl[k] = *reinterpret_cast<const half8*>(&y);
    }
#pragma unroll
    for (int k = 0; k < T::NK; ++k) asm volatile("" : "+a"(wh[k]), "+a"(wl[k]));
  }
  half8 hd_h[HEADP ? 2 : 1], hd_l[HEADP ? 2 : 1];      // HEADP: A fragments of the output conv (taps as M), K-steps 0, 1
  if (HEADP) {
    const uint4* hs = reinterpret_cast<const uint4*>(a.res) + lane;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint4 x = hs[(2 * kk) * 64], y = hs[(2 * kk + 1) * 64];
      hd_h[HEADP ? kk : 0] = *reinterpret_cast<const half8*>(&x);
      hd_l[HEADP ? kk : 0] = *reinterpret_cast<const half8*>(&y);
    }
  }
  // pixel of the segment that lane j computes (rotated columns: k_conv_x3s)
  constexpr int ROT = T::PITCH % 16;
  const int pr = j / 16, pc = ((j % 16) - pr * ROT) & 15;
  int lane_base[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int seg = pset * 2 + (s == 0 ? khalf : 1 - khalf);
    const int srow = seg * 2 + pr;
    lane_base[s] = (khalf * T::HCB + gh) * T::PLANE + srow * T::PITCH + pc;
  }
  if (tid < kC) s_bias[tid] = a.bias[tid];

  // DMA instruction i = 4 e + wave fills LDS slots [64 i, 64 i + 64) of a buffer; slot L = (part, block, row, column)
  // in the tile's LDS order.  srel = byte offset of the slot's source relative to (plane P, block 0, hi, padded row
  // 8 ty, padded column 16 tx), modulo 2^32 (the previous depth plane lies below it).
  const unsigned phw = (unsigned)(g.PH * g.PW);
  unsigned srel[KW];
#pragma unroll
  for (int e = 0; e < KW; ++e) {
    const int L = (4 * e + wave) * 64 + lane;
    const int part = L / (T::NCB * T::PLANE);
    const int rem = L - part * (T::NCB * T::PLANE);
    const int vb = rem / T::PLANE;
    const int rc = rem - vb * T::PLANE;
    const int r = rc / T::PITCH, cc = rc - r * T::PITCH;
    const int dz = vb >> 2, cb = vb & 3;
    srel[e] = L < AggDma::NSL ? (unsigned)((((dz - 1) * 8 + cb * 2 + part) * (int)phw + r * g.PW + cc) * 16) : 0u;
  }
  const FastDiv div_tx((unsigned)a.tiles_x), div_ty((unsigned)a.tiles_y), div_dl((unsigned)g.Dl);
  auto tile_off = [&](int tile, int& img, int& ty, int& tx) {     // byte offset of the tile's origin, plane P
    unsigned txu, tyu, du;
    const unsigned t2 = div_tx.divmod((unsigned)tile, txu);
    img = (int)div_ty.divmod(t2, tyu);
    ty = (int)tyu;
    tx = (int)txu;
    const unsigned n = div_dl.divmod((unsigned)img, du);
    const unsigned P = (unsigned)img + n + 1u;
    return (unsigned)__builtin_amdgcn_readfirstlane((int)((P * 8u * phw + (unsigned)(ty * 8 * g.PW + tx * 16)) * 16u));
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
  SN_AGG_STAMP_WG(0);
  int c_img, c_ty, c_tx;
  unsigned toff = tile_off(tile, c_img, c_ty, c_tx);
#pragma unroll
  for (int e = 0; e < KW; ++e) dma(e, 0, toff);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const size_t plane_o = (size_t)a.Ho * a.Wo;
  const float slope = a.lrelu ? kSlope : 1.0f;
  // deferred epilogue of the previous tile: the finished sums and where they go
  float fin[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) fin[r] = 0.f;
  bool d_in = false;
  char* d_base = reinterpret_cast<char*>(a.out);
  unsigned d_off = 0;
  auto flush = [&](int q) {              // channel block q of the deferred segment (OUTSLOT) / couts r = 4q .. 4q+3
    if (OUTSLOT) {
      half4 hh, hl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = fin[4 * q + e];
        v = fmaxf(v, v * slope);
        const _Float16 hi = (_Float16)v;
        hh[e] = hi;
        hl[e] = (_Float16)((v - (float)hi) * kSplitScale);
      }
      if (d_in) {
        char* oq = d_base + (size_t)(2 * q) * phw * 16;
        __builtin_nontemporal_store(hh, reinterpret_cast<half4*>(oq + d_off));
        __builtin_nontemporal_store(hl, reinterpret_cast<half4*>(oq + (size_t)phw * 16 + d_off));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q + e;
        const int co = (r & 3) + 8 * (r >> 2) + 4 * gh;       // HEADP: the tap index (rows 27..31 of P are padding)
        float v = fin[r];
        if (!HEADP && a.lrelu) v = v > 0.f ? v : v * kSlope;  // (HEADP activated before the contraction)
        if (d_in && co < OCH) reinterpret_cast<float*>(d_base)[(size_t)d_off + (size_t)co * plane_o] = v;
      }
    }
  };

  int cur = 0;
  for (int it = 0; tile < t_end; tile += nlb, ++it) {
    SN_AGG_STAMP(0);
    const int nxt = tile + nlb;
    const int more = __builtin_amdgcn_readfirstlane(nxt < t_end ? 1 : 0);
    int n_img = c_img, n_ty = c_ty, n_tx = c_tx;
    const unsigned ntoff = more ? tile_off(nxt, n_img, n_ty, n_tx) : toff;     // no next tile: this one again
    const uint4* s_xh = smem4 + cur * BUF;
    const uint4* s_xl = s_xh + T::NCB * T::PLANE;

    f32x16 acc0[2], acc1[2];
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    auto koff_of = [&](int k) {
      const int kc = k / T::TAPS, tap = k - kc * T::TAPS;
      const int ky = tap / 3, kx = tap - ky * 3;
      return 2 * kc * T::PLANE + ky * T::PITCH + kx;
    };
    uint4 bh[2][2], bl[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bh[0][s] = s_xh[lane_base[s] + koff_of(0)];
      bl[0][s] = s_xl[lane_base[s] + koff_of(0)];
    }
#pragma unroll
    for (int k = 0; k < T::NK; ++k) {
      const int cb_ = k & 1, nx = cb_ ^ 1;
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
        const half8 xh = *reinterpret_cast<const half8*>(&bh[cb_][s]);
        const half8 xl = *reinterpret_cast<const half8*>(&bl[cb_][s]);
        acc0[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xh, k == 0 ? zero : acc0[s], 0, 0, 0);
        acc1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xh, k == 0 ? zero : acc1[s], 0, 0, 0);
        acc1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xl, acc1[s], 0, 0, 0);
      }
      if (2 * k < KW) dma(2 * k, cur ^ 1, ntoff);
      if (2 * k + 1 < KW) dma(2 * k + 1, cur ^ 1, ntoff);
      if (k >= 2 && k < 6) flush(k - 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    SN_AGG_STAMP(1);
    // ship the partial of the segment the pair partner finishes
    {
      float* dst = s_red + (size_t)wave * 16 * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = acc0[1][r] + acc1[1][r] * kSplitInv;
    }
    SN_AGG_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile has landed (the DMA is invisible to hipcc)
    SN_AGG_STAMP(3);
    lds_barrier();                      // partials visible; every wave is done with buffer `cur`
    SN_AGG_STAMP(4);
    {
      const float* src = s_red + (size_t)(wave ^ 2) * 16 * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * gh;
        fin[r] = acc0[0][r] + acc1[0][r] * kSplitInv + src[r * 64] + s_bias[OUTSLOT ? 8 * (r >> 2) + 4 * gh + (r & 3) : co];
      }
      if (HEADP) {
        // y = lrelu(sum) as hi / lo fp16 pairs: pk[q][hp] = channels 8 q + 4 gh + 2 hp, + 1 of this lane's pixel
        unsigned pkh[4][2], pkl[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int hp = 0; hp < 2; ++hp) {
            float v0 = fin[4 * q + 2 * hp], v1 = fin[4 * q + 2 * hp + 1];
            if (a.lrelu) {
              v0 = v0 > 0.f ? v0 : v0 * kSlope;
              v1 = v1 > 0.f ? v1 : v1 * kSlope;
            }
            typedef _Float16 half2v __attribute__((ext_vector_type(2)));
            half2v h, l;
            h[0] = (_Float16)v0;
            h[1] = (_Float16)v1;
            l[0] = (_Float16)((v0 - (float)h[0]) * kSplitScale);
            l[1] = (_Float16)((v1 - (float)h[1]) * kSplitScale);
            pkh[q][hp] = *reinterpret_cast<const unsigned*>(&h);
            pkl[q][hp] = *reinterpret_cast<const unsigned*>(&l);
          }
        f32x16 p0, p1;
#pragma unroll
        for (int r = 0; r < 16; ++r) p0[r] = p1[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          // half exchange (lanes 32-63 of the first operand <-> lanes 0-31 of the second): lane (j, g) ends up with the
          // whole 8-channel block 2 kk + g of its pixel = K-step kk's B operand
          const auto h0 = __builtin_amdgcn_permlane32_swap(pkh[2 * kk][0], pkh[2 * kk + 1][0], false, false);
          const auto h1 = __builtin_amdgcn_permlane32_swap(pkh[2 * kk][1], pkh[2 * kk + 1][1], false, false);
          const auto l0 = __builtin_amdgcn_permlane32_swap(pkl[2 * kk][0], pkl[2 * kk + 1][0], false, false);
          const auto l1 = __builtin_amdgcn_permlane32_swap(pkl[2 * kk][1], pkl[2 * kk + 1][1], false, false);
          const uint4 sh = uint4{h0[0], h1[0], h0[1], h1[1]}, sl = uint4{l0[0], l1[0], l0[1], l1[1]};
          const half8 xh = *reinterpret_cast<const half8*>(&sh), xl = *reinterpret_cast<const half8*>(&sl);
          p0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hd_h[HEADP ? kk : 0], xh, p0, 0, 0, 0);
          p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hd_l[HEADP ? kk : 0], xh, p1, 0, 0, 0);
          p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hd_h[HEADP ? kk : 0], xl, p1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) fin[r] = p0[r] + p1[r] * kSplitInv;      // row = tap (r & 3) + 8 (r >> 2) + 4 gh
      }
      const int e_seg = pset * 2 + khalf;
      const int e_y = c_ty * 8 + e_seg * 2 + pr;
      const int e_x = c_tx * 16 + pc;
      d_in = e_y < a.Ho && e_x < a.Wo;
      if (OUTSLOT) {
        unsigned du;
        const unsigned n = div_dl.divmod((unsigned)c_img, du);
        const size_t P = (size_t)c_img + n + 1;
        d_base = reinterpret_cast<char*>(a.out) + P * 8 * phw * 16;
        d_off = ((unsigned)(e_y + 1) * (unsigned)g.PW + (unsigned)(e_x + 1)) * 16u + gh * 8u;
      } else {
        d_base = reinterpret_cast<char*>(a.out + (size_t)c_img * OCH * plane_o);
        d_off = (unsigned)e_y * (unsigned)a.Wo + (unsigned)e_x;
      }
    }
    SN_AGG_STAMP(5);
    lds_barrier();                      // partial buffer free
    SN_AGG_STAMP(6);
    cur ^= 1;
    toff = ntoff;
    c_img = n_img;
    c_ty = n_ty;
    c_tx = n_tx;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) flush(q);
  SN_AGG_STAMP_WG(1);
}

// ------------------------------------------------------------------------------------------
// The 5x5 stride-2 down-convs 1..3 of the fp16 modes (k_conv_x3s<5, 2, 32, 4, 32, 32, ...>) in the same style: the input
// is a zero-bordered split-slot tensor (two zero pixels around the image, the grid rounded up to whole tiles), staging
// is LDS-DMA into a double-buffered halo tile, the epilogue is deferred.  The 4 x 32 tile of k_conv_x3s needs 101 KB of
// LDS and cannot be double buffered, so a round here is a 4 x 16 tile (55 KB: two 2 x 16 segments, one per wave pair
// member): wave (khalf, pset) accumulates K half `khalf` of segment `pset` — 25 K-steps x 3 MFMAs, B fragments three
// K-steps ahead — and the two members of a pair take turns finishing the segment (round parity), so every wave ships
// one partial and finishes one segment per two rounds.  K order and the three MFMAs per K-step are those of k_conv_x3s:
// the same bits.
// ------------------------------------------------------------------------------------------
struct SlotGeom {      // split-slot tensor [img][4 blocks][hi | lo][PH][PW]; pixel (y, x) at row y + py, column x + px
  int PH, PW, py, px;
};

struct DownDma {
  static constexpr int TR = 4, TC = 16, ROWS_IN = 11, COLS_IN = 35, HALF = 20, PITCH = 40, PLANE = ROWS_IN * PITCH;
  static constexpr int NCB = 4, HCB = 2, NK = 25;
  static constexpr int NSL = 2 * NCB * PLANE;                   // LDS slots of a halo tile (parity-split columns): 3520
  static constexpr int NINST = (NSL + 63) / 64;                 // 55
  static constexpr int KW = (NINST + 3) / 4;                    // DMA instructions per wave: 14
  static constexpr int BUF = KW * 4 * 64;
  static constexpr int RED_FLOATS = 4 * 16 * 64;
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF * 16 + (size_t)RED_FLOATS * 4 + 32 * 4 + 16;
  // padded input grid for an Ho x Wo output: two zero rows above the image; EIGHT zero columns to its left (two are read)
  // and a row pitch that is a multiple of 16 slots, so that the producer's 512-byte runs stay 128-byte aligned — with a
  // two-column border and an odd pitch k_down0_f16, which is bound by its 59 MB per pair of stores, took 25 % longer
  static constexpr int PADY = 2, PADX = 8;
  __host__ __device__ static int ph(int Ho) { return (Ho + TR - 1) / TR * (2 * TR) + 3; }
  __host__ __device__ static int pw(int Wo) { return (Wo + TC - 1) / TC * (2 * TC) + 16; }
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
  static_assert((2 * PITCH) % 16 == 0, "the two rows of a segment land on disjoint banks");
};

__global__ __launch_bounds__(256, 1) void k_down_x3s_dma(ConvArgs a, const uint4* __restrict__ vin, SlotGeom gi, SlotGeom go) {
  using T = DownDma;
  constexpr int BUF = T::BUF, KW = T::KW;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  float* s_red = reinterpret_cast<float*>(smem4 + 2 * BUF);
  float* s_bias = s_red + T::RED_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = wave >> 1, pset = wave & 1;
  const int gh = lane >> 5, j = lane & 31;

  half8 wh[T::NK], wl[T::NK];
  {
    const uint4* wsrc = reinterpret_cast<const uint4*>(a.wpk) + (size_t)khalf * T::NK * 2 * 64 + lane;
#pragma unroll
    for (int k = 0; k < T::NK; ++k) {
      const uint4 x = wsrc[(2 * k) * 64], y = wsrc[(2 * k + 1) * 64];
      wh[k] = *reinterpret_cast<const half8*>(&x);
      wl[k] = *reinterpret_cast<const half8*>(&y);
    }
#pragma unroll
    for (int k = 0; k < T::NK; ++k) asm volatile("" : "+a"(wh[k]), "+a"(wl[k]));
  }
  const int pr = j / 16, pc = j % 16;
  const int srow = pset * 2 + pr;                      // output row of the tile this lane computes
  const int lane_base = (khalf * T::HCB + gh) * T::PLANE + 2 * srow * T::PITCH + pc;
  if (tid < kC) s_bias[tid] = a.bias[tid];

  const unsigned iphw = (unsigned)(gi.PH * gi.PW), ophw = (unsigned)(go.PH * go.PW);
  unsigned srel[KW];
#pragma unroll
  for (int e = 0; e < KW; ++e) {
    const int L = (4 * e + wave) * 64 + lane;
    const int part = L / (T::NCB * T::PLANE);
    const int rem = L - part * (T::NCB * T::PLANE);
    const int vb = rem / T::PLANE;
    const int rc = rem - vb * T::PLANE;
    const int r = rc / T::PITCH, di = rc - r * T::PITCH;
    const int cc = di < T::HALF ? 2 * di : 2 * (di - T::HALF) + 1;       // LDS keeps even and odd columns apart
    srel[e] = (L < T::NSL && cc < T::COLS_IN) ? (unsigned)(((vb * 2 + part) * (int)iphw + r * gi.PW + cc) * 16) : 0u;
  }
  const FastDiv div_tx((unsigned)a.tiles_x), div_ty((unsigned)a.tiles_y);
  auto tile_off = [&](int tile, int& img, int& ty, int& tx) {
    unsigned txu, tyu;
    const unsigned t2 = div_tx.divmod((unsigned)tile, txu);
    img = (int)div_ty.divmod(t2, tyu);
    ty = (int)tyu;
    tx = (int)txu;
    return (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(((unsigned)img * 8u * iphw + (unsigned)(ty * 2 * T::TR * gi.PW + tx * 2 * T::TC + (T::PADX - 2))) * 16u));
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

  const float slope = a.lrelu ? kSlope : 1.0f;
  float fin[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) fin[r] = 0.f;
  bool d_in = false;
  int pending = 0;
  char* d_base = reinterpret_cast<char*>(a.out);
  unsigned d_off = 0;
  auto flush = [&](int q) {
    half4 hh, hl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = fin[4 * q + e];
      v = fmaxf(v, v * slope);
      const _Float16 hi = (_Float16)v;
      hh[e] = hi;
      hl[e] = (_Float16)((v - (float)hi) * kSplitScale);
    }
    if (d_in) {
      char* oq = d_base + (size_t)(2 * q) * ophw * 16;
      __builtin_nontemporal_store(hh, reinterpret_cast<half4*>(oq + d_off));
      __builtin_nontemporal_store(hl, reinterpret_cast<half4*>(oq + (size_t)ophw * 16 + d_off));
    }
  };
  auto koff_of = [&](int k) {
    const int ky = k / 5, kx = k - ky * 5;
    return ky * T::PITCH + (kx & 1) * T::HALF + (kx >> 1);
  };

  int cur = 0;
  constexpr int PD = 3;                   // B fragments in flight: one wave per SIMD, nobody else hides the LDS latency
  uint4 bh[PD + 1], bl[PD + 1];
  auto prefetch = [&](int buf) {          // the first PD fragments of a tile: issued as soon as the tile has landed
    const uint4* xh_ = smem4 + buf * BUF + lane_base;
    const uint4* xl_ = xh_ + T::NCB * T::PLANE;
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      bh[d] = xh_[koff_of(d)];
      bl[d] = xl_[koff_of(d)];
    }
  };
  prefetch(0);
  for (int it = 0; tile < t_end; tile += nlb, ++it) {
    const int nxt = tile + nlb;
    const int more = __builtin_amdgcn_readfirstlane(nxt < t_end ? 1 : 0);
    int n_img = c_img, n_ty = c_ty, n_tx = c_tx;
    unsigned ntoff = toff;
    const uint4* s_xh = smem4 + cur * BUF + lane_base;
    const uint4* s_xl = s_xh + T::NCB * T::PLANE;

    f32x16 acc0, acc1, zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
    for (int k = 0; k < T::NK; ++k) {
      if (k + PD < T::NK) {
        bh[(k + PD) % (PD + 1)] = s_xh[koff_of(k + PD)];
        bl[(k + PD) % (PD + 1)] = s_xl[koff_of(k + PD)];
      }
      __builtin_amdgcn_sched_barrier(0);
      const half8 xh = *reinterpret_cast<const half8*>(&bh[k % (PD + 1)]);
      const half8 xl = *reinterpret_cast<const half8*>(&bl[k % (PD + 1)]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xh, k == 0 ? zero : acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xh, k == 0 ? zero : acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xl, acc1, 0, 0, 0);
      // the next tile's coordinates are worked out behind the first MFMAs, its DMA starts one K-step later
      if (k == 0 && more) ntoff = tile_off(nxt, n_img, n_ty, n_tx);
      if (k >= 1 && k <= KW) dma(k - 1, cur ^ 1, ntoff);
      if (k >= 2 && k < 6 && pending) flush(k - 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    pending = 0;
    const int fin_role = ((it & 1) == khalf) ? 1 : 0;        // wave-uniform: this round's finisher of the pair
    if (!fin_role) {
      float* dst = s_red + (size_t)wave * 16 * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = acc0[r] + acc1[r] * kSplitInv;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile has landed
    // ONE barrier per round: partials visible, every wave done with buffer `cur`.  The partial buffer needs no second
    // one: a wave writes its region every other round only (the roles alternate), and the barrier of the round in
    // between orders its partner's read before that write.
    lds_barrier();
    prefetch(cur ^ 1);
    if (fin_role) {
      const float* src = s_red + (size_t)(wave ^ 2) * 16 * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) fin[r] = acc0[r] + acc1[r] * kSplitInv + src[r * 64] + s_bias[(r & 3) + 8 * (r >> 2) + 4 * gh];
      const int e_y = c_ty * T::TR + srow;
      const int e_x = c_tx * T::TC + pc;
      d_in = e_y < a.Ho && e_x < a.Wo;
      d_base = reinterpret_cast<char*>(a.out) + (size_t)c_img * 8 * ophw * 16;
      d_off = ((unsigned)(e_y + go.py) * (unsigned)go.PW + (unsigned)(e_x + go.px)) * 16u + gh * 8u;
      pending = 1;
    }
    cur ^= 1;
    toff = ntoff;
    c_img = n_img;
    c_ty = n_ty;
    c_tx = n_tx;
  }
  if (pending) {
#pragma unroll
    for (int q = 0; q < 4; ++q) flush(q);
  }
}

}  // namespace sn
