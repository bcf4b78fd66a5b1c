// sn_tower_f32.hpp — the 32 -> 32 3x3 (dilated) convolutions of the refinement tower in SN_PREC_FP32 (gfx950).
//
// Exact fp32 on v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, the fp32 vector rate: 157.3 TFLOP/s), i.e. 16x the matrix
// time of the fp16 tower at 2x its bytes: arithmetic intensity 58 FLOP/B against a ridge of ~20 — purely matrix-pipe
// bound.  BASELINE configs[1] ("single pair ... fp32") runs through this path; behind DnnNode::Run
// (stereonet_infer/src/stereonet_node.cpp:812) like the rest of the network.
//
// The generic implicit-GEMM kernel k_conv_c32_mfma (which still runs every other fp32 layer) reached 0.34 of that peak
// on these layers: it re-reads its weights from LDS per tile, stages through registers with address arithmetic per
// element and keeps one wave per SIMD.  Here, for the plain fp32 NCHW tensors of the tower:
//   * WEIGHTS-STATIONARY in LDS: the 144 A operands of every lane (9 taps x 16 channel pairs; lane (co, kh) holds
//     w[co][2 kk + kh][tap]) sit in a 36 KB table [operand][lane] for the life of a persistent workgroup — one
//     conflict-free ds_read_b32 per (tap, channel pair), shared by the wave's two segments.  (Kept in registers they
//     left hipcc 250+ VGPRs with scratch reloads inside the MFMA loop: 0.37 of the peak.)
//   * the B operand of lane (pixel, kh) is ONE ds_read_b32 of the staged input tile [channel][row][column]
//     (channel 2 kk + kh): 256 contiguous bytes per half-wave, conflict-free, immediate offsets only;
//   * a tile (8 rows x 64 columns, 16 segments of 32 pixels, 2 per wave of a 512-thread workgroup) is 288 MFMAs = 18.4 k
//     matrix cycles per wave; the input is staged in channel phases of CPH channels (42 KB at dilation 1) so that TWO
//     workgroups = four waves per SIMD fit a CU:
//     while one stages its next phase (global -> registers -> LDS, bounds-checked: these tensors have no border), the
//     other one's eight waves keep the matrix pipes busy — no hand-built ring needed at 4.6 - 9 k cycles per phase;
//   * bias, residual and LeakyReLU in the epilogue, 128-byte coalesced stores.
#pragma once

namespace sn {

template <int DIL_, int CPH_>
struct F32Tile {
  static constexpr int DIL = DIL_, CPH = CPH_;           // channels staged per phase
  static constexpr int TH = 8, TW = 64;                  // 16 segments of 32 pixels: wave w of the 8 owns row w
  static constexpr int LP = (DIL + 3) / 4 * 4;           // halo columns staged left and right: whole 16-byte groups
  static constexpr int ROWS = TH + 2 * DIL, COLS = TW + 2 * LP;
  static constexpr int GPR = COLS / 4;                   // 16-byte groups per row
  static constexpr int PLANE = ROWS * COLS;
  static constexpr int NGRP = CPH * ROWS * GPR;          // groups per phase
  static constexpr int NPH = 32 / CPH;
  static constexpr int TILE_BYTES = (NGRP + 63) / 64 * 1024;             // whole 1 KiB DMA instructions
  static constexpr int W_BYTES = 144 * 64 * 4 + 128;                     // weight table [operand][lane] + the 32 biases
  static constexpr int LDS_BYTES = W_BYTES + TILE_BYTES;
  static_assert(CPH % 2 == 0 && 32 % CPH == 0, "phases of whole channel pairs");
  static_assert(TILE_BYTES < 65536, "ds_read_b32 immediate offsets");
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

template <int DIL, int CPH, bool RES>
__global__ __launch_bounds__(512, 4) void k_ref_conv_f32(const float* __restrict__ in, float* out, const float* res,
                                                         const float* __restrict__ wpk,      // [ci][tap][co]
                                                         const float* __restrict__ bias, int nimg, int H, int W, int lrelu) {
  using T = F32Tile<DIL, CPH>;
  extern __shared__ __attribute__((aligned(16))) float ldsf[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 31, kh = lane >> 5;

  float* const wtab = ldsf;                              // [kk * 9 + tap][lane]
  float* const tile = ldsf + T::W_BYTES / 4;
  for (int i = tid; i < 144 * 64; i += 512) {
    const int m = i >> 6, l = i & 63, kk = m / 9, tap = m - kk * 9;
    wtab[i] = wpk[((2 * kk + (l >> 5)) * 9 + tap) * kC + (l & 31)];
  }
  float* const s_bias = wtab + 144 * 64;
  if (tid < kC) s_bias[tid] = bias[tid];
  const int tiles_x = (W + T::TW - 1) / T::TW, tiles_y = (H + T::TH - 1) / T::TH;
  const int per_img = tiles_x * tiles_y, total = per_img * nimg;
  const size_t HW = (size_t)H * W;
  const float slope = lrelu ? kSlope : 1.0f;
  // B operand: channel 2 kk + kh, this wave's row, column px (+ segment, tap as immediates)
  const float* lane_b = tile + kh * T::PLANE + wave * T::COLS + (T::LP - DIL) + px;
  const int lane_q = lane / T::GPR, lane_r = lane - lane_q * T::GPR;      // the lane's part of a DMA group index
  const float* lane_a = wtab + lane;

  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int img = t / per_img, rem = t - img * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int y0 = ty * T::TH, x0 = tx * T::TW;
    const float* src = in + (size_t)img * kC * HW;
    const bool edge = y0 - DIL < 0 || x0 - T::LP < 0 || y0 + T::TH + DIL > H || x0 + T::TW + T::LP > W;     // uniform
    f32x16 acc[2];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < T::NPH; ++p) {
      __syncthreads();                   // everyone is done reading the previous phase
      // Stage CPH channels of the (TH + 2 DIL) x (TW + 2 LP) window by LDS-DMA, 16 bytes per lane (glds16: no data
      // registers, asynchronous; a register-staged loop ran its loads one latency after the other and made this kernel
      // slower than the generic one).  The window starts LP = 4 or 8 columns left of the tile so that every group of four
      // floats is 16-byte aligned in memory (W is a multiple of 16) and lies wholly inside or wholly outside the image.
      // Group g = 64 k + lane of the [channel][row][group] image: the 64 k part is decomposed on the scalar unit, the
      // lane part (lane / GPR, lane % GPR) once per kernel — 1.3 k VALU instructions per tile went into per-element
      // divisions before, 4.4 per MFMA, on a pipe the fp32 MFMA shares.  Addresses are clamped into the image; tiles
      // that touch the image edge then overwrite the out-of-image groups with the zero padding in a second pass.
      {
        const float* pbase = src + (size_t)(p * T::CPH) * HW;
        constexpr int NDMA = (T::NGRP + 63) / 64;
        for (int k = wave; k < NDMA; k += 8) {
          const int b = k * 64;                                   // uniform
          const int cb = b / (T::ROWS * T::GPR), rb2 = b - cb * (T::ROWS * T::GPR);
          const int rb = rb2 / T::GPR, gb = rb2 - rb * T::GPR;
          int gc = gb + lane_r, r = rb + lane_q, c = cb;
          if (gc >= T::GPR) { gc -= T::GPR; ++r; }
          if (r >= T::ROWS) { r -= T::ROWS; ++c; }
          c = c < T::CPH ? c : T::CPH - 1;                        // tail instruction: re-fetch into the pad behind the tile
          int gy = y0 - DIL + r, gx = x0 - T::LP + 4 * gc;
          if (edge) {
            gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= W ? W - 4 : gx);
          }
          const unsigned off = (unsigned)c * (unsigned)(HW * 4) + ((unsigned)gy * (unsigned)W + (unsigned)gx) * 4u;
          glds16(lds_addr(tile) + (unsigned)k * 1024u, off, pbase);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (edge) {
          __syncthreads();
          for (int i = tid; i < T::NGRP; i += 512) {
            const int c = i / (T::ROWS * T::GPR), r2 = i - c * (T::ROWS * T::GPR);
            const int r = r2 / T::GPR, gc = r2 - r * T::GPR;
            const int gy = y0 - DIL + r, gx = x0 - T::LP + 4 * gc;
            if (!((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W))
              reinterpret_cast<float4*>(tile)[i] = float4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int kl = 0; kl < T::CPH / 2; ++kl)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap - ky * 3;
          const float w = lane_a[((p * (T::CPH / 2) + kl) * 9 + tap) * 64];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            // segment s of row `wave`
            const float b = lane_b[(2 * kl) * T::PLANE + (ky * DIL) * T::COLS + s * 32 + kx * DIL];
            if (p == 0 && kl == 0 && tap == 0) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, zero, 0, 0, 0);
            else acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, acc[s], 0, 0, 0);
          }
        }
    }
    // epilogue: accumulator register r of lane (px, kh) = channel (r & 3) + 8 (r >> 2) + 4 kh of pixel px.  Stored as it
    // lies that is sixteen 4-byte stores (and residual loads) per segment; the VMEM instructions of the epilogue cost
    // 13 % of the kernel (measured with the traffic switched off).  So the values cross a 2 KB per-wave LDS scratch
    // (the tile area, free after the last phase) sixteen channels at a time and leave as 16-byte accesses: lane l then
    // owns pixels 4 (l & 7) .. + 3 of channel l >> 3 — four stores per segment, the same 128-byte runs per channel.
    __syncthreads();                     // every wave is done reading the tile
    {
      float* scr = tile + wave * 512;    // [16 channels][32 pixels]
      const int y = y0 + wave;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int xg = x0 + s * 32 + 4 * (lane & 7);
        const bool ok = y < H && xg < W;                 // W % 4 == 0: a group of four pixels is inside or outside
#pragma unroll
        for (int half = 0; half < 2; ++half) {           // channels 16 half .. 16 half + 15
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8) {
            const int r = half * 8 + r8;
            const int cl = (r & 3) + 8 * ((r >> 2) & 1) + 4 * kh;        // channel within the half
            scr[cl * 32 + px] = acc[s][r];
          }
          // The lanes of a wave exchange values through LDS here without a barrier (the LDS queue of a wave is in order).
          // To the compiler that is a thread whose stores nobody reads: where `ok` differs between lanes it made the
          // stores of the not-ok lanes conditional (dead-store elimination against the next pass's stores) and the
          // pixels those lanes own came out stale — only on tiles cut by the right image edge.  The clobbers pin the
          // stores before and the loads after this point.
          asm volatile("" ::: "memory");
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int cl = (lane >> 3) + 8 * i, co = half * 16 + cl;
            f32x4 v = *reinterpret_cast<const f32x4*>(scr + cl * 32 + 4 * (lane & 7));
            const size_t o = ((size_t)img * kC + co) * HW + (size_t)y * W + xg;
            const float bco = s_bias[co];
            f32x4 rv = {0.f, 0.f, 0.f, 0.f};
            if (RES && ok) rv = *reinterpret_cast<const f32x4*>(res + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float u = v[e] + rv[e] + bco, tt = u * slope;
              v[e] = u > tt ? u : tt;
            }
            if (ok) *reinterpret_cast<f32x4*>(out + o) = v;
            asm volatile("" ::: "memory");
          }
        }
      }
    }
  }
}

}  // namespace sn
