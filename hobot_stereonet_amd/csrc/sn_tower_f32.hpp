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

// Development only (scripts/f32_tower_timing.hip -DSN_F32_TIMING): s_memtime stamps of the first 64 steps of wave 0 of
// workgroup 17: sn_f32_stamps[step * 8 + k].
#ifdef SN_F32_TIMING
__device__ unsigned long long sn_f32_stamps[64 * 8 + 8];
#define SN_F32_STAMP(k)                                                                      \
  do {                                                                                       \
    if (blockIdx.x == 17 && wave == 0 && stamp_step < 64) {                                  \
      const unsigned long long t_ = __builtin_readcyclecounter();                            \
      if (lane == 0) sn_f32_stamps[stamp_step * 8 + (k)] = t_;                               \
    }                                                                                        \
  } while (0)
#else
#define SN_F32_STAMP(k) do { } while (0)
#endif

namespace sn {

// TWV = columns of a work unit: 64 (a whole tile, two segments per wave) or 32 (a HALF tile, one segment per wave); NW = waves
// of the workgroup = rows of a tile.
template <int DIL_, int CPH_, int TWV_ = 64, int NW_ = 8>
struct F32Tile {
  static constexpr int DIL = DIL_, CPH = CPH_, NW = NW_;  // channels staged per phase
  static constexpr int TH = NW_, TW = TWV_;               // segments of 32 pixels: wave w owns row w
  static constexpr int NSEG = TW / 32;
  static constexpr int LP = (DIL + 3) / 4 * 4;           // halo columns staged left and right: whole 16-byte groups
  static constexpr int ROWS = TH + 2 * DIL, COLS = TW + 2 * LP;
  static constexpr int GPR = COLS / 4;                   // 16-byte groups per row
  static constexpr int PLANE = ROWS * COLS;
  static constexpr int NGRP = CPH * ROWS * GPR;          // groups per phase
  static constexpr int NDMA = (NGRP + 63) / 64;          // 1 KiB LDS-DMA instructions per phase
  static constexpr int NPH = 32 / CPH;
  static constexpr int BUF_BYTES = NDMA * 1024;                          // one staging buffer (whole DMA instructions)
  static constexpr int W_BYTES = 144 * 64 * 4 + 128;                     // weight table [operand][lane] + the 32 biases
  static constexpr int SCR_BYTES = NW * 2048;                            // the epilogue's per-wave exchange scratch
  static constexpr int LDS_BYTES = W_BYTES + 2 * BUF_BYTES + SCR_BYTES;
  static_assert(CPH % 2 == 0 && 32 % CPH == 0 && NPH % 2 == 0, "phases of whole channel pairs, an even number of them");
  static_assert(BUF_BYTES < 65536, "ds_read_b32 immediate offsets");
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
  static_assert(GPR >= 8 && ROWS >= 8, "one carry per digit in the DMA group decomposition");
};

// Work distribution.  A launch has `total` TH x 64 tiles = 2 total HALF tiles (TH x 32); workgroup b of G owns the contiguous
// half-tile range [floor(b U / G), floor((b + 1) U / G)), U = 2 total, and walks it as whole tiles wherever both halves of a
// tile are its own, as single halves at the ends.  Why halves: a tile is 288 MFMAs per wave = 8 us of a CU's matrix pipes, and
// 1280x720 is 1800 tiles on 256 CUs = 7.03 per CU: in whole tiles every launch lasts 8 rounds (12 % of the tower idle); in
// halves 14.06 -> 15.
//
// Pipeline (round 6).  Through round 5 a workgroup staged a phase, waited for it, and ran its MFMAs, with a second workgroup
// per CU meant to fill the gaps.  Measured (scripts/f32_tower_probe.py, profiles/r06_f32_tower_probe.txt): ONE workgroup per
// CU 205 us per layer, TWO 189 us — two workgroups that start together and take the same time per phase stay in step, stage
// together and compute together.  Now ONE workgroup per CU double-buffers its own staging: the DMA of step s + 1 (the next
// channel phase, or phase 0 of the NEXT unit) is issued right after the barrier that opens step s and lands under step s's
// 72 .. 144 MFMAs per wave; one barrier per step.  The epilogue of a unit runs at the start of the next unit's phase 0 (after
// that step's DMA issue, before its first MFMA re-initialises the accumulators), so the stores are a whole MFMA phase old
// when the next s_waitcnt vmcnt(0) comes.
template <int DIL, int CPH, bool RES, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void k_ref_conv_f32(const float* __restrict__ in, float* out, const float* res,
                                                         const float* __restrict__ wpk,      // [ci][tap][co]
                                                         const float* __restrict__ bias, int nimg, int H, int W, int lrelu) {
  using T = F32Tile<DIL, CPH, 64, NW>;
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float ldsf[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 31, kh = lane >> 5;

  float* const wtab = ldsf;                              // [kk * 9 + tap][lane]
  float* const s_bias = wtab + 144 * 64;
  float* const bufs = ldsf + T::W_BYTES / 4;             // two staging buffers
  float* const scr = bufs + 2 * (T::BUF_BYTES / 4) + wave * 512;      // this wave's [16 channels][32 pixels]
  const int tiles_x = (W + T::TW - 1) / T::TW, tiles_y = (H + T::TH - 1) / T::TH;
  const int per_img = tiles_x * tiles_y, total = per_img * nimg;
  const size_t HW = (size_t)H * W;
  const float slope = lrelu ? kSlope : 1.0f;
  const float* lane_a = wtab + lane;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  struct Unit {
    int img, y0, x0, w64, valid;         // all wave-uniform
  };
  const long U2 = 2L * total;
  int u = (int)((long)blockIdx.x * U2 / gridDim.x);
  const int u1 = (int)(((long)blockIdx.x + 1) * U2 / gridDim.x);
  auto next_unit = [&]() __attribute__((always_inline)) {
    Unit r{0, 0, 0, 0, 0};
    while (u < u1) {
      const int t = u >> 1;
      const int img = t / per_img, rem = t - img * per_img;
      const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
      const int y0 = ty * T::TH, x0 = tx * T::TW;
      if (!(u & 1) && u + 1 < u1) {
        u += 2;
        r = Unit{img, y0, x0, 1, 1};
        break;
      }
      const int xh = x0 + 32 * (u & 1);
      u += 1;
      if (xh < W) {                       // (the right half of a tile cut by the image edge is empty)
        r = Unit{img, y0, xh, 0, 1};
        break;
      }
    }
    return r;
  };

  // group index i = 64 k + lane of the [channel][row][16-byte group] image of a phase -> (c, r, gc); the 64 k part is
  // decomposed on the scalar unit, the lane part once per call (per-element divisions were 4.4 VALU instructions per MFMA
  // on a pipe the fp32 MFMA shares)
#define SN_F32_DECOMP(U_, k_, lq_, lr_, c_, r_, gc_)                                  \
  {                                                                                   \
    const int b_ = (k_) * 64;                                                          \
    const int cb_ = b_ / (U_::ROWS * U_::GPR), rb2_ = b_ - cb_ * (U_::ROWS * U_::GPR); \
    const int rb_ = rb2_ / U_::GPR, gb_ = rb2_ - rb_ * U_::GPR;                        \
    gc_ = gb_ + (lr_);                                                                 \
    r_ = rb_ + (lq_);                                                                  \
    c_ = cb_;                                                                          \
    if (gc_ >= U_::GPR) { gc_ -= U_::GPR; ++r_; }                                      \
    if (r_ >= U_::ROWS) { r_ -= U_::ROWS; ++c_; }                                      \
  }
  auto unit_edge = [&](auto twv, const Unit& un) __attribute__((always_inline)) {
    using U = F32Tile<DIL, CPH, decltype(twv)::value, NW>;
    return un.y0 - DIL < 0 || un.x0 - U::LP < 0 || un.y0 + U::TH + DIL > H || un.x0 + U::TW + U::LP > W;     // uniform
  };
  // Staging: CPH channels (one phase) of the (TH + 2 DIL) x (TW + 2 LP) window of a unit go to a buffer by LDS-DMA, 16 bytes
  // per lane (no data registers, asynchronous).  The window starts LP = 4 or 8 columns left of the unit so that every group of
  // four floats is 16-byte aligned in memory (W % 4 == 0) and lies wholly inside or wholly outside the image.  Addresses are
  // clamped into the image; units that touch the image edge overwrite the out-of-image groups with zeros afterwards (fill).
  // This wave's instructions are k = wave + NW j, j < KWMAX; their per-lane byte offsets inside a channel phase are computed
  // ONCE per unit (offsets), and the instructions themselves are issued BETWEEN the MFMA groups of the step before (dma_one):
  // issued in a block after the step's barrier they cost every wave about 1.1 k cycles per step with the matrix pipe idle
  // (eight waves queueing at the address unit, scripts/f32_tower_timing.hip); behind an MFMA the issue is free.
  constexpr int KWMAX = (T::NDMA + NW - 1) / NW;
  auto offsets = [&](auto twv, const Unit& un, unsigned (&voff)[KWMAX]) __attribute__((always_inline)) {
    using U = F32Tile<DIL, CPH, decltype(twv)::value, NW>;
    const int lane_q = lane / U::GPR, lane_r = lane - lane_q * U::GPR;
    const bool edge = unit_edge(twv, un);
#pragma unroll
    for (int jj = 0; jj < KWMAX; ++jj) {
      const int k = wave + NW * jj;
      int c, r, gc;
      SN_F32_DECOMP(U, k, lane_q, lane_r, c, r, gc)
      c = c < U::CPH ? c : U::CPH - 1;                        // tail instruction: re-fetch into the pad behind the window
      int gy = un.y0 - DIL + r, gx = un.x0 - U::LP + 4 * gc;
      if (edge) {
        gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
        gx = gx < 0 ? 0 : (gx >= W ? W - 4 : gx);
      }
      voff[jj] = (unsigned)c * (unsigned)(HW * 4) + ((unsigned)gy * (unsigned)W + (unsigned)gx) * 4u;
    }
  };
  auto offsets_any = [&](const Unit& un, unsigned (&voff)[KWMAX]) __attribute__((always_inline)) {
    if (un.w64) offsets(std::integral_constant<int, 64>{}, un, voff);
    else offsets(std::integral_constant<int, 32>{}, un, voff);
  };
  // instruction jj of this wave for phase p of unit `un` (whose offsets are voff) into buffer dst
  auto dma_one = [&](int jj, const Unit& un, int p, float* dst, const unsigned (&voff)[KWMAX]) __attribute__((always_inline)) {
    const int ndma = un.w64 ? F32Tile<DIL, CPH, 64, NW>::NDMA : F32Tile<DIL, CPH, 32, NW>::NDMA;
    const int k = wave + NW * jj;
    if (k < ndma) {                      // uniform
      // (wave-uniform by construction; the unit went through a struct, which hides that from the divergence analysis)
      const size_t pb_ = reinterpret_cast<size_t>(in + ((size_t)un.img * kC + (size_t)p * CPH) * HW);
      const float* pbase = reinterpret_cast<const float*>(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(pb_ >> 32)) << 32) |
                                                          (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pb_));
      glds16((unsigned)__builtin_amdgcn_readfirstlane((int)(lds_addr(dst) + (unsigned)k * 1024u)), voff[jj], pbase);
    }
  };
  // zero padding: every wave rewrites the out-of-image groups of ITS OWN DMA instructions (after its own vmcnt(0), before
  // the step's barrier: no second barrier)
  auto fill = [&](auto twv, const Unit& un, float* dst) __attribute__((always_inline)) {
    using U = F32Tile<DIL, CPH, decltype(twv)::value, NW>;
    const int lane_q = lane / U::GPR, lane_r = lane - lane_q * U::GPR;
    for (int k = wave; k < U::NDMA; k += NW) {
      int c, r, gc;
      SN_F32_DECOMP(U, k, lane_q, lane_r, c, r, gc)
      const int gy = un.y0 - DIL + r, gx = un.x0 - U::LP + 4 * gc;
      if (c < U::CPH && !((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W))
        reinterpret_cast<float4*>(dst)[k * 64 + lane] = float4{0.f, 0.f, 0.f, 0.f};
    }
  };
#undef SN_F32_DECOMP
  // epilogue: accumulator register r of lane (px, kh) = channel (r & 3) + 8 (r >> 2) + 4 kh of pixel px.  Stored as it
  // lies that is sixteen 4-byte stores (and residual loads) per segment; the VMEM instructions of the epilogue cost
  // 13 % of the kernel (measured with the traffic switched off).  So the values cross a 2 KB per-wave LDS scratch
  // sixteen channels at a time and leave as 16-byte accesses: lane l then owns pixels 4 (l & 7) .. + 3 of channel
  // l >> 3 — four stores per segment, the same 128-byte runs per channel.
  auto epilogue = [&](auto nseg, const Unit& un, const f32x16 (&acc)[2]) __attribute__((always_inline)) {
    constexpr int NS = decltype(nseg)::value;
    const int y = un.y0 + wave;
    // residual values: all of the unit's loads in flight before the first exchange (one workgroup per CU has the registers;
    // loaded where they are used, the eight loads of a tile were eight round trips one after the other: +16 us per layer)
    constexpr bool PRE = RES && NW <= 8;
    f32x4 rvs[PRE ? NS : 1][2][2];
    if constexpr (PRE) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int xg = un.x0 + s * 32 + 4 * (lane & 7);
        const bool ok = y < H && xg < W;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int co = half * 16 + (lane >> 3) + 8 * i;
            const size_t o = ((size_t)un.img * kC + co) * HW + (size_t)y * W + xg;
            rvs[s][half][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) rvs[s][half][i] = *reinterpret_cast<const f32x4*>(res + o);
          }
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int xg = un.x0 + s * 32 + 4 * (lane & 7);
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
          const size_t o = ((size_t)un.img * kC + co) * HW + (size_t)y * W + xg;
          const float bco = s_bias[co];
          f32x4 rv = {0.f, 0.f, 0.f, 0.f};
          if constexpr (PRE) rv = rvs[s][half][i];
          else if (RES && ok) rv = *reinterpret_cast<const f32x4*>(res + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float uu = v[e] + rv[e] + bco, tt = uu * slope;
            v[e] = uu > tt ? uu : tt;
          }
          if (ok) *reinterpret_cast<f32x4*>(out + o) = v;
          asm volatile("" ::: "memory");
        }
      }
    }
  };
  auto epilogue_any = [&](const Unit& un, const f32x16 (&acc)[2]) __attribute__((always_inline)) {
    if (un.w64) epilogue(std::integral_constant<int, 2>{}, un, acc);
    else epilogue(std::integral_constant<int, 1>{}, un, acc);
  };

  f32x16 acc[2];
  Unit prev{0, 0, 0, 0, 0};
  [[maybe_unused]] int stamp_step = 0;
  // one unit: NPH steps; step p reads buffer p & 1
  unsigned voff_cur[KWMAX], voff_nxt[KWMAX];
  auto run = [&](auto twv, const Unit& cur, const Unit& nxt) __attribute__((always_inline)) {
    using U = F32Tile<DIL, CPH, decltype(twv)::value, NW>;
    constexpr int G = U::CPH / 2;        // MFMA groups of a step: one channel pair each (9 NSEG MFMAs per wave)
    // B operand: channel 2 kk + kh, this wave's row, column px (+ segment, tap as immediates)
    const int lane_boff = kh * U::PLANE + wave * U::COLS + (U::LP - DIL) + px;
    const bool edge = unit_edge(twv, cur);
#pragma unroll
    for (int p = 0; p < U::NPH; ++p) {
      float* const b = bufs + (p & 1) * (T::BUF_BYTES / 4);
      float* const bn = bufs + ((p + 1) & 1) * (T::BUF_BYTES / 4);
      SN_F32_STAMP(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's part of step p has landed (and its older stores)
      SN_F32_STAMP(1);
      if (edge) fill(twv, cur, b);
      __syncthreads();                   // step p is complete in LDS; everyone is done reading the other buffer
      SN_F32_STAMP(2);
      const bool last = p + 1 == U::NPH;
      if (last && nxt.valid) offsets_any(nxt, voff_nxt);
      SN_F32_STAMP(3);
      if (p == 0 && prev.valid) epilogue_any(prev, acc);
      SN_F32_STAMP(4);
      // the step after this one: the unit's next phase, or phase 0 of the next unit
      auto dma_group = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < KWMAX; ++jj)
          if (jj % G == g) {
            if (!last) dma_one(jj, cur, p + 1, bn, voff_cur);
            else if (nxt.valid) dma_one(jj, nxt, 0, bn, voff_nxt);
          }
      };
      const float* lane_b = b + lane_boff;
      if constexpr (NW <= 8) {
        // Operands of channel pair kl + 1 (9 A values, 9 NSEG B values) are fetched while the MFMAs of pair kl run: left to
        // itself hipcc put every ds_read right in front of the MFMA that consumes it and waited for it (lgkmcnt(0) every
        // second MFMA).
        float wv[2][9], bv[2][9][U::NSEG];
        auto fetch = [&](int kl, int slot) __attribute__((always_inline)) {
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            wv[slot][tap] = lane_a[((p * G + kl) * 9 + tap) * 64];
#pragma unroll
            for (int s = 0; s < U::NSEG; ++s)
              bv[slot][tap][s] = lane_b[(2 * kl) * U::PLANE + (ky * DIL) * U::COLS + s * 32 + kx * DIL];       // segment s of row `wave`
          }
        };
        fetch(0, 0);
#pragma unroll
        for (int kl = 0; kl < G; ++kl) {
          if (kl + 1 < G) fetch(kl + 1, (kl + 1) & 1);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int s = 0; s < U::NSEG; ++s) {
              if (p == 0 && kl == 0 && tap == 0) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[0][0], bv[0][0][s], zero, 0, 0, 0);
              else acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[kl & 1][tap], bv[kl & 1][tap][s], acc[s], 0, 0, 0);
            }
          // issue order for the machine scheduler: (the first group's operands up front,) then the next group's reads spread
          // under this group's MFMAs (0x100 = DS read, 0x008 = MFMA; hipcc merges neighbouring 4-byte reads: upper bounds)
          if (kl == 0) __builtin_amdgcn_sched_group_barrier(0x100, 9 + 9 * U::NSEG, 0);
#pragma unroll
          for (int i = 0; i < 9 * U::NSEG; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, U::NSEG == 2 ? 2 : 3, 0);
          }
          // nothing crosses the end of a group: the DMA instructions stay between the groups, and the next step's
          // s_waitcnt vmcnt(0) (not a memory operation to the scheduler's eyes) cannot rise into the MFMA sequence — it did,
          // and the wave waited for a DMA it had issued a few hundred cycles earlier in the middle of the work meant to cover it
          __builtin_amdgcn_sched_barrier(0);
          dma_group(kl);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // four waves per SIMD (128 VGPRs): the other waves cover a wave's LDS latency; operands read where they are used
#pragma unroll
        for (int kl = 0; kl < G; ++kl) {
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const float w = lane_a[((p * G + kl) * 9 + tap) * 64];
#pragma unroll
            for (int s = 0; s < U::NSEG; ++s) {
              const float bb = lane_b[(2 * kl) * U::PLANE + (ky * DIL) * U::COLS + s * 32 + kx * DIL];
              if (p == 0 && kl == 0 && tap == 0) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, bb, zero, 0, 0, 0);
              else acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, bb, acc[s], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          dma_group(kl);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      SN_F32_STAMP(5);
#ifdef SN_F32_TIMING
      ++stamp_step;
#endif
    }
  };

  Unit cur = next_unit();
  if (!cur.valid) return;                // uniform
#ifdef SN_F32_TIMING
  if (blockIdx.x == 17 && tid == 0) {
    sn_f32_stamps[64 * 8 + 0] = __builtin_readcyclecounter();
    sn_f32_stamps[64 * 8 + 1] = wall_clock64();
  }
#endif
  offsets_any(cur, voff_cur);
#pragma unroll
  for (int jj = 0; jj < KWMAX; ++jj) dma_one(jj, cur, 0, bufs, voff_cur);
  // the weight table [kk * 9 + tap][lane] and the biases, under the first DMA (the first step's barrier publishes them)
  for (int i = tid; i < 144 * 64; i += NT) {
    const int m = i >> 6, l = i & 63, kk = m / 9, tap = m - kk * 9;
    wtab[i] = wpk[((2 * kk + (l >> 5)) * 9 + tap) * kC + (l & 31)];
  }
  if (tid < kC) s_bias[tid] = bias[tid];
  while (cur.valid) {
    const Unit nxt = next_unit();
    if (cur.w64) run(std::integral_constant<int, 64>{}, cur, nxt);
    else run(std::integral_constant<int, 32>{}, cur, nxt);
    prev = cur;
    cur = nxt;
#pragma unroll
    for (int jj = 0; jj < KWMAX; ++jj) voff_cur[jj] = voff_nxt[jj];
  }
  epilogue_any(prev, acc);
#ifdef SN_F32_TIMING
  if (blockIdx.x == 17 && tid == 0) {
    sn_f32_stamps[64 * 8 + 2] = __builtin_readcyclecounter();
    sn_f32_stamps[64 * 8 + 3] = wall_clock64();
  }
#endif
}

}  // namespace sn
