// sn_down01.hpp — the first TWO down-convs of the feature tower as ONE 13x13 stride-4 convolution (fp16 modes).
//
// SN-K4's down-convs have no activation between them (oracle/stereonet_oracle.c so_features), so
//   down1(down0(x))[o, oy, ox] = b1[o] + sum_{c, ky1, kx1} W1[o, c, ky1, kx1] * D0[c, 2 oy - 2 + ky1, 2 ox - 2 + kx1]
//   D0[c, r, s]                = b0[c] + sum_{i, ky0, kx0} W0[c, i, ky0, kx0] * X[i, 2 r - 2 + ky0, 2 s - 2 + kx0]
// is linear in the int8 image X:
//   = beff[o] + sum_{i, u, v} Weff[o, i, u, v] * X[i, 4 oy - 6 + u, 4 ox - 6 + v],   u = 2 ky1 + ky0, v = 2 kx1 + kx0 in 0..12
// — the conv-conv fold every inference compiler does next to the BatchNorm fold.  What it buys here (per 32 quarter-
// resolution pixels): 78 fp16 MFMAs instead of 150 (down-conv 1 on split operands, three per product) + 4 x 16 (down-conv
// 0), because K shrinks from 800 + 4 x 75 to 3 x 13 x 13 = 507 and the int8 image is EXACT in fp16, so only the weights
// need the hi/lo split (two MFMAs per product) — and the half-resolution tensor (59 MB per pair written by k_down0_f16
// and read back by k_down_x3s_dma) never exists.
// The one place where the fold is not a plain convolution is down-conv 1's zero padding of D0: its taps that fall outside
// the half-resolution map contribute NOTHING (not b0, not a partial window).  That only concerns the first and last row
// and column of the quarter-resolution map: nine weight classes (row class x column class, each {first, inner, last}),
// folded on the host in double precision.  k_down01_f16 computes the inner class (all but 2 rows and 2 columns of the
// map), k_down01_border the other eight.
#pragma once
#include "sn_kernels.hpp"

#include <vector>

namespace sn {

struct Down01 {
  static constexpr int TR = 8, TC = 32;                   // output tile: 8 rows of one 32-pixel segment
  static constexpr int KW = 13;                           // folded kernel size
  static constexpr int ROWS = 4 * (TR - 1) + KW;          // 41 input rows of a tile
  static constexpr int NDW = (4 * (TC - 1) + 16) / 4;     // 35 staged dwords (140 columns) per row: the window starts 8 columns
                                                          // left of 4 * ox0 so that every B fragment is 8-byte aligned in LDS
  static constexpr int PITCH = 144;                       // halves per LDS row
  static constexpr int BUF = 3 * ROWS * PITCH;            // halves per buffer
  static constexpr int LDS_BYTES = 2 * BUF * 2 + 16;      // + a spare slot: where staging threads without an element write
  static constexpr int NLOAD = (3 * ROWS * NDW + 255) / 256;
  static constexpr int NK = 3 * KW;                       // K-steps: one (channel, window row) each, 16 window columns
  static constexpr int NCLS = 9, INNER = 4;               // weight classes: 3 * row class + column class
  static constexpr int W_AGPR_LO = 21;                    // lo fragments homed in AGPRs (all 39 hi fragments are)
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
  static_assert((PITCH * 2) % 8 == 0 && PITCH >= NDW * 4, "8-byte aligned rows");
};

// ---- host: the fold (double precision) --------------------------------------------------------------------------------
// w0 [32][3][5][5], b0 [32], w1 [32][32][5][5], b1 [32] (PyTorch layouts)  ->  weff [9][32][3][13][13], beff [9][32]
inline void compose_down01(const float* w0, const float* b0, const float* w1, const float* b1, std::vector<double>& weff,
                           std::vector<double>& beff) {
  constexpr int KW = Down01::KW;
  weff.assign((size_t)Down01::NCLS * kC * 3 * KW * KW, 0.0);
  beff.assign((size_t)Down01::NCLS * kC, 0.0);
  for (int cls = 0; cls < Down01::NCLS; ++cls) {
    const int rc = cls / 3, cc = cls % 3;
    // first row / column: D0 rows 2 * 0 - 2 + k < 0 for k < 2; last: 2 (Ho - 1) - 2 + 4 = 2 Ho = one past the map
    const int ky_lo = rc == 0 ? 2 : 0, ky_hi = rc == 2 ? 3 : 4, kx_lo = cc == 0 ? 2 : 0, kx_hi = cc == 2 ? 3 : 4;
    for (int o = 0; o < kC; ++o) {
      double* we = &weff[((size_t)cls * kC + o) * 3 * KW * KW];
      double be = b1[o];
      for (int c = 0; c < kC; ++c)
        for (int ky1 = ky_lo; ky1 <= ky_hi; ++ky1)
          for (int kx1 = kx_lo; kx1 <= kx_hi; ++kx1) {
            const double a = w1[(((size_t)o * kC + c) * 5 + ky1) * 5 + kx1];
            be += a * b0[c];
            for (int i = 0; i < 3; ++i)
              for (int ky0 = 0; ky0 < 5; ++ky0)
                for (int kx0 = 0; kx0 < 5; ++kx0)
                  we[((size_t)i * KW + 2 * ky1 + ky0) * KW + 2 * kx1 + kx0] += a * w0[(((size_t)c * 3 + i) * 5 + ky0) * 5 + kx0];
          }
      beff[(size_t)cls * kC + o] = be;
    }
  }
}

// A fragments [class][K-step t = ci * 13 + u][hi | lo][lane][8]: lane (co, g) holds window columns 8 g + e, i.e. folded
// column v = 8 g + e - 2 (window column 0 is image column 4 ox - 8; columns 0, 1 and 15 carry zero weights)
inline void pack_down01(const std::vector<double>& weff, std::vector<_Float16>& pk) {
  constexpr int KW = Down01::KW;
  pk.assign((size_t)Down01::NCLS * Down01::NK * 2 * 64 * 8, (_Float16)0.f);
  for (int cls = 0; cls < Down01::NCLS; ++cls)
    for (int t = 0; t < Down01::NK; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int co = lane & 31, v = 8 * (lane >> 5) + e - 2;
          if (v < 0 || v >= KW) continue;
          const double w = weff[(((size_t)cls * kC + co) * 3 * KW + t) * KW + v];      // [ci][u] flattened = t
          const _Float16 hi = (_Float16)w;
          const size_t base = ((((size_t)cls * Down01::NK + t) * 2) * 64 + lane) * 8 + e;
          pk[base] = hi;
          pk[base + 64 * 8] = (_Float16)((w - (double)hi) * (double)kSplitScale);
        }
}

// four int8 of a dword -> x / 128 as four fp16, exactly: 0x4800 | (b ^ 0x80) is the fp16 number 8 + (b + 128) / 128
__device__ __forceinline__ half4 i8x4_to_f16(uint32_t v) {
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  const uint32_t u = v ^ 0x80808080u;
  const uint32_t p0 = __builtin_amdgcn_perm(0x48484848u, u, 0x04010400u);
  const uint32_t p1 = __builtin_amdgcn_perm(0x48484848u, u, 0x04030402u);
  const half2v nine = {(_Float16)9.0f, (_Float16)9.0f};
  const half2v a = *reinterpret_cast<const half2v*>(&p0) - nine, b = *reinterpret_cast<const half2v*>(&p1) - nine;
  half4 r;
  r[0] = a[0];
  r[1] = a[1];
  r[2] = b[0];
  r[3] = b[1];
  return r;
}

// hi/lo split of one accumulator quad -> the two 8-byte halves of a split slot
__device__ __forceinline__ void split_store4(const float (&v)[4], char* hi_ptr, char* lo_ptr) {
  half4 hh, hl;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 hi = (_Float16)v[e];
    hh[e] = hi;
    hl[e] = (_Float16)((v[e] - (float)hi) * kSplitScale);
  }
  __builtin_nontemporal_store(hh, reinterpret_cast<half4*>(hi_ptr));
  __builtin_nontemporal_store(hl, reinterpret_cast<half4*>(lo_ptr));
}

// ------------------------------------------------------------------------------------------
// Inner class.  Persistent 256-thread workgroup, one per CU: every wave keeps ALL 78 A fragments (39 K-steps x hi / lo =
// 312 registers, 240 of them in the AGPR half of the file, which an MFMA reads directly) and owns two of the tile's eight
// rows; the int8 window of the next tile is in flight during the MFMAs and is converted to fp16 (x / 128, exact) on its
// way into the other LDS buffer.  B fragment of lane (pixel j, k-group g) at K-step (ci, u): the 8 consecutive window
// columns 4 j + 8 g .. + 7 of window row 4 r + u — 16 bytes at an 8-byte aligned LDS address, consecutive lanes 8 bytes
// apart (conflict-free).
// Output: split-slot tensor [img][4 blocks][hi | lo][go.PH][go.PW], pixel (y, x) at (y + go.py, x + go.px) — the input
// layout of k_down_x3s_dma (or plain: {Ho, Wo, 0, 0}).  Border pixels (first / last row and column) are not stored.
// ------------------------------------------------------------------------------------------
template <bool W4>      // W % 4 == 0: every staged dword is wholly inside or outside its image row
__global__ __launch_bounds__(256, 1) void k_down01_f16(const int8_t* __restrict__ in6, int H, int W,
                                                    const uint4* __restrict__ wfrag,   // inner class: [39][hi|lo][64]
                                                    const float* __restrict__ bias,    // inner class: [32]
                                                    uint4* __restrict__ out, int Ho, int Wo, int tiles_x, int tiles_y,
                                                    int nimg, int oPH, int oPW, int opy, int opx) {
  using T = Down01;
  extern __shared__ __attribute__((aligned(16))) _Float16 s_x[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, g = lane >> 5;

  half8 wh[T::NK], wl[T::NK];
#pragma unroll
  for (int t = 0; t < T::NK; ++t) {
    const uint4 a = wfrag[(2 * t) * 64 + lane], b = wfrag[(2 * t + 1) * 64 + lane];
    wh[t] = *reinterpret_cast<const half8*>(&a);
    wl[t] = *reinterpret_cast<const half8*>(&b);
  }
#pragma unroll
  for (int t = 0; t < T::NK; ++t) {
    asm volatile("" : "+a"(wh[t]));
    if (t < T::W_AGPR_LO) asm volatile("" : "+a"(wl[t]));
  }
  f32x16 bv, zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    bv[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * g];
    zero[r] = 0.f;
  }

  // staging: dword idx = e * 256 + tid of the [3][ROWS][NDW] window; (channel, row, dword) never change, so they are
  // decoded once into one packed register per round: bits 0..15 LDS byte offset, 16..21 row, 22..27 dword, 28..29 channel,
  // 31 = a real element
  unsigned pk[T::NLOAD];
#pragma unroll
  for (int e = 0; e < T::NLOAD; ++e) {
    const int idx = e * 256 + tid;
    const int c = idx / (T::ROWS * T::NDW);
    const int rem = idx - c * (T::ROWS * T::NDW);
    const int r = rem / T::NDW, q = rem - r * T::NDW;
    const unsigned lo = (unsigned)(((c * T::ROWS + r) * T::PITCH + 4 * q) * 2);
    pk[e] = idx < 3 * T::ROWS * T::NDW ? (lo | (unsigned)r << 16 | (unsigned)q << 22 | (unsigned)c << 28 | 1u << 31) : 0u;
  }
  static_assert(T::BUF * 2 < 65536 && T::ROWS < 64 && T::NDW < 64, "packed staging table");

  const int total = tiles_x * tiles_y * nimg;
  const FastDiv div_tx((unsigned)tiles_x), div_ty((unsigned)tiles_y);
  auto decode = [&](int tile, int& img, int& ty, int& tx) {
    unsigned txu, tyu;
    const unsigned t2 = div_tx.divmod((unsigned)tile, txu);
    img = (int)div_ty.divmod(t2, tyu);
    ty = (int)tyu;
    tx = (int)txu;
  };
  // Staging pipeline, all of it inside the MFMA loop of a tile (one wave per SIMD: whatever is not interleaved with the
  // MFMAs is exposed): even K-steps convert one dword of the window of tile i+1 (in registers since the previous tile) into
  // the other LDS buffer, odd K-steps request that dword of tile i+2 (17 per thread).  Loads are unconditional:
  // an element outside the image (27 % of the tiles touch an edge at 1280 x 720) reads the image's first dword instead and
  // is zeroed when it is converted (`zmask`); that needs W % 4 == 0, so that a dword is wholly inside or outside a row.
  // Otherwise (W = 1242) interior tiles take the same path unchecked and edge tiles a checked one behind the loop.
  uint32_t pre[T::NLOAD];
  unsigned zmask = 0;                       // bit e: element e of `pre` lies outside the image
  constexpr bool w4 = W4;                   // (a straight-line loop body: hipcc then counts the loads in flight across
                                            // the back edge exactly instead of waiting for all of them, stores included)
  struct Win {                              // uniform: the window of the tile being fetched
    const int8_t* base;                     // the eye's three planes
    int iy0, xs;
    bool nochk, slow;
  };
  auto fetch_setup = [&](int tile) -> Win {
    int img, ty, tx;
    decode(tile, img, ty, tx);
    const int n = img >> 1, eye = img & 1;
    Win w;
    w.iy0 = 4 * ty * T::TR - 6;
    w.xs = 4 * tx * T::TC - 8;                                                 // window origin (xs % 4 == 0)
    w.base = in6 + ((size_t)n * 6 + eye * 3) * H * (size_t)W;
    const bool inside = w.iy0 >= 0 && w.iy0 + T::ROWS <= H && w.xs >= 0 && w.xs + 4 * T::NDW <= W;
    w.nochk = inside;
    w.slow = !w4 && !inside;
    return w;
  };
  // -> bit e of the zero mask
  // `lz` = a zero the compiler cannot see through, made once per tile: pk[e] | lz keeps the unpacking of the staging table
  // inside the tile loop (hoisted, the 4 x 17 unpacked values spill) without a volatile asm per element, which the
  // machine scheduler would not move anything across
  auto fetch_one = [&](const Win& w, int e, unsigned lz) -> unsigned {
    const unsigned p = pk[e] | lz;
    const int r = (p >> 16) & 63, q = (p >> 22) & 63, c = (p >> 28) & 3;
    const int y = w.iy0 + r, x = w.xs + 4 * q;
    // (rounds past the window's last dword re-read its first one; commit_one() sends them to a spare slot)
    const bool ok = w.nochk | ((unsigned)y < (unsigned)H && x >= 0 && x < W);
    const unsigned off = ok ? (unsigned)((c * H + y) * W + x) : 0u;             // < 2^32: one eye's three planes
    __builtin_memcpy(&pre[e], w.base + off, 4);
    return ok ? 0u : 1u << e;
  };
  auto fetch_slow = [&](const Win& w) {     // checked, byte-wise where a dword straddles the right edge; waits for its loads
#pragma unroll
    for (int e = 0; e < T::NLOAD; ++e) {
      unsigned p = pk[e];
      asm volatile("" : "+v"(p));
      const int r = (p >> 16) & 63, q = (p >> 22) & 63, c = (p >> 28) & 3;
      const int y = w.iy0 + r, x = w.xs + 4 * q;
      uint32_t v = 0;
      if ((p >> 31) && (unsigned)y < (unsigned)H) {
        const int8_t* const row = w.base + ((size_t)c * H + y) * (size_t)W;
        if (x >= 0 && x + 3 < W) {
          __builtin_memcpy(&v, row + x, 4);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if ((unsigned)(x + k) < (unsigned)W) v |= (uint32_t)(uint8_t)row[x + k] << (8 * k);
        }
      }
      pre[e] = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing of the checked path stays in flight
  };
  // one staged dword -> four halves -> LDS; only the last round has threads without an element
  auto commit_one = [&](_Float16* buf, int e, unsigned zm, unsigned lz) {
    const unsigned p = pk[e] | lz;
    const unsigned off = (e < T::NLOAD - 1 || (p >> 31)) ? (p & 0xffffu) : (unsigned)(2 * T::BUF * 2);      // spare slot behind the buffers
    const uint32_t v = (zm >> e) & 1u ? 0u : pre[e];
    *reinterpret_cast<half4*>(reinterpret_cast<char*>(buf) + off) = i8x4_to_f16(v);
  };
  static_assert(T::NLOAD * 256 - 3 * T::ROWS * T::NDW < 256, "only the last staging round is partial");
  auto opaque_zero = []() {
    unsigned z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
  };
  auto fetch_all = [&](int tile) -> unsigned {
    const Win w = fetch_setup(tile);
    const unsigned lz = opaque_zero();
    unsigned zm = 0;
    if (!W4 && w.slow) {
      fetch_slow(w);
    } else {
#pragma unroll
      for (int e = 0; e < T::NLOAD; ++e) zm |= fetch_one(w, e, lz);
    }
    return zm;
  };

  // XCD-aware persistent schedule (as k_conv_x3s): each XCD walks its own contiguous band of tiles
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
  const int t_end = (int)((long)(xcd + 1) * total / 8);
  int tile = (int)((long)xcd * total / 8) + lb;
  if (tile >= t_end) return;
  zmask = fetch_all(tile);
#pragma unroll
  for (int e = 0; e < T::NLOAD; ++e) commit_one(s_x, e, zmask, opaque_zero());
  zmask = fetch_all(tile + nlb < t_end ? tile + nlb : tile);          // the window the first iteration converts
  __syncthreads();
  int cur = 0;

  const size_t plane_b = (size_t)oPH * oPW * 16;               // bytes of one (block, part) plane
  const unsigned lane_lds = (unsigned)((4 * (2 * wave) * T::PITCH + 4 * j + 8 * g) * 2);
  const unsigned io_lane = (unsigned)j * 16u + (unsigned)g * 8u;
  constexpr int PD = 2;                                         // K-steps of B fragments in flight
  static_assert(2 * T::NLOAD <= T::NK, "one conversion and one request every other K-step");
  auto koff = [](int t) { return ((t / T::KW) * T::ROWS + (t % T::KW)) * T::PITCH * 2; };

  for (; tile < t_end; tile += nlb) {
    const int nxt2 = tile + 2 * nlb;
    const Win w2 = fetch_setup(nxt2 < t_end ? nxt2 : tile);     // no such tile: this one again (harmless, no branch per K-step)
    const bool slow2 = w2.slow;                                 // uniform
    unsigned zm2 = 0;
    const unsigned lz = opaque_zero();
    const char* bufc = reinterpret_cast<const char*>(s_x + cur * T::BUF) + lane_lds;
    auto bfrag = [&](int t, int s) {
      const uint2* p = reinterpret_cast<const uint2*>(bufc + koff(t) + s * 4 * T::PITCH * 2);
      const uint2 a = p[0], b = p[1];
      uint4 x = make_uint4(a.x, a.y, b.x, b.y);
      return x;
    };
    uint4 xb[PD + 1][2];
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      xb[d][0] = bfrag(d, 0);
      xb[d][1] = bfrag(d, 1);
    }
    f32x16 acc0[2], acc1[2];
#pragma unroll
    for (int t = 0; t < T::NK; ++t) {
      if (t + PD < T::NK) {
        xb[(t + PD) % (PD + 1)][0] = bfrag(t + PD, 0);
        xb[(t + PD) % (PD + 1)][1] = bfrag(t + PD, 1);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const half8 x = *reinterpret_cast<const half8*>(&xb[t % (PD + 1)][s]);
        acc0[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], x, t == 0 ? bv : acc0[s], 0, 0, 0);
        acc1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t], x, t == 0 ? zero : acc1[s], 0, 0, 0);
      }
      // element e: converted (tile i+1 -> the other buffer) at K-step 2 e, requested again (tile i+2 -> the same register)
      // at K-step 2 e + 1 — every load has a whole tile's MFMAs to land, the VALU work is spread evenly over the loop
      if ((t & 1) == 0 && t / 2 < T::NLOAD) commit_one(s_x + (cur ^ 1) * T::BUF, t / 2, zmask, lz);
      if ((t & 1) == 1 && t / 2 < T::NLOAD && (W4 || !slow2)) zm2 |= fetch_one(w2, t / 2, lz);
      // Issue order inside the K-step (in-order issue, one wave per SIMD: an MFMA behind an MFMA waits for the matrix pipe,
      // and whatever stands behind THAT waits with it): one MFMA, then a share of the step's other instructions, four
      // times — the staging arithmetic runs in the 32 cycles each MFMA occupies the pipe instead of after all four.
      // (0x008 MFMA, 0x100 DS read, 0x006 VALU | SALU, 0x230 VMEM read | DS write)
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x006, 7, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x006, 7, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x230, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      int img, ty, tx;
      decode(tile, img, ty, tx);
      const int x0 = tx * T::TC, ox = x0 + j;
      char* const tbase = reinterpret_cast<char*>(out) + (size_t)img * 8 * plane_b;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int oy = ty * T::TR + 2 * wave + s;
        const bool ok = oy >= 1 && oy < Ho - 1 && ox >= 1 && ox < Wo - 1;
        char* const rowp = tbase + ((size_t)(oy + opy) * oPW + x0 + opx) * 16 + io_lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc0[s][4 * q + e] + acc1[s][4 * q + e] * kSplitInv;
          if (ok) split_store4(v, rowp + (size_t)(2 * q) * plane_b, rowp + (size_t)(2 * q + 1) * plane_b);
        }
      }
    }
    if (!W4 && slow2) fetch_slow(w2);                           // (W % 4 != 0 only) an edge window, behind the loop
    zmask = zm2;
    lds_barrier();
    cur ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// The eight border classes: first / last row and column of the quarter-resolution map (1.7 % of it at 1280 x 720).  One
// WORKGROUP of three waves per run of up to 32 pixels of one class, wave c takes input channel c (13 of the 39 K-steps); A
// fragments streamed from L2 (80 KB per class), B fragments gathered from the int8 image with per-byte bounds tests, the
// same two MFMAs per K-step as the inner kernel; the three partial sums meet in LDS, channel 0 + channel 1 + channel 2.
// (Rounds 5: one wave per run walked the three channels one after the other — three dependent request phases, and in the
// column classes every lane of a load touches its own image row: 29 us for a single pair, 1.7x the inner kernel that does
// 60x the work; profiles/r06_f16_b1_kernel_summary.txt.)
// ------------------------------------------------------------------------------------------
template <bool W4>      // W % 4 == 0 (as k_down01_f16): straight-line unconditional loads; otherwise the checked byte-wise form
__global__ __launch_bounds__(192) void k_down01_border(const int8_t* __restrict__ in6, int H, int W,
                                                       const uint4* __restrict__ wfrag,   // [9][39][hi|lo][64]
                                                       const float* __restrict__ bias,    // [9][32]
                                                       uint4* __restrict__ out, int Ho, int Wo, int nimg, int oPH, int oPW,
                                                       int opy, int opx) {
  using T = Down01;
  const int lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
  const int nT = (Wo - 2 + 31) / 32, nL = (Ho - 2 + 31) / 32, per_img = 4 + 2 * nT + 2 * nL;
  __shared__ float s_part[2][64][64];              // partial sums of waves 1, 2: [wave - 1][eye, accumulator, register][lane]
  const int wid = (int)blockIdx.x;                 // one run per workgroup
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // = input channel of this wave
  // the run of pixels is taken in BOTH eyes of a pair (nimg = 2 pairs): each A fragment fetched from L2 then serves two
  const int n = wid / per_img;                     // MFMAs per weight half
  int ch = wid - n * per_img;
  int cls, oy, ox;
  bool valid = true;
  if (ch < 4) {                                    // corners
    cls = (ch >> 1) * 6 + (ch & 1) * 2;
    oy = (ch >> 1) ? Ho - 1 : 0;
    ox = (ch & 1) ? Wo - 1 : 0;
    valid = j == 0;
  } else if ((ch -= 4) < 2 * nT) {                 // first / last row without the corners
    const int bottom = ch >= nT ? 1 : 0;
    const int p = (ch - bottom * nT) * 32 + j;
    cls = bottom ? 7 : 1;
    oy = bottom ? Ho - 1 : 0;
    ox = 1 + p;
    valid = p < Wo - 2;
  } else {                                         // first / last column without the corners
    ch -= 2 * nT;
    const int right = ch >= nL ? 1 : 0;
    const int p = (ch - right * nL) * 32 + j;
    cls = right ? 5 : 3;
    ox = right ? Wo - 1 : 0;
    oy = 1 + p;
    valid = p < Ho - 2;
  }
  if (!valid) {                                    // parked lanes compute a harmless pixel
    oy = 0;
    ox = 0;
  }
  const int8_t* const base = in6 + (size_t)n * 6 * H * (size_t)W;        // eye e: planes 3 e .. 3 e + 2
  const uint4* wsrc = wfrag + (size_t)cls * T::NK * 2 * 64 + lane;
  f32x16 acc0[2], acc1[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc0[0][r] = acc0[1][r] = wave == 0 ? bias[cls * kC + (r & 3) + 8 * (r >> 2) + 4 * g] : 0.f;
    acc1[0][r] = acc1[1][r] = 0.f;
  }
  const int xw = 4 * ox - 8 + 8 * g;
  // one input channel (13 K-steps) at a time: its 26 window dwords and 26 A fragments are requested together, so the wave
  // pays the memory latency three times instead of 39
  // W % 4 == 0 (uniform): a dword is wholly inside or outside its row, so the load is unconditional (an outside one reads
  // the plane's first dword) and the selection happens when the value is used — no branch, nothing waits inside the
  // request phase.  Otherwise the checked, byte-wise form.
  constexpr bool w4 = W4;
  auto load4 = [&](const int8_t* row, int x, bool row_ok, bool& ok) -> uint32_t {      // window columns x .. x + 3 of one image row
    uint32_t v = 0;
    if constexpr (w4) {
      ok = row_ok && x >= 0 && x < W;
      __builtin_memcpy(&v, ok ? row + x : base, 4);
      return v;
    }
    ok = true;
    if (!row_ok) return 0u;
    if (x >= 0 && x + 3 < W) {
      __builtin_memcpy(&v, row + x, 4);
    } else if (x + 3 >= 0 && x < W) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((unsigned)(x + k) < (unsigned)W) v |= (uint32_t)(uint8_t)row[x + k] << (8 * k);
    }
    return v;
  };
  {
    const int c = wave;
    uint32_t v0[2][T::KW], v1[2][T::KW];
    uint4 fa[T::KW], fb[T::KW];
    unsigned keep0 = 0, keep1 = 0;                 // (the same for both eyes: same pixel, same window)
#pragma unroll
    for (int u = 0; u < T::KW; ++u) {
      const int y = 4 * oy - 6 + u;
      const bool row_ok = (unsigned)y < (unsigned)H;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int8_t* row = base + ((size_t)(3 * e + c) * H + (row_ok ? y : 0)) * (size_t)W;
        bool k0, k1;
        v0[e][u] = load4(row, xw, row_ok, k0);
        v1[e][u] = load4(row, xw + 4, row_ok, k1);
        keep0 |= k0 ? 1u << u : 0u;
        keep1 |= k1 ? 1u << u : 0u;
      }
      fa[u] = wsrc[(2 * (c * T::KW + u)) * 64];
      fb[u] = wsrc[(2 * (c * T::KW + u) + 1) * 64];
    }
#pragma unroll
    for (int u = 0; u < T::KW; ++u)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const half4 a = i8x4_to_f16((keep0 >> u) & 1u ? v0[e][u] : 0u), b = i8x4_to_f16((keep1 >> u) & 1u ? v1[e][u] : 0u);
        half8 x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          x[k] = a[k];
          x[4 + k] = b[k];
        }
        acc0[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(&fa[u]), x, acc0[e], 0, 0, 0);
        acc1[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(&fb[u]), x, acc1[e], 0, 0, 0);
      }
  }
  if (wave > 0) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s_part[wave - 1][(e * 2 + 0) * 16 + r][lane] = acc0[e][r];
        s_part[wave - 1][(e * 2 + 1) * 16 + r][lane] = acc1[e][r];
      }
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[e][r] = (acc0[e][r] + s_part[0][(e * 2 + 0) * 16 + r][lane]) + s_part[1][(e * 2 + 0) * 16 + r][lane];
      acc1[e][r] = (acc1[e][r] + s_part[0][(e * 2 + 1) * 16 + r][lane]) + s_part[1][(e * 2 + 1) * 16 + r][lane];
    }
  if (valid) {
    const size_t plane_b = (size_t)oPH * oPW * 16;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      char* const p = reinterpret_cast<char*>(out) + (size_t)(2 * n + e) * 8 * plane_b + ((size_t)(oy + opy) * oPW + ox + opx) * 16 + g * 8;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = acc0[e][4 * q + k] + acc1[e][4 * q + k] * kSplitInv;
        split_store4(v, p + (size_t)(2 * q) * plane_b, p + (size_t)(2 * q + 1) * plane_b);
      }
    }
  }
}

}  // namespace sn
