// sn_stream_block.hpp — fused residual block of the fp16 refinement tower, ROW-STREAMING form (gfx950).
//
//   y = lrelu(x + conv2(lrelu(conv1(x) + b1)) + b2)        3x3, dilation DIL, 32 -> 32 -> 32 channels
//
// replaces the two k_ref_conv_f16_v2 launches of a block (DnnNode::Run's network, the one call site
// stereonet_infer/src/stereonet_node.cpp:812; layer semantics DESIGN.md §2).  The intermediate t never leaves LDS
// and x is read from memory ONCE: 118 MB per pair and block instead of 295 MB.
//
// Shape of the computation (what makes it different from the tile-fused kernel k_ref_block_f16_h, which recomputes a
// halo ring of t per 6 x 62 tile and pays 2.4x the matrix work of one conv):
//   * a workgroup walks DOWN a vertical strip of OW = TW - 2 DIL output columns, R rows per step, as a line-buffer
//     pipeline: the x rows of a step arrive by LDS-DMA into a ring, conv1 turns them into R rows of t in a second
//     ring, conv2 (one step behind) turns t rows into y rows.  No row of t is ever computed twice; the only
//     recomputation is the 2 DIL halo columns of a strip (64/62 at dilation 1) and two rows per restart.
//   * a dilated block (both convs dilation DIL) separates into DIL independent ROW PHASES: rows y = py (mod DIL) form a
//     sub-image on which the block is a dilation-1 block vertically; the horizontal dilation stays an LDS column
//     offset.  The kernel therefore streams "sub-rows" with a global row stride of DIL * Ws.
//   * ONE 512-thread workgroup per CU, two waves per SIMD with different jobs: waves 0-3 run conv1 (they hold only
//     conv1's 18 weight fragments) and issue the LDS-DMA groups of the x ring; waves 4-7 run conv2 (conv2's weights)
//     and store y straight from the accumulators as 16-byte slots (v_permlane32_swap half exchange).  A SIMD always has
//     one wave of each kind, so one wave's epilogue / VMEM issue sits beside the other's MFMAs; per super-step a workgroup
//     issues 17 DMA and 16 store instructions for 288 MFMAs (the per-layer kernel: 24 + 16..32 for 72), which is what
//     bounded the earlier tower kernels
//     (DESIGN.md §5).  (A first form with four MFMA waves holding both weight sets and four helper waves owning all
//     VMEM traffic measured 116 us per block: with 244 VGPRs hipcc kept only three B fragments in flight, and a lone
//     wave on a SIMD has nobody to cover its LDS latency — 72 cycles per MFMA.)
//
// Pipeline, one workgroup barrier per super-step q (all rings indexed by the workgroup's running slot number q):
//     conv1 waves:  DMA group of slot q+2 -> x ring;  conv1 of slot q (x groups q-1, q -> t slot q)
//     conv2 waves:  epilogue of slot q-2 (accumulators kept across the barrier, residual from x groups q-3, q-2,
//                   stores);  conv2 MFMAs of slot q-1 (t slots q-2, q-1)
// so conv2's epilogue runs while the SIMD's conv1 wave is in its MFMAs and vice versa.  Ring depths follow from
// that: x ring NXS = 6 groups (q-3 .. q+2), t ring 3 slots.
// A work unit = (image, row phase, strip, sub-rows [v0, v1)); it occupies n + 1 slots, n = ceil((v1 - v0 + 2) / R):
// slot 0 only pre-loads the two x rows above the unit's first t row, slot j >= 1 computes t rows v0 - 1 + R (j - 1) ..
// and y rows v0 - 2 + R (j - 1) .. (the first two are junk and never stored).  Units follow each other in the slot
// stream without draining the pipeline.  The flattened (strip-phase, sub-row) sequence is cut into equal contiguous
// shares, one per workgroup, so the grid is balanced to within one step whatever the geometry.
#pragma once

// Development only (scripts/stream_block_probe.hip -DSN_STREAM_TIMING; `dump` is otherwise unused): s_memtime stamps of super-steps 8..23 of
// workgroup 17 into `dump` (wave w, stamp k of super-step q at u64 index ((w * 16 + q - 8) * 8 + k)).
#ifdef SN_STREAM_TIMING
#define SN_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (blockIdx.x == 17 && q >= 8 && q < 24) {                                                       \
      const unsigned long long t_ = __builtin_readcyclecounter();                                     \
      if (lane == 0) reinterpret_cast<unsigned long long*>(dump + 64)[(wave * 16 + q - 8) * 8 + (k)] = t_; \
    }                                                                                                 \
  } while (0)
// whole-workgroup stamps: u64 index 1152 + 4 * blockIdx.x + {0: clock64 at start, 1: at end, 2 / 3: wall_clock64}
#define SN_STAMP_WG(k)                                                                                          \
  do {                                                                                                          \
    if (threadIdx.x == 0) {                                                                                     \
      unsigned long long* d_ = reinterpret_cast<unsigned long long*>(dump + 64) + 1152 + 4 * blockIdx.x + (k);  \
      d_[0] = __builtin_readcyclecounter();                                                                     \
      d_[2] = wall_clock64();                                                                                   \
    }                                                                                                           \
  } while (0)
#else
#define SN_STAMP(k) do { } while (0)
#define SN_STAMP_WG(k) do { } while (0)
#endif

namespace sn {

// HEAD_ = true: the LAST block of the tower with the refinement head folded in (see "Tail form" below).
template <int DIL_, int TW_ = 64, int R_ = 4, int NXS_ = 6, int NWR_ = 4, bool HEAD_ = false>
struct StreamTile {
  static constexpr bool HEAD = HEAD_;
  // HEAD: conv2's waves fetch the residual rows of a slot one super-step early (into registers), so the x ring needs
  // groups q-2 .. q+2 only
  static constexpr int DIL = DIL_, TW = TW_, R = R_, NXS = NXS_, NTS = 3, PF = NXS_ - (HEAD_ ? 3 : 4);
  static constexpr int NWR = NWR_;                         // waves per role (conv1 / conv2): 4 or 8
  static constexpr int HS = HEAD_ ? 1 : 0;                 // the head needs one more y row / column on every side
  static constexpr int YW = TW - 2 * DIL;                  // valid y columns of a strip
  static constexpr int OW = YW - 2 * HS;                   // output columns of a strip = strip pitch
  static constexpr int X0OFF = -HS;                        // image column of y column 0 of strip 0
  static constexpr int LAG = HEAD_ ? 3 : 2;                // super-steps the last pipeline stage runs behind the slot stream
  static constexpr int PROWS = 2 * R + 2;                  // HEAD: row ring of the head's partial sums P[row][tap][TW] (fp32)
  static constexpr int PRING_BYTES = HEAD_ ? PROWS * 9 * TW * 4 : 0;
  static constexpr int WIN_BYTES = HEAD_ ? NWR_ * 2 * 64 * 4 : 0;      // HEAD: per conv1 wave, two 64-column windows of the low-resolution map
  static constexpr int XW = TW + 2 * DIL;                  // x columns of a strip
  static constexpr int CSEG = TW / 32;                     // 32-pixel MFMA segments per row
  static constexpr int SPW = R * CSEG / NWR;               // segments per wave and conv
  static constexpr int XROW = 4 * XW, TROW = 4 * TW;       // slots per ring row: [channel block][column]
  static constexpr int XGROUP = R * XROW;                  // slots of one DMA group (R rows)
  static constexpr int NINST = (XGROUP + 63) / 64;         // 1 KiB LDS-DMA instructions per group
  static constexpr int KW = (NINST + NWR - 1) / NWR;       // ... per conv1 wave, at most
  static constexpr int XGP = NINST * 64;                   // ring pitch of a group (the last instruction may overshoot)
  static constexpr int TGP = R * TROW;
  static constexpr int XRING = NXS * XGP;
  static constexpr int TRING = NTS * TGP + 64;             // conv2's kx taps of the junk columns run past the last row
  static constexpr int LDS_BYTES = (XRING + TRING) * 16 + 2 * 2 * 16 * 4 + PRING_BYTES + WIN_BYTES;      // + bias tables [conv][k-half][16]
  // image rows a DMA group may touch outside [0, H): a unit's slot 0 starts R sub-rows above its first output row
  // (row -R DIL at v0 = 0) and its last group ends at sub-row hsub + R, i.e. image row <= H + (R + 2) DIL - 2
  static constexpr int ROWS_ABOVE = (R + HS) * DIL, ROWS_BELOW = (R + 2 + HS) * DIL - 1;
  static_assert(!HEAD_ || (DIL_ == 1 && R_ == 4 && TW_ == 64 && NWR_ == 4), "the tail form is written for the dilation-1 strip shape");
  static_assert(SPW == 1 || SPW == 2, "segments per wave");
  static_assert(CSEG % SPW == 0, "a wave's segments lie in one row");
  static_assert(PF == 2, "prefetch distance the counted waits are written for");
  static_assert(R == 4 || R == 2, "rows per step");
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

struct StreamSched {
  int nstrips;       // strips per image row
  int hsub;          // sub-rows per row phase: ceil(H / DIL)
  int total_rows;    // nimg * DIL * nstrips * hsub
  int rows_per_wg;   // share of the flattened sequence per workgroup
};

// Walks the slots of one workgroup's share: units in order, n + 1 slots each.  All members wave-uniform.
template <int R, int HS = 0>
struct StreamIter {
  int f, f_end, hsub;            // cursor (first flat row of the NEXT unit), end of the share
  int sp, v0, v1, n, j, live;    // current unit: strip-phase, output sub-rows [v0, v1), steps; slot j = 0 .. n
  __device__ __forceinline__ void next_unit() {
    if (f >= f_end) { live = 0; return; }
    sp = f / hsub;
    v0 = f - sp * hsub;
    int L = hsub - v0;
    if (L > f_end - f) L = f_end - f;
    v1 = v0 + L;
    n = (L + 2 + 2 * HS + R - 1) / R;
    j = 0;
    f += L;
    live = 1;
  }
  // the iterator of a pipeline stage that runs `lag` super-steps behind the slot stream
  __device__ __forceinline__ void step(int q, int lag, int f0, int f1, int hsub_) {
    if (q < lag) { live = 0; return; }
    if (q == lag) { f = f0; f_end = f1; hsub = hsub_; next_unit(); return; }
    if (live && ++j > n) next_unit();
  }
};

// fp16 pair of the activated values (act_pack2_f16, sn_kernels.hpp: round to fp16, then LeakyReLU on the packed pair).
// fp32 packed math was measured and is NOT the way: v_pk_mul_f32 (and the v_cvt_f32_f16 + v_pk_fma_f32 hipcc then makes of the
// residual add) has the same instruction count as the scalar form and ran 173 instead of 157 us per dilation-1 block.
__device__ __forceinline__ unsigned lrelu_pack2(float a, float b) { return act_pack2_f16(a, b, (_Float16)kSlope); }

// acc + (float)(fp16 half `HI` of the packed pair r): ONE v_fma_mix_f32.  Pinned as inline asm: left to hipcc the
// conversion and the add come out as v_cvt_f32_f16 + v_pk_fma_f32 as soon as the result feeds a packed conversion
// (3 instructions per pair instead of 2, and the packed fp32 instruction is not full rate).
template <int HI>
__device__ __forceinline__ float res_add(unsigned r, float acc) {
  float d;
  if (HI) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(r), "v"(acc));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(r), "v"(acc));
  return d;
}

// 32-bit LDS byte address of a pointer into the workgroup's shared memory (operand of hand-written ds_* instructions)
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

// 36 MFMAs of one conv for this wave's two segments.  rp[ky] = LDS pointer of the row of tap row ky (lane part
// included), COLW = slots per channel block row.  The B fragments are fetched in batches of six (one tap row of one
// channel half: 3 kx x 2 segments) one batch ahead of the MFMAs that consume them — left to itself hipcc used a
// single fragment register set (read, wait, MFMA, read, ...), exposing the full LDS latency 36 times per conv.
template <int DIL, int COLW, int SPW>
__device__ __forceinline__ void stream_conv36(const uint4* const (&rp)[3], const half8 (&wf)[18], const f32x16& bv,
                                              f32x16 (&acc)[SPW]) {
  constexpr int NB = 3 * SPW;            // fragments per batch
  half8 b[2][NB];
  auto fetch = [&](int batch, half8 (&dst)[NB]) {
    const int kk = batch / 3, ky = batch - kk * 3;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int s = 0; s < SPW; ++s)
        dst[kx * SPW + s] = *reinterpret_cast<const half8*>(rp[ky] + (2 * kk * COLW + s * 32 + kx * DIL));
  };
  fetch(0, b[0]);
#pragma unroll
  for (int batch = 0; batch < 6; ++batch) {
    if (batch + 1 < 6) fetch(batch + 1, b[(batch + 1) & 1]);
    const int kk = batch / 3, ky = batch - kk * 3;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int s = 0; s < SPW; ++s) {
        const int tap = ky * 3 + kx;
        if (batch == 0 && kx == 0) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[tap * 2 + kk], b[0][s], bv, 0, 0, 0);
        else acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[tap * 2 + kk], b[batch & 1][kx * SPW + s], acc[s], 0, 0, 0);
      }
  }
  // issue order for the machine scheduler: the bias + the first fragments up front, then one read per MFMA, so that
  // AHEAD LDS reads are always in flight (0x100 = DS read, 0x008 = MFMA)
  constexpr int AHEAD = SPW == 2 ? 8 : 5, NM = 18 * SPW;
  __builtin_amdgcn_sched_group_barrier(0x100, 4 + AHEAD, 0);
#pragma unroll
  for (int i = 0; i < NM - AHEAD; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
}

// LDS-DMA as inline asm: hipcc models __builtin_amdgcn_global_load_lds as a pending LDS write and would put
// s_waitcnt vmcnt(0) in front of LDS reads it can see in the same wave (it did in the helper-wave form of this kernel),
// draining the ring; the asm form is invisible to that bookkeeping and its completion is counted by hand.
// lds_dst = wave-uniform LDS byte address (lane l lands at lds_dst + 16 l), base + voff = each lane's source.
__device__ __forceinline__ void glds16(unsigned lds_dst, unsigned voff, const void* base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_dst), "v"(voff), "s"(base)
      : "memory");
}

// 4-byte form: lane l lands at lds_dst + 4 l
__device__ __forceinline__ void glds4(unsigned lds_dst, unsigned voff, const void* base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %2, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_dst), "v"(voff), "s"(base)
      : "memory");
}

// Arguments of the tail form (HEAD): the refinement head conv 3x3 32 -> 1 + `disp = relu(up + D r)` + wire quantisation,
// exactly k_head_final_f16's arithmetic (same MFMA sequence for P, same order of the nine additions).
struct StreamHeadArgs {
  const float* w;          // [32][9] fp32 head weights
  const float* disp_low;   // [nimg][hl][wl] map the level starts from
  float* out_disp;         // [nimg][H][W] (nullable)
  int32_t* out_raw;        // [nimg][H][W] wire map (nullable)
  float bias, dmax, inv_q;
  int hl, wl, H, W;        // H, W: size of the output maps (<= g.H, g.W)
  UpScale ups;
  unsigned long long* stat = nullptr;   // nullable: sum of |D r| over the written pixels (refine_stat_commit)
};

template <int DIL, int TW, int R, int NXS, int NWR, bool HEAD = false>
__global__ __launch_bounds__(128 * NWR, NWR / 2) __attribute__((amdgpu_waves_per_eu(NWR / 2, NWR / 2))) void k_ref_block_stream_f16(const uint4* __restrict__ xin, uint4* __restrict__ yout,
                                                                const uint4* __restrict__ wfrag1, const float* __restrict__ bias1,
                                                                const uint4* __restrict__ wfrag2, const float* __restrict__ bias2,
                                                                RefGeom g, StreamSched sc, uint4* __restrict__ dump, StreamHeadArgs ha) {
  using T = StreamTile<DIL, TW, R, NXS, NWR, HEAD>;
  constexpr int HS = T::HS;
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  uint4* const xring = lds;
  uint4* const tring = lds + T::XRING;
  float* const s_bias = reinterpret_cast<float*>(tring + T::TRING);      // [conv][k-half][16]
  float* const pring = s_bias + 64;                                       // HEAD: [PROWS][9][TW]
  float* const s_win = pring + T::PRING_BYTES / 4;                        // HEAD: [conv1 wave][2][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, gh = lane >> 5;
  const int role = wave / NWR, rw = wave % NWR;             // role 0: conv1 + DMA, role 1: conv2 + stores

  const int f0 = (int)blockIdx.x * sc.rows_per_wg;
  int f1 = f0 + sc.rows_per_wg;
  if (f1 > sc.total_rows) f1 = sc.total_rows;
  if (f0 >= f1) return;                                     // uniform for the workgroup: before any barrier
  SN_STAMP_WG(0);
  // super-steps of this workgroup: its slots + LAG to drain conv2's MFMAs and epilogue (+ the head's last stage)
  int nss = T::LAG;
  for (int f = f0; f < f1;) {
    const int v0 = f % sc.hsub;
    int L = sc.hsub - v0;
    if (L > f1 - f) L = f1 - f;
    nss += (L + 2 + 2 * HS + R - 1) / R + 1;
    f += L;
  }
  auto decode_sp = [&](int sp, int& img, int& py, int& x0) {
    const int t = sp / sc.nstrips;
    x0 = (sp - t * sc.nstrips) * T::OW + T::X0OFF;
    img = t / DIL;
    py = t - img * DIL;
  };
  const unsigned plane_b = (unsigned)g.Hs * (unsigned)g.Ws * 16u;          // bytes per channel block

  // this wave's 18 weight fragments (conv1 or conv2) and the bias tables
  half8 wf[18];
  {
    const uint4* wsrc = role ? wfrag2 : wfrag1;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const uint4 a = wsrc[i * 64 + lane];
      wf[i] = *reinterpret_cast<const half8*>(&a);
    }
    if (tid < 64) {                       // accumulator register r of k-half g2 holds channel (r & 3) + 8 (r >> 2) + 4 g2
      const int c = tid >> 5, g2 = (tid >> 4) & 1, r = tid & 15;
      s_bias[tid] = (c ? bias2 : bias1)[(r & 3) + 8 * (r >> 2) + 4 * g2];
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) asm volatile("" : "+v"(wf[i]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }



  const int seg0 = rw * T::SPW;
  const int rowW = seg0 / T::CSEG;        // this wave's row of a step (both of its segments lie in it)
  const int cseg0 = seg0 % T::CSEG;

  if (role == 0) {
    // ============================ conv1 waves: DMA + conv1 ============================
    const int lane_x = gh * T::XW + cseg0 * 32 + j;        // B operand: block 2 kk + gh, column cseg0 * 32 + j (+ kx DIL)
    const int lane_w = cseg0 * 32 + j;                     // t write
    // this wave's DMA instructions i = rw, rw + 4, ..: per-lane source offsets relative to the group origin
    constexpr int KWMAX = T::KW;
    const int kw = (T::NINST - rw + NWR - 1) / NWR;        // instructions of this wave (uniform: KWMAX or KWMAX - 1)
    unsigned dma_voff[KWMAX];
#pragma unroll
    for (int k = 0; k < KWMAX; ++k) {
      const int i = rw + NWR * k;
      int s = i * 64 + lane;
      s = s < T::XGROUP ? s : T::XGROUP - 1;               // the overshoot of the last instruction lands in the ring pitch
      const int r = s / T::XROW;
      const int rem = s - r * T::XROW;
      const int blk = rem / T::XW;
      const int c = rem - blk * T::XW;
      dma_voff[k] = (unsigned)blk * plane_b + ((unsigned)(r * DIL) * (unsigned)g.Ws + (unsigned)c) * 16u;
    }
    // signed: the group of a strip's first slot starts up to 2 DIL image rows above row 0 and 2 DIL columns left of column
    // 0 — inside the tensor's zero border for dilation 1 / 2, in the zero slots IN FRONT of the tensor for dilation 4 / 8
    // (ref_front() in the host code) when it is image 0, channel block 0
    long last_base = 0;
    auto dma_issue = [&](const StreamIter<R, HS>& it, int grp) {
      if (it.live) {
        int img, py, x0;
        decode_sp(it.sp, img, py, x0);
        const int row = (it.v0 - HS - R + R * it.j) * DIL + py;          // image row of the group's first row (>= -(R + HS) DIL)
        last_base = (((long)img * 4 * g.Hs + (row + kRefPad)) * (long)g.Ws + (x0 - 2 * DIL + kRefPad)) * 16;
      }
      const char* src = reinterpret_cast<const char*>(xin) + last_base;
      const unsigned dst = lds_addr(xring + grp * T::XGP);
#pragma unroll
      for (int k = 0; k < KWMAX; ++k)
        if (k < kw) glds16(dst + (unsigned)(rw + NWR * k) * 1024u, dma_voff[k], src);
    };
    // `extra` = store instructions younger than the group that must have landed
    auto wait_group = [&](auto extra) {    // all but this wave's youngest DMA group have landed
      constexpr int E = decltype(extra)::value;
      if (kw == KWMAX) wait_vmcnt<KWMAX + E>();
      else wait_vmcnt<KWMAX - 1 + E>();
    };
    StreamIter<R, HS> dm, c1, fin, fin2;
    dm.live = c1.live = fin.live = fin2.live = 0;
    int fin_img = 0, fin_x0 = 0, fin2_img = 0, fin2_x0 = 0;
    // first low-resolution column any lane of a strip's row touches (x0 of upsample_map at the strip's column 0)
    auto win_first_col = [&](int xs) {
      float sx = ((float)xs + 0.5f) * ha.ups.rs - 0.5f;
      sx = sx < 0.f ? 0.f : sx;
      return (int)sx;
    };
    int p4 = 0;                           // HEAD: (4 q) mod PROWS, the P ring position of slot q
    float moved = 0.f;                    // HEAD: this lane's sum of |D r| (refinement statistic)
    dm.step(0, 0, f0, f1, sc.hsub);
    dma_issue(dm, 0);
    dm.step(1, 0, f0, f1, sc.hsub);
    dma_issue(dm, 1);
    wait_group(std::integral_constant<int, 0>{});      // group 0 landed (group 1 may still be in flight)
    block_barrier();                      // + the bias tables

    int c1_py = 0, c1_x0 = 0;
    int qx = 0, qt = 0;                   // q mod NXS, q mod NTS
    for (int q = 0; q < nss; ++q) {
      dm.step(q + 2, 0, f0, f1, sc.hsub);
      c1.step(q, 0, f0, f1, sc.hsub);
      const int do1 = c1.live && c1.j >= 1;
      const int gx0 = qx, gx1 = qx >= 1 ? qx - 1 : qx - 1 + NXS;
      const int gxp = qx + 2 >= NXS ? qx + 2 - NXS : qx + 2;
      SN_STAMP(0);
      if constexpr (HEAD) {
        // ---- last stage of the tail form, slot q-3 (its P rows were written in super-step q-1, the rows above them in
        // q-2): lane c = column c of this wave's row; out(y, x) = relu(up + D (bias + sum of nine shifted P values)).
        // Runs BEFORE this step's DMA group is issued: its two stores are then older than the group and the counted
        // wait below needs no extra term; the upsample taps come through scalar loads (lgkmcnt, not vmcnt).
        fin.step(q, 3, f0, f1, sc.hsub);
        if (fin.live && fin.j >= 1) {
          if (fin.j == 1) {
            int py;
            decode_sp(fin.sp, fin_img, py, fin_x0);
          }
          const int o = fin.v0 - 4 + R * (fin.j - 1) + rowW;                   // output row of this wave
          if (o >= fin.v0 && o < fin.v1 && o < ha.H) {                          // uniform
            const int pb = p4 + 8 >= T::PROWS ? p4 + 8 - T::PROWS : p4 + 8;     // ring row of slot q-3's row 0
            float acc = ha.bias;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              int pr = pb + rowW - 2 + ky;
              pr = pr < 0 ? pr + T::PROWS : (pr >= T::PROWS ? pr - T::PROWS : pr);
              const float* prow = pring + pr * (9 * TW) + lane - 1;
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) acc += prow[(ky * 3 + kx) * TW + kx];
            }
            const int X = fin_x0 + lane;
            // upsample_map's arithmetic on the two row windows the previous super-step staged in LDS (column cb + l at l)
            float sy = ((float)o + 0.5f) * ha.ups.rs - 0.5f;
            float sx = ((float)X + 0.5f) * ha.ups.rs - 0.5f;
            sy = sy < 0.f ? 0.f : sy;
            sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx;
            const int x1 = x0 < ha.wl - 1 ? x0 + 1 : x0;
            const float ly = sy - (float)y0, lx = sx - (float)x0;
            const float hy = 1.0f - ly, hx = 1.0f - lx;
            const int cb = win_first_col(fin_x0);
            const float* w0 = s_win + rw * 128;
            const float v = hy * (hx * w0[x0 - cb] + lx * w0[x1 - cb]) + ly * (hx * w0[64 + x0 - cb] + lx * w0[64 + x1 - cb]);
            const float up = v * ha.ups.mul;
            float d = up + ha.dmax * acc;
            const float mv = fabsf(ha.dmax * acc);
            d = d > 0.f ? d : 0.f;
            if (lane >= 1 && lane <= T::OW && X < ha.W) {
              moved += mv;
              const size_t oi = ((size_t)fin_img * ha.H + o) * ha.W + X;
              if (ha.out_disp) ha.out_disp[oi] = d;
              if (ha.out_raw) ha.out_raw[oi] = (int32_t)__float2int_rn(d * ha.inv_q);
            }
          }
        }
      }
      if constexpr (HEAD) {
        // the two low-resolution row windows of the NEXT super-step's last stage, by 4-byte LDS-DMA: issued before this
        // step's x group, so the counted wait at the end of the step (everything but the youngest group) covers them
        fin2.step(q, 2, f0, f1, sc.hsub);
        if (fin2.live && fin2.j >= 1) {
          if (fin2.j == 1) {
            int py;
            decode_sp(fin2.sp, fin2_img, py, fin2_x0);
          }
          const int o = fin2.v0 - 4 + R * (fin2.j - 1) + rowW;
          if (o >= fin2.v0 && o < fin2.v1 && o < ha.H) {                        // uniform
            float sy = ((float)o + 0.5f) * ha.ups.rs - 0.5f;
            sy = sy < 0.f ? 0.f : sy;
            const int y0 = (int)sy;
            const int y1 = y0 < ha.hl - 1 ? y0 + 1 : y0;
            int c = win_first_col(fin2_x0) + lane;
            c = c < ha.wl - 1 ? c : ha.wl - 1;
            const float* lowp = ha.disp_low + (size_t)fin2_img * ha.hl * ha.wl;
            const unsigned dst = lds_addr(s_win + rw * 128);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // this wave's reads of the windows above are done
            glds4(dst, (unsigned)c * 4u, lowp + (size_t)y0 * ha.wl);
            glds4(dst + 256u, (unsigned)c * 4u, lowp + (size_t)y1 * ha.wl);
          }
        }
      }
      dma_issue(dm, gxp);                 // overwrites group q+2-NXS: last read (residual of slot q-3 / HEAD: q-2) in super-step q-1
      SN_STAMP(1);
      if (do1) {
        if (c1.j == 1) {                  // a new unit: strip origin and row phase (two scalar divisions)
          int img;
          decode_sp(c1.sp, img, c1_py, c1_x0);
        }
        // x rows rowW-2 .. rowW of group q (negative: the last rows of group q-1)
        const uint4* xp[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int rr = rowW - 2 + ky;
          xp[ky] = xring + (rr < 0 ? gx1 : gx0) * T::XGP + (rr & (R - 1)) * T::XROW + lane_x;
        }
        const f32x16 bv = *reinterpret_cast<const f32x16*>(s_bias + gh * 16);
        f32x16 acc[T::SPW];
        stream_conv36<DIL, T::XW, T::SPW>(xp, wf, bv, acc);
        SN_STAMP(2);
        // epilogue: t = lrelu(acc) as fp16, zero outside the image (conv2's zero padding)
        const int trow = (c1.v0 - HS - 1 + R * (c1.j - 1) + rowW) * DIL + c1_py;       // image row of this wave's t row
        const bool row_ok = trow >= 0 && trow < g.H;
        const int tc0 = c1_x0 - DIL;                                                  // image column of t column 0
        const bool interior = row_ok && tc0 >= 0 && tc0 + TW <= g.W;
        uint4* tw = tring + qt * T::TGP + rowW * T::TROW + lane_w;
        // (two bodies selected by ONE uniform branch: with the test inside, every value of the interior path went
        // through a v_cndmask as well)
        auto write_t = [&](auto is_interior) {
#pragma unroll
          for (int s = 0; s < T::SPW; ++s) {
            bool inside = true;
            if (!decltype(is_interior)::value) {
              const int c = tc0 + (cseg0 + s) * 32 + j;
              inside = row_ok && c >= 0 && c < g.W;
            }
            unsigned pk[4][2];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
#pragma unroll
              for (int hp = 0; hp < 2; ++hp) {
                const unsigned u = lrelu_pack2(acc[s][4 * qd + 2 * hp], acc[s][4 * qd + 2 * hp + 1]);
                pk[qd][hp] = (decltype(is_interior)::value || inside) ? u : 0u;
              }
            // four 8-byte half slots per segment at a 16-byte stride.  (Whole 16-byte slots after a v_permlane32_swap half
            // exchange remove this half of the kernel's LDS bank conflicts, but the eight extra VALU instructions per wave
            // and step cost more than the conflicts did: 155.6 vs 152.8 us per dilation-1 block, round 4)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
              *reinterpret_cast<uint2*>(reinterpret_cast<char*>(tw + qd * T::TW + s * 32) + gh * 8) = uint2{pk[qd][0], pk[qd][1]};
          }
        };
        if (interior) write_t(std::true_type{});
        else write_t(std::false_type{});
      }
      SN_STAMP(3);
      wait_group(std::integral_constant<int, 0>{});      // group q+1 landed: younger than it is only this super-step's group
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SN_STAMP(4);
      block_barrier();
      qx = qx + 1 == NXS ? 0 : qx + 1;
      qt = qt + 1 == T::NTS ? 0 : qt + 1;
      p4 = p4 + R >= T::PROWS ? p4 + R - T::PROWS : p4 + R;
    }
    if constexpr (HEAD) refine_stat_commit(ha.stat, moved);
    SN_STAMP_WG(1);
  } else {
    // ============================ conv2 waves: conv2 + residual + stores ============================
    const int lane_t = gh * T::TW + cseg0 * 32 + j;
    const int lane_y = 2 * DIL + cseg0 * 32 + j;           // residual: block qd, 8 bytes at gh * 8
    block_barrier();

    StreamIter<R, HS> c2, ep;
    c2.live = ep.live = 0;
    // HEAD: the head's A fragments as k_head_final_f16 builds them (row i = tap, k = 8 gh + e <-> channel 16 kk + 8 gh + e,
    // fp32 weights split hi / lo), and the residual of the slot whose MFMAs run in this super-step, fetched one step early
    half8 ah[2], al[2];
    uint2 rres[T::SPW][4];
    int p4 = 0;
    if constexpr (HEAD) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float wv = j < 9 ? ha.w[(16 * kk + 8 * gh + e) * 9 + j] : 0.f;
          const _Float16 hi = (_Float16)wv;
          ah[kk][e] = hi;
          al[kk][e] = (_Float16)((wv - (float)hi) * kSplitScale);
        }
#pragma unroll
      for (int s = 0; s < T::SPW; ++s)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) rres[s][qd] = uint2{0u, 0u};
    }
    int ep_img = 0, ep_py = 0, ep_x0 = 0;
    // after the half exchange below lane (j, gh) owns the whole 16-byte slots of channel blocks 2 gh and 2 gh + 1
    const unsigned lane_o = (unsigned)(cseg0 * 32 + j) * 16u + (unsigned)(2 * gh) * plane_b;
    f32x16 acc[T::SPW];
#pragma unroll
    for (int s = 0; s < T::SPW; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    int qx = 0, qt = 0;
    for (int q = 0; q < nss; ++q) {
      c2.step(q, 1, f0, f1, sc.hsub);
      ep.step(q, 2, f0, f1, sc.hsub);
      const int do2 = c2.live && c2.j >= 1;
      const int doe = ep.live && ep.j >= 1;
      const int gx2 = qx >= 2 ? qx - 2 : qx - 2 + NXS, gx3 = qx >= 3 ? qx - 3 : qx - 3 + NXS;
      const int gt1 = qt >= 1 ? qt - 1 : qt - 1 + T::NTS, gt2 = qt >= 2 ? qt - 2 : qt - 2 + T::NTS;
      SN_STAMP(0);
      // ---- epilogue of slot q-2 (its MFMAs ran in super-step q-1, the accumulators crossed the barrier): while this
      // wave is here, the SIMD's conv1 wave has the matrix pipe to itself; y = lrelu(x + acc) -> global memory ----
      if (doe) {
        if (ep.j == 1) decode_sp(ep.sp, ep_img, ep_py, ep_x0);
        const int sub = ep.v0 - HS - 2 + R * (ep.j - 1) + rowW;
        const int row = sub * DIL + ep_py;
        if constexpr (HEAD) {
          // ---- tail form: y (rounded to fp16 exactly as the tensor would have held it, zero outside the image) goes
          // straight into the head's MFMAs as the B operand; P[tap][pixel] -> the P ring; nothing is stored ----
          const bool row_ok = row >= 0 && row < g.H;
          const bool interior = row_ok && ep_x0 >= 0 && ep_x0 + TW <= g.W;      // uniform: no masking at all
          int prow = p4 + 2 + rowW;                            // ring row of slot q-2's row rowW
          prow = prow >= T::PROWS ? prow - T::PROWS : prow;
#pragma unroll
          for (int s = 0; s < T::SPW; ++s) {
            const int col = ep_x0 + (cseg0 + s) * 32 + j;
            const bool ok = row_ok && col >= 0 && col < g.W;
            unsigned pk[4][2];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const half4 rv = *reinterpret_cast<const half4*>(&rres[s][qd]);
              const uint2 rw2 = *reinterpret_cast<const uint2*>(&rv);
              pk[qd][0] = lrelu_pack2(res_add<0>(rw2.x, acc[s][4 * qd]), res_add<1>(rw2.x, acc[s][4 * qd + 1]));
              pk[qd][1] = lrelu_pack2(res_add<0>(rw2.y, acc[s][4 * qd + 2]), res_add<1>(rw2.y, acc[s][4 * qd + 3]));
            }
            if (!interior) {
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                pk[qd][0] = ok ? pk[qd][0] : 0u;
                pk[qd][1] = ok ? pk[qd][1] : 0u;
              }
            }
            // half exchange between channel blocks (0, 1) and (2, 3): lanes gh = 0 end up with the whole slot of block 2 kk,
            // lanes gh = 1 with that of block 2 kk + 1 = the B operand of K-step kk
            f32x16 a0, a1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              a0[r] = 0.f;
              a1[r] = 0.f;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
              const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * kk][0], pk[2 * kk + 1][0], false, false);
              const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * kk][1], pk[2 * kk + 1][1], false, false);
              const uint4 sl = uint4{r0[0], r1[0], r0[1], r1[1]};
              const half8 xh = *reinterpret_cast<const half8*>(&sl);
              a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], xh, a0, 0, 0, 0);
              a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk], xh, a1, 0, 0, 0);
            }
            // accumulator row (r & 3) + 8 (r >> 2) + 4 gh: lanes gh = 0 hold taps 0..3 (r 0..3) and 8 (r 4), gh = 1 taps 4..7
            float* dst = pring + prow * (9 * TW) + (cseg0 + s) * 32 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(4 * gh + r) * TW] = a0[r] + a1[r] * kSplitInv;
            if (gh == 0) dst[8 * TW] = a0[4] + a1[4] * kSplitInv;
          }
        } else
        if (sub >= ep.v0 && sub < ep.v1 && row < g.H) {     // uniform: the unit's first two rows are junk
          const int rr = rowW - 2;
          const uint4* xrow = xring + (rr < 0 ? gx3 : gx2) * T::XGP + (rr & (R - 1)) * T::XROW + lane_y;
          const unsigned ob = (((unsigned)ep_img * 4u * (unsigned)g.Hs + (unsigned)(row + kRefPad)) * (unsigned)g.Ws +
                               (unsigned)(ep_x0 + kRefPad)) * 16u;
#pragma unroll
          for (int s = 0; s < T::SPW; ++s) {
            unsigned pk[4][2];               // [channel block][channels 4 gh + {0,1} | {2,3}] as packed fp16 pairs
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const half4 rv = *reinterpret_cast<const half4*>(reinterpret_cast<const char*>(xrow + qd * T::XW + s * 32) + gh * 8);
              const uint2 rw2 = *reinterpret_cast<const uint2*>(&rv);
              pk[qd][0] = lrelu_pack2(res_add<0>(rw2.x, acc[s][4 * qd]), res_add<1>(rw2.x, acc[s][4 * qd + 1]));
              pk[qd][1] = lrelu_pack2(res_add<0>(rw2.y, acc[s][4 * qd + 2]), res_add<1>(rw2.y, acc[s][4 * qd + 3]));
            }
            // half exchange (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second): blocks
            // (0, 2) and (1, 3) trade halves, so lanes gh = 0 end up with the full slots of blocks 0, 1 and lanes gh = 1
            // with those of blocks 2, 3 -> two 16-byte stores per segment instead of four 8-byte ones (the store tail is
            // bound by the number of store instructions, not by their bytes)
            uint4 sl[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const auto r0 = __builtin_amdgcn_permlane32_swap(pk[k][0], pk[k + 2][0], false, false);
              const auto r1 = __builtin_amdgcn_permlane32_swap(pk[k][1], pk[k + 2][1], false, false);
              sl[k] = uint4{r0[0], r1[0], r0[1], r1[1]};       // [own channels 0-3 | partner's 4-7] of block 2 gh + k
            }
            const int c = (cseg0 + s) * 32 + j;
            if (c < T::OW && ep_x0 + c < g.W) {
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                char* o = reinterpret_cast<char*>(yout) + (ob + (unsigned)k * plane_b + (unsigned)s * 512u);     // uniform
                *reinterpret_cast<uint4*>(o + lane_o) = sl[k];
              }
            }
          }
        }
      }
      SN_STAMP(1);
      // ---- conv2 MFMAs of slot q-1: t rows rowW-2 .. rowW of t slot q-1 (negative: the last rows of slot q-2) ----
      if (do2) {
        if constexpr (HEAD) {        // residual rows of THIS slot (x groups q-2 / q-1), kept in registers for the next super-step's epilogue
          const int rr = rowW - 2;
          const int gx1 = qx >= 1 ? qx - 1 : qx - 1 + NXS;
          const uint4* xrow = xring + (rr < 0 ? gx2 : gx1) * T::XGP + (rr & (R - 1)) * T::XROW + lane_y;
#pragma unroll
          for (int s = 0; s < T::SPW; ++s)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
              rres[s][qd] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(xrow + qd * T::XW + s * 32) + gh * 8);
        }
        const uint4* tp[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int rr = rowW - 2 + ky;
          tp[ky] = tring + (rr < 0 ? gt2 : gt1) * T::TGP + (rr & (R - 1)) * T::TROW + lane_t;
        }
        const f32x16 bv = *reinterpret_cast<const f32x16*>(s_bias + (2 + gh) * 16);
        stream_conv36<DIL, T::TW, T::SPW>(tp, wf, bv, acc);
      }
      SN_STAMP(2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SN_STAMP(3);
      block_barrier();
      qx = qx + 1 == NXS ? 0 : qx + 1;
      qt = qt + 1 == T::NTS ? 0 : qt + 1;
      p4 = p4 + R >= T::PROWS ? p4 + R - T::PROWS : p4 + R;
    }
  }
}

}  // namespace sn
