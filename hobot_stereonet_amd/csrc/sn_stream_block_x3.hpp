// sn_stream_block_x3.hpp — fused residual block of the refinement tower on SPLIT operands (SN_PREC_F16X3), row-streaming
// form (gfx950): the line-buffer pipeline of sn_stream_block.hpp with every activation and weight as an fp16 hi / lo pair
// (value = hi + lo / 2048) and three MFMAs per product,
//
//   y = lrelu(x + conv2(lrelu(conv1(x) + b1)) + b2)        3x3, dilation DIL, 32 -> 32 -> 32 channels
//   x * w ~= xh * wh + (xh * wl + xl * wh) / 2048           acc0 <- xh wh,  acc1 <- xh wl, xl wh   (k_ref_conv_f16x3's sequence)
//
// It replaces the two k_ref_conv_f16x3 launches of a block (DnnNode::Run's network, stereonet_infer/src/stereonet_node.cpp:812;
// this is the arithmetic an SN_PREC_AUTO handle falls back to when a model leaves the fp16 tower's envelope).
// Why: the per-layer form is MEMORY bound — hi + lo tensors are fp32-sized, a one-pair chunk (x, t: 236 MB) already fills
// the 256 MB Infinity Cache, and a block moves 2 x 354 MB (in, out, residual, twice): 84 us per layer and pair = 4.2 TB/s
// with the matrix pipes at 0.24 of their peak (profiles/r05_f16x3_b64_bench.json; two workgroups per CU measured -6 %,
// profiles/r06_x3_wpc_ab.txt).  Here t never leaves LDS and x is read once: 236 MB per block.
//
// Differences from the fp16 kernel (read that file first: rings, slots, units, iterators and the counted waits are the same):
//   * x ring and t ring exist twice (hi part, lo part PART slots behind it); a step is R = 2 rows so that both fit 160 KB;
//   * waves per role NWR = 4 (default): two waves per SIMD, one of each role, as in the fp16 kernel — one role's epilogue and
//     DMA issue sit beside the other's MFMAs.  A wave holds its conv's 36 weight fragments (hi + lo, 144 registers), two
//     accumulators and ONE batch of fragments (the SIMD's other wave covers the reads): 236 VGPRs, no scratch.  The first form
//     (NWR = 2, one wave per SIMD with 512 registers, double-buffered hand-scheduled fragment batches; SN_X3_NWR=2) measured
//     1110-1118 pairs/s where this one runs 1229-1235; NWR = 4 WITH the double-buffered batches (256 VGPRs + 20 bytes of
//     scratch) 1164-1176 (profiles/r06_x3_nwr_ab.txt);
//   * the epilogues are k_ref_conv_f16x3's, expression for expression (v = acc0 + acc1 / 2048 [+ xh + xl / 2048], LeakyReLU,
//     hi = fp16(v), lo = fp16((v - hi) 2048)): the streamed block is bit-identical to the two launches
//     (tests/test_gpu_x3_stream.py through sn_dbg_ref_block_f16x3).
//   * conv2's waves read the residual rows of a slot into registers one super-step early (as the fp16 kernel's tail form
//     does): the x ring holds five groups instead of six, which is what lets dilation 8 (80-column rows) fit.
#pragma once

namespace sn {

template <int DIL_, int TW_ = 64, int R_ = 2, int NXS_ = 5, int NWR_ = 2>
struct StreamTileX3 {
  // conv2's waves fetch the residual rows of a slot one super-step early (into registers: a wave has 512), so the x ring
  // needs groups q-2 .. q+2 only — five groups of 80-column rows (dilation 8) fit beside the t ring, six do not
  static constexpr int DIL = DIL_, TW = TW_, R = R_, NXS = NXS_, NTS = 3, PF = NXS_ - 3;
  static constexpr int NWR = NWR_;                         // waves per role (conv1 / conv2)
  static constexpr int OW = TW - 2 * DIL;                  // valid y columns of a strip = strip pitch
  static constexpr int XW = TW + 2 * DIL;                  // x columns of a strip
  static constexpr int CSEG = TW / 32;                     // 32-pixel MFMA segments per row
  static constexpr int SPW = R * CSEG / NWR;               // segments per wave and conv
  static constexpr int XROW = 4 * XW, TROW = 4 * TW;       // slots per ring row: [channel block][column]
  static constexpr int XGROUP = R * XROW;                  // slots of one DMA group (R rows), per part
  static constexpr int NINST = (XGROUP + 63) / 64;         // 1 KiB LDS-DMA instructions per group and part
  static constexpr int KW = (2 * NINST + NWR - 1) / NWR;   // ... per conv1 wave (hi + lo), at most
  static constexpr int XGP = NINST * 64;                   // ring pitch of a group (the last instruction may overshoot)
  static constexpr int TGP = R * TROW;
  static constexpr int XRING = NXS * XGP;                  // slots per part
  static constexpr int TRING = NTS * TGP + 64;             // conv2's kx taps of the junk columns run past the last row
  static constexpr int LDS_BYTES = 2 * (XRING + TRING) * 16 + 2 * 2 * 16 * 4;      // + bias tables [conv][k-half][16]
  static constexpr int ROWS_ABOVE = R * DIL, ROWS_BELOW = (R + 2) * DIL - 1;
  static_assert(SPW == 1 || SPW == 2, "segments per wave");
  static_assert(CSEG % SPW == 0, "a wave's segments lie in one row");
  static_assert(PF == 2, "prefetch distance the counted waits are written for");
  static_assert(R == 2 || R == 4, "rows per step");
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
  static_assert(ROWS_ABOVE <= 2 * kRefPad, "a strip's first group stays inside the zero rows above an image (border + the plane before)");
};

// value of a split accumulator pair / a split fp16 pair (the expressions of k_ref_conv_f16x3's epilogue)
__device__ __forceinline__ float x3_acc(float a0, float a1) { return a0 + a1 * kSplitInv; }

// 54 SPW x 2 MFMAs of one conv for this wave's segments: for every (k-half, tap) acc0 += wh xh; acc1 += wl xh; acc1 += wh xl —
// k_ref_conv_f16x3's order per accumulator.  rp[ky] = LDS pointer into the HI part (lane part included), the lo part sits
// PART slots behind; COLW = slots per channel block row.  Fragments of one tap row of one k-half (3 kx x SPW segments, hi
// and lo) are fetched one batch ahead of the MFMAs that consume them.
template <int DIL, int COLW, int SPW, int PART, bool DB = true>
__device__ __forceinline__ void stream_conv108(const uint4* const (&rp)[3], const half8 (&wh)[18], const half8 (&wl)[18],
                                               const f32x16& bv, f32x16 (&a0)[SPW], f32x16 (&a1)[SPW]) {
  constexpr int NB = 3 * SPW;            // (kx, segment) pairs per batch
  constexpr int NBUF = DB ? 2 : 1;       // DB = false (two waves per SIMD): the other wave covers the fragment reads
  half8 bh[NBUF][NB], bl[NBUF][NB];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int batch, half8 (&dh)[NB], half8 (&dl)[NB]) {
    const int kk = batch / 3, ky = batch - kk * 3;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int s = 0; s < SPW; ++s) {
        const uint4* p = rp[ky] + (2 * kk * COLW + s * 32 + kx * DIL);
        dh[kx * SPW + s] = *reinterpret_cast<const half8*>(p);
        dl[kx * SPW + s] = *reinterpret_cast<const half8*>(p + PART);
      }
  };
  if (DB) fetch(0, bh[0], bl[0]);
#pragma unroll
  for (int batch = 0; batch < 6; ++batch) {
    if (DB) {
      if (batch + 1 < 6) fetch(batch + 1, bh[(batch + 1) & (NBUF - 1)], bl[(batch + 1) & (NBUF - 1)]);
    } else {
      fetch(batch, bh[0], bl[0]);
    }
    const int kk = batch / 3, ky = batch - kk * 3;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int s = 0; s < SPW; ++s) {
        const int w = (ky * 3 + kx) * 2 + kk;
        const half8 xh = bh[batch & (NBUF - 1)][kx * SPW + s], xl = bl[batch & (NBUF - 1)][kx * SPW + s];
        if (batch == 0 && kx == 0) {
          a0[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[w], xh, bv, 0, 0, 0);
          a1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[w], xh, zero, 0, 0, 0);
        } else {
          a0[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[w], xh, a0[s], 0, 0, 0);
          a1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[w], xh, a1[s], 0, 0, 0);
        }
        a1[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[w], xl, a1[s], 0, 0, 0);
      }
  }
  if (DB) {
    // issue order for the machine scheduler: the first batch's fragments up front, then two reads per three MFMAs
    // (0x100 = DS read, 0x008 = MFMA)
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * NB + 2, 0);
#pragma unroll
    for (int i = 0; i < 18 * SPW; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
  }
}

// xin / yout: the hi tensors; the lo tensors sit lo_bytes behind them.  wfrag1 / wfrag2: [hi 18][lo 18] x 64 slots
// (upload_ref_f16x3).
template <int DIL, int TW, int R, int NXS, int NWR>
__global__ __launch_bounds__(128 * NWR, NWR / 2) __attribute__((amdgpu_waves_per_eu(NWR / 2, NWR / 2))) void k_ref_block_stream_x3(
    const uint4* __restrict__ xin, uint4* __restrict__ yout, size_t lo_bytes, const uint4* __restrict__ wfrag1,
    const float* __restrict__ bias1, const uint4* __restrict__ wfrag2, const float* __restrict__ bias2, RefGeom g, StreamSched sc) {
  using T = StreamTileX3<DIL, TW, R, NXS, NWR>;
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  uint4* const xring = lds;                                  // hi part; lo part + XRING
  uint4* const tring = lds + 2 * T::XRING;                   // hi part; lo part + TRING
  float* const s_bias = reinterpret_cast<float*>(tring + 2 * T::TRING);      // [conv][k-half][16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, gh = lane >> 5;
  const int role = wave / NWR, rw = wave % NWR;             // role 0: conv1 + DMA, role 1: conv2 + stores

  const int f0 = (int)blockIdx.x * sc.rows_per_wg;
  int f1 = f0 + sc.rows_per_wg;
  if (f1 > sc.total_rows) f1 = sc.total_rows;
  if (f0 >= f1) return;                                     // uniform for the workgroup: before any barrier
  // super-steps of this workgroup: its slots + 2 to drain conv2's MFMAs and epilogue
  int nss = 2;
  for (int f = f0; f < f1;) {
    const int v0 = f % sc.hsub;
    int L = sc.hsub - v0;
    if (L > f1 - f) L = f1 - f;
    nss += (L + 2 + R - 1) / R + 1;
    f += L;
  }
  auto decode_sp = [&](int sp, int& img, int& py, int& x0) {
    const int t = sp / sc.nstrips;
    x0 = (sp - t * sc.nstrips) * T::OW;
    img = t / DIL;
    py = t - img * DIL;
  };
  const unsigned plane_b = (unsigned)g.Hs * (unsigned)g.Ws * 16u;          // bytes per channel block

  // this wave's 36 weight fragments (conv1 or conv2, hi + lo) and the bias tables
  half8 wfh[18], wfl[18];
  {
    const uint4* wsrc = role ? wfrag2 : wfrag1;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const uint4 a = wsrc[i * 64 + lane], b = wsrc[(18 + i) * 64 + lane];
      wfh[i] = *reinterpret_cast<const half8*>(&a);
      wfl[i] = *reinterpret_cast<const half8*>(&b);
    }
    if (tid < 64) {                       // accumulator register r of k-half g2 holds channel (r & 3) + 8 (r >> 2) + 4 g2
      const int c = tid >> 5, g2 = (tid >> 4) & 1, r = tid & 15;
      s_bias[tid] = (c ? bias2 : bias1)[(r & 3) + 8 * (r >> 2) + 4 * g2];
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      asm volatile("" : "+v"(wfh[i]));
      asm volatile("" : "+v"(wfl[i]));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }

  const int seg0 = rw * T::SPW;
  const int rowW = seg0 / T::CSEG;        // this wave's row of a step (its segments lie in it)
  const int cseg0 = seg0 % T::CSEG;

  if (role == 0) {
    // ============================ conv1 waves: DMA + conv1 ============================
    const int lane_x = gh * T::XW + cseg0 * 32 + j;        // B operand: block 2 kk + gh, column cseg0 * 32 + j (+ kx DIL)
    const int lane_w = cseg0 * 32 + j;                     // t write
    // this wave's DMA instructions i = rw, rw + NWR, .. of the 2 NINST of a group (i < NINST: hi part, else lo part):
    // per-lane source offsets relative to the group origin, LDS destination relative to the group's hi part
    constexpr int KWMAX = T::KW;
    const int kw = (2 * T::NINST - rw + NWR - 1) / NWR;    // instructions of this wave (uniform: KWMAX or KWMAX - 1)
    unsigned dma_voff[KWMAX];
#pragma unroll
    for (int k = 0; k < KWMAX; ++k) {
      int i = rw + NWR * k;
      i = i < T::NINST ? i : i - T::NINST;
      int s = i * 64 + lane;
      s = s < T::XGROUP ? s : T::XGROUP - 1;               // the overshoot of the last instruction lands in the ring pitch
      const int r = s / T::XROW;
      const int rem = s - r * T::XROW;
      const int blk = rem / T::XW;
      const int c = rem - blk * T::XW;
      dma_voff[k] = (unsigned)blk * plane_b + ((unsigned)(r * DIL) * (unsigned)g.Ws + (unsigned)c) * 16u;
    }
    long last_base = 0;
    auto dma_issue = [&](const StreamIter<R, 0>& it, int grp) {
      if (it.live) {
        int img, py, x0;
        decode_sp(it.sp, img, py, x0);
        const int row = (it.v0 - R + R * it.j) * DIL + py;               // image row of the group's first row (>= -R DIL)
        last_base = (((long)img * 4 * g.Hs + (row + kRefPad)) * (long)g.Ws + (x0 - 2 * DIL + kRefPad)) * 16;
      }
      const char* src = reinterpret_cast<const char*>(xin) + last_base;
      const unsigned dst = lds_addr(xring + grp * T::XGP);
#pragma unroll
      for (int k = 0; k < KWMAX; ++k)
        if (k < kw) {
          const int i = rw + NWR * k;                        // uniform
          const bool lo = i >= T::NINST;
          const int ii = lo ? i - T::NINST : i;
          glds16(dst + (lo ? (unsigned)T::XRING * 16u : 0u) + (unsigned)ii * 1024u, dma_voff[k], src + (lo ? lo_bytes : (size_t)0));
        }
    };
    auto wait_group = [&]() {              // all but this wave's youngest DMA group have landed
      if (kw == KWMAX) wait_vmcnt<KWMAX>();
      else wait_vmcnt<KWMAX - 1>();
    };
    StreamIter<R, 0> dm, c1;
    dm.live = c1.live = 0;
    dm.step(0, 0, f0, f1, sc.hsub);
    dma_issue(dm, 0);
    dm.step(1, 0, f0, f1, sc.hsub);
    dma_issue(dm, 1);
    wait_group();                         // group 0 landed (group 1 may still be in flight)
    block_barrier();                      // + the bias tables

    int c1_py = 0, c1_x0 = 0;
    int qx = 0, qt = 0;                   // q mod NXS, q mod NTS
    for (int q = 0; q < nss; ++q) {
      dm.step(q + 2, 0, f0, f1, sc.hsub);
      c1.step(q, 0, f0, f1, sc.hsub);
      const int do1 = c1.live && c1.j >= 1;
      const int gx0 = qx, gx1 = qx >= 1 ? qx - 1 : qx - 1 + NXS;
      const int gxp = qx + 2 >= NXS ? qx + 2 - NXS : qx + 2;
      dma_issue(dm, gxp);                 // overwrites group q+2-NXS = q-3: last read (residual fetch of slot q-2) in super-step q-1
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
        f32x16 a0[T::SPW], a1[T::SPW];
        stream_conv108<DIL, T::XW, T::SPW, T::XRING, (NWR <= 2)>(xp, wfh, wfl, bv, a0, a1);
        // epilogue: t = lrelu(acc0 + acc1 / 2048) as an fp16 hi / lo pair, zero outside the image (conv2's zero padding)
        const int trow = (c1.v0 - 1 + R * (c1.j - 1) + rowW) * DIL + c1_py;           // image row of this wave's t row
        const bool row_ok = trow >= 0 && trow < g.H;
        const int tc0 = c1_x0 - DIL;                                                  // image column of t column 0
        uint4* tw = tring + qt * T::TGP + rowW * T::TROW + lane_w;
#pragma unroll
        for (int s = 0; s < T::SPW; ++s) {
          const int c = tc0 + (cseg0 + s) * 32 + j;
          const bool inside = row_ok && c >= 0 && c < g.W;
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            half4 hh, hl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = x3_acc(a0[s][4 * qd + e], a1[s][4 * qd + e]);
              v = lrelu_fast(v);
              const _Float16 hi = (_Float16)v;
              hh[e] = hi;
              hl[e] = (_Float16)((v - (float)hi) * kSplitScale);
            }
            if (!inside) {
              hh = half4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
              hl = hh;
            }
            char* dst = reinterpret_cast<char*>(tw + qd * T::TW + s * 32) + gh * 8;
            *reinterpret_cast<half4*>(dst) = hh;
            *reinterpret_cast<half4*>(dst + (size_t)T::TRING * 16) = hl;
          }
        }
      }
      wait_group();                       // group q+1 landed: younger than it is only this super-step's group
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      block_barrier();
      qx = qx + 1 == NXS ? 0 : qx + 1;
      qt = qt + 1 == T::NTS ? 0 : qt + 1;
    }
  } else {
    // ============================ conv2 waves: conv2 + residual + stores ============================
    const int lane_t = gh * T::TW + cseg0 * 32 + j;
    const int lane_y = 2 * DIL + cseg0 * 32 + j;           // residual: block qd, 8 bytes at gh * 8
    block_barrier();

    StreamIter<R, 0> c2, ep;
    c2.live = ep.live = 0;
    int ep_img = 0, ep_py = 0, ep_x0 = 0;
    // after the half exchange below lane (j, gh) owns the whole 16-byte slots of channel blocks 2 gh and 2 gh + 1
    const unsigned lane_o = (unsigned)(cseg0 * 32 + j) * 16u + (unsigned)(2 * gh) * plane_b;
    f32x16 a0[T::SPW], a1[T::SPW];
    uint2 rrh[T::SPW][4], rrl[T::SPW][4];    // residual of the slot whose MFMAs ran last: [segment][channel block], hi and lo
#pragma unroll
    for (int s = 0; s < T::SPW; ++s) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        a0[s][r] = 0.f;
        a1[s][r] = 0.f;
      }
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        rrh[s][qd] = uint2{0u, 0u};
        rrl[s][qd] = uint2{0u, 0u};
      }
    }
    int qx = 0, qt = 0;
    for (int q = 0; q < nss; ++q) {
      c2.step(q, 1, f0, f1, sc.hsub);
      ep.step(q, 2, f0, f1, sc.hsub);
      const int do2 = c2.live && c2.j >= 1;
      const int doe = ep.live && ep.j >= 1;
      const int gx1 = qx >= 1 ? qx - 1 : qx - 1 + NXS, gx2 = qx >= 2 ? qx - 2 : qx - 2 + NXS;
      const int gt1 = qt >= 1 ? qt - 1 : qt - 1 + T::NTS, gt2 = qt >= 2 ? qt - 2 : qt - 2 + T::NTS;
      // ---- epilogue of slot q-2 (its MFMAs ran in super-step q-1, the accumulators crossed the barrier):
      // y = lrelu(x + acc) as an fp16 hi / lo pair -> global memory ----
      if (doe) {
        if (ep.j == 1) decode_sp(ep.sp, ep_img, ep_py, ep_x0);
        const int sub = ep.v0 - 2 + R * (ep.j - 1) + rowW;
        const int row = sub * DIL + ep_py;
        if (sub >= ep.v0 && sub < ep.v1 && row < g.H) {     // uniform: the unit's first two rows are junk
          const unsigned ob = (((unsigned)ep_img * 4u * (unsigned)g.Hs + (unsigned)(row + kRefPad)) * (unsigned)g.Ws +
                               (unsigned)(ep_x0 + kRefPad)) * 16u;
#pragma unroll
          for (int s = 0; s < T::SPW; ++s) {
            unsigned ph[4][2], pl[4][2];     // [channel block][channels 4 gh + {0,1} | {2,3}] as packed fp16 pairs, hi and lo
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const half4 vh = *reinterpret_cast<const half4*>(&rrh[s][qd]);
              const half4 vl = *reinterpret_cast<const half4*>(&rrl[s][qd]);
              half4 hh, hl;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float v = x3_acc(a0[s][4 * qd + e], a1[s][4 * qd + e]);
                v += (float)vh[e] + (float)vl[e] * kSplitInv;
                v = lrelu_fast(v);
                const _Float16 hi = (_Float16)v;
                hh[e] = hi;
                hl[e] = (_Float16)((v - (float)hi) * kSplitScale);
              }
              const uint2 uh = *reinterpret_cast<const uint2*>(&hh), ul = *reinterpret_cast<const uint2*>(&hl);
              ph[qd][0] = uh.x;
              ph[qd][1] = uh.y;
              pl[qd][0] = ul.x;
              pl[qd][1] = ul.y;
            }
            // half exchange (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second): blocks
            // (0, 2) and (1, 3) trade halves, so lanes gh = 0 end up with the full slots of blocks 0, 1 and lanes gh = 1
            // with those of blocks 2, 3 -> two 16-byte stores per segment and part instead of four 8-byte ones
            const int c = (cseg0 + s) * 32 + j;
            const bool ok = c < T::OW && ep_x0 + c < g.W;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
              unsigned(&pk)[4][2] = part ? pl : ph;
              uint4 sl[2];
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[k][0], pk[k + 2][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[k][1], pk[k + 2][1], false, false);
                sl[k] = uint4{r0[0], r1[0], r0[1], r1[1]};       // [own channels 0-3 | partner's 4-7] of block 2 gh + k
              }
              if (ok) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                  char* o = reinterpret_cast<char*>(yout) + (part ? lo_bytes : (size_t)0) +
                            (ob + (unsigned)k * plane_b + (unsigned)s * 512u);     // uniform
                  *reinterpret_cast<uint4*>(o + lane_o) = sl[k];
                }
              }
            }
          }
        }
      }
      // ---- conv2 MFMAs of slot q-1: t rows rowW-2 .. rowW of t slot q-1 (negative: the last rows of slot q-2) ----
      if (do2) {
        {      // residual rows of THIS slot (x groups q-2 / q-1), kept in registers for the next super-step's epilogue
          const int rr = rowW - 2;
          const uint4* xrow = xring + (rr < 0 ? gx2 : gx1) * T::XGP + (rr & (R - 1)) * T::XROW + lane_y;
#pragma unroll
          for (int s = 0; s < T::SPW; ++s)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const char* rp = reinterpret_cast<const char*>(xrow + qd * T::XW + s * 32) + gh * 8;
              rrh[s][qd] = *reinterpret_cast<const uint2*>(rp);
              rrl[s][qd] = *reinterpret_cast<const uint2*>(rp + (size_t)T::XRING * 16);
            }
        }
        const uint4* tp[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int rr = rowW - 2 + ky;
          tp[ky] = tring + (rr < 0 ? gt2 : gt1) * T::TGP + (rr & (R - 1)) * T::TROW + lane_t;
        }
        const f32x16 bv = *reinterpret_cast<const f32x16*>(s_bias + (2 + gh) * 16);
        stream_conv108<DIL, T::TW, T::SPW, T::TRING, (NWR <= 2)>(tp, wfh, wfl, bv, a0, a1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      block_barrier();
      qx = qx + 1 == NXS ? 0 : qx + 1;
      qt = qt + 1 == T::NTS ? 0 : qt + 1;
    }
  }
}

}  // namespace sn
