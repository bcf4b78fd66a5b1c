// stream_block_probe.hip — development harness for k_ref_block_stream_f16 (hobot_stereonet_amd/csrc/sn_stream_block.hpp):
// checks the fused streaming block against a CPU restatement (fp16 operands, fp32 accumulation, t rounded to fp16) on small
// images and against the two-launch tower path on a full-size chunk, then times both.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/build/stream_block_probe scripts/stream_block_probe.hip
//   scripts/build/stream_block_probe [iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../hobot_stereonet_amd/csrc/sn_kernels.hpp"

using namespace sn;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

static RefGeom make_geom(int Hp, int Wp) {
  RefGeom g{};
  g.H = Hp;
  g.W = Wp;
  g.tiles_x = (Wp + 63) / 64;
  g.tiles_y = (Hp + 7) / 8;
  g.Hs = (Hp + 15) / 16 * 16 + 2 * kRefPad;
  g.Ws = g.tiles_x * 64 + 2 * kRefPad;
  g.rev = 0;
  return g;
}

struct Layer {
  std::vector<float> w, b;      // [co][ci][3][3], [co]
  uint4* wfrag = nullptr;
  float* bias = nullptr;
};

static void upload(Layer& L) {
  std::vector<_Float16> pk((size_t)18 * 64 * 8);
  for (int tap = 0; tap < 9; ++tap)
    for (int kk = 0; kk < 2; ++kk)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int co = lane & 31, ci = 16 * kk + 8 * (lane >> 5) + e;
          pk[(((size_t)tap * 2 + kk) * 64 + lane) * 8 + e] = (_Float16)L.w[((size_t)co * kC + ci) * 9 + tap];
        }
  CK(hipMalloc(&L.wfrag, pk.size() * 2));
  CK(hipMalloc(&L.bias, kC * 4));
  CK(hipMemcpy(L.wfrag, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(L.bias, L.b.data(), kC * 4, hipMemcpyHostToDevice));
}

static float q16(float v) { return (float)(_Float16)v; }

#ifndef PROBE_NWR
#define PROBE_NWR 4
#endif
template <int DIL>
static hipError_t launch_stream(hipStream_t st, const Layer& L1, const Layer& L2, const RefGeom& g, int ncu, const uint4* x, uint4* y,
                                int nimg, uint4* dump, int wg_override = 0) {
  using T = StreamTile<DIL, 64, 4, 6, PROBE_NWR>;
  auto kern = k_ref_block_stream_f16<DIL, 64, 4, 6, PROBE_NWR>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr = true;
  }
  StreamSched sc;
  sc.nstrips = (g.W + T::OW - 1) / T::OW;
  sc.hsub = (g.H + DIL - 1) / DIL;
  sc.total_rows = nimg * DIL * sc.nstrips * sc.hsub;
  int nwg = wg_override > 0 ? wg_override : ncu;
  if (nwg > sc.total_rows) nwg = sc.total_rows;
  sc.rows_per_wg = (sc.total_rows + nwg - 1) / nwg;
  const int grid = (sc.total_rows + sc.rows_per_wg - 1) / sc.rows_per_wg;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * PROBE_NWR), T::LDS_BYTES, st, x, y, L1.wfrag, L1.bias, L2.wfrag, L2.bias, g, sc, dump, StreamHeadArgs{});
  return hipGetLastError();
}

template <int DIL, int TW>
static hipError_t launch_v2(hipStream_t st, const Layer& L, const RefGeom& g, int ncu, const uint4* in, uint4* out, const uint4* res,
                            int nimg, unsigned* ctr) {
  using T = RefTile2<DIL, TW, 8, 3>;
  auto kern = res ? k_ref_conv_f16_v2<DIL, TW, true, 8, 2, 3> : k_ref_conv_f16_v2<DIL, TW, false, 8, 2, 3>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
  if (e != hipSuccess) return e;
  RefGeom gt = g;
  gt.tiles_x = (g.W + TW - 1) / TW;
  gt.tiles_y = (g.H + 7) / 8;
  const int total = gt.tiles_x * gt.tiles_y * nimg;
  const int band = (total + 7) / 8;
  int cap = ncu * 2 / 8;
  const int nlb = cap < band ? cap : band;
  hipLaunchKernelGGL(kern, dim3(nlb * 8), dim3(256), T::LDS_BYTES, st, in, out, res, L.wfrag, L.bias, gt, nimg, 1, ctr);
  return hipGetLastError();
}

static size_t slots_of(const RefGeom& g, int nimg) { return (size_t)nimg * 4 * g.Hs * g.Ws; }
static size_t slack_of(const RefGeom& g) { return (size_t)24 * g.Ws + 4096; }

template <int DIL>
static int check_small(int H, int W, int ncu, int wg_override, unsigned seed) {
  const RefGeom g = make_geom(H, W);
  const size_t slots = slots_of(g, 1), all = slots + slack_of(g);
  std::mt19937 rng(seed);
  std::normal_distribution<float> nd(0.f, 1.f);
  Layer L1, L2;
  for (Layer* L : {&L1, &L2}) {
    L->w.resize((size_t)kC * kC * 9);
    L->b.resize(kC);
    for (auto& v : L->w) v = q16(nd(rng) / 17.f);
    for (auto& v : L->b) v = nd(rng);
    upload(*L);
  }
  std::vector<float> x((size_t)kC * H * W);
  for (auto& v : x) v = q16(nd(rng));
  auto idx = [&](int c, int y, int xx) { return ((((size_t)(c >> 3)) * g.Hs + y + kRefPad) * g.Ws + xx + kRefPad) * 8 + (c & 7); };
  std::vector<_Float16> hin(all * 8, (_Float16)0.f);
  for (int c = 0; c < kC; ++c)
    for (int y = 0; y < H; ++y)
      for (int xx = 0; xx < W; ++xx) hin[idx(c, y, xx)] = (_Float16)x[((size_t)c * H + y) * W + xx];
  // CPU: t = q16(lrelu(conv1(x) + b1)), y = lrelu(x + conv2(t) + b2)
  auto conv = [&](const std::vector<float>& in, const Layer& L, std::vector<float>& out) {
    out.assign((size_t)kC * H * W, 0.f);
    for (int co = 0; co < kC; ++co)
      for (int y = 0; y < H; ++y)
        for (int xx = 0; xx < W; ++xx) {
          double a = L.b[co];
          for (int ci = 0; ci < kC; ++ci)
            for (int ky = 0; ky < 3; ++ky) {
              const int yy = y + (ky - 1) * DIL;
              if (yy < 0 || yy >= H) continue;
              for (int kx = 0; kx < 3; ++kx) {
                const int xc = xx + (kx - 1) * DIL;
                if (xc < 0 || xc >= W) continue;
                a += (double)L.w[((size_t)co * kC + ci) * 9 + ky * 3 + kx] * in[((size_t)ci * H + yy) * W + xc];
              }
            }
          out[((size_t)co * H + y) * W + xx] = (float)a;
        }
  };
  std::vector<float> t, v2;
  conv(x, L1, t);
  for (auto& v : t) v = q16(v > 0 ? v : 0.2f * v);
  conv(t, L2, v2);
  std::vector<float> ref((size_t)kC * H * W);
  for (size_t i = 0; i < ref.size(); ++i) {
    const float v = x[i] + v2[i];
    ref[i] = v > 0 ? v : 0.2f * v;
  }
  uint4 *da, *db, *dump;
  CK(hipMalloc(&da, all * 16));
  CK(hipMalloc(&db, all * 16));
  CK(hipMalloc(&dump, 65536));
  CK(hipMemcpy(da, hin.data(), all * 16, hipMemcpyHostToDevice));
  CK(hipMemset(db, 0, all * 16));
  CK(launch_stream<DIL>(nullptr, L1, L2, g, ncu, da, db, 1, dump, wg_override));
  CK(hipDeviceSynchronize());
  std::vector<_Float16> hout(all * 8);
  CK(hipMemcpy(hout.data(), db, all * 16, hipMemcpyDeviceToHost));
  double maxe = 0, sume = 0, scale = 0;
  size_t bad = 0;
  for (int c = 0; c < kC; ++c)
    for (int y = 0; y < H; ++y)
      for (int xx = 0; xx < W; ++xx) {
        const double r = ref[((size_t)c * H + y) * W + xx], got = (float)hout[idx(c, y, xx)];
        const double e = std::fabs(got - r);
        if (!(e <= 1e30)) ++bad;
        maxe = std::fmax(maxe, e);
        sume += e;
        scale = std::fmax(scale, std::fabs(r));
      }
  // the zero border must have survived
  size_t dirty = 0;
  for (int c = 0; c < 4; ++c)
    for (int y = 0; y < g.Hs; ++y)
      for (int xx = 0; xx < g.Ws; ++xx) {
        if (y >= kRefPad && y < kRefPad + H && xx >= kRefPad && xx < kRefPad + W) continue;
        for (int e = 0; e < 8; ++e)
          if ((float)hout[(((size_t)c * g.Hs + y) * g.Ws + xx) * 8 + e] != 0.f) ++dirty;
      }
  const double mean = sume / ref.size();
  const bool ok = bad == 0 && dirty == 0 && maxe <= 3e-3 * scale && mean < 3e-4 * scale;
  printf("small DIL=%d %dx%d wg=%d: max %.3e mean %.3e scale %.2f nan %zu border %zu -> %s\n", DIL, H, W, wg_override, maxe, mean, scale,
         bad, dirty, ok ? "OK" : "FAIL");
  hipFree(da);
  hipFree(db);
  hipFree(dump);
  return ok ? 0 : 1;
}

template <int DIL>
static int full_size(int H, int W, int nimg, int ncu, int iters) {
  const RefGeom g = make_geom(H, W);
  const size_t slots = slots_of(g, nimg), all = slots + slack_of(g);
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  Layer L1, L2;
  for (Layer* L : {&L1, &L2}) {
    L->w.resize((size_t)kC * kC * 9);
    L->b.resize(kC);
    for (auto& v : L->w) v = q16(nd(rng) / 17.f);
    for (auto& v : L->b) v = nd(rng);
    upload(*L);
  }
  std::vector<_Float16> hin(all * 8, (_Float16)0.f);
  for (int n = 0; n < nimg; ++n)
    for (int c = 0; c < kC; ++c)
      for (int y = 0; y < H; ++y) {
        _Float16* row = &hin[((((size_t)n * 4 + (c >> 3)) * g.Hs + y + kRefPad) * g.Ws + kRefPad) * 8];
        for (int xx = 0; xx < W; ++xx) row[(size_t)xx * 8 + (c & 7)] = getenv("PROBE_ZERO") ? (_Float16)0.f : (_Float16)nd(rng);      // PROBE_ZERO: all-zero activations (how much of the time is data-dependent power)
      }
  uint4 *dx, *dt, *dy, *dy2, *dump;
  unsigned* ctr;
  CK(hipMalloc(&dx, all * 16));
  CK(hipMalloc(&dt, all * 16));
  CK(hipMalloc(&dy, all * 16));
  CK(hipMalloc(&dy2, all * 16));
  CK(hipMalloc(&dump, 65536));
  CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&ctr), 4096, hipDeviceMallocFinegrained));
  CK(hipMemcpy(dx, hin.data(), all * 16, hipMemcpyHostToDevice));
  CK(hipMemset(dt, 0, all * 16));
  CK(hipMemset(dy, 0, all * 16));
  CK(hipMemset(dy2, 0, all * 16));
  // two-launch path: t = conv1(x); y = lrelu(x + conv2(t)) (not in place here so that x survives)
  auto two = [&]() {
    CK(hipMemsetAsync(ctr, 0, 4096, nullptr));
    CK((launch_v2<DIL, 64>(nullptr, L1, g, ncu, dx, dt, nullptr, nimg, ctr)));
    CK((launch_v2<DIL, 64>(nullptr, L2, g, ncu, dt, dy, dx, nimg, ctr + 512)));
  };
  two();
  CK(launch_stream<DIL>(nullptr, L1, L2, g, ncu, dx, dy2, nimg, dump));
  CK(hipDeviceSynchronize());
  std::vector<_Float16> a(all * 8), b(all * 8);
  CK(hipMemcpy(a.data(), dy, all * 16, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), dy2, all * 16, hipMemcpyDeviceToHost));
  double maxe = 0, sume = 0, scale = 0;
  size_t diff = 0, bad = 0;
  for (size_t i = 0; i < slots * 8; ++i) {
    const double r = (float)a[i], v = (float)b[i];
    const double e = std::fabs(r - v);
    if (!(e <= 1e30)) ++bad;
    if (e != 0) ++diff;
    maxe = std::fmax(maxe, e);
    sume += e;
    scale = std::fmax(scale, std::fabs(r));
  }
  const bool ok = bad == 0 && maxe <= 3e-3 * scale;
  printf("full DIL=%d %dx%d x%d: stream vs two launches: max %.3e mean %.3e scale %.2f differing %zu of %zu nan %zu -> %s\n", DIL, H, W,
         nimg, maxe, sume / (slots * 8), scale, diff, slots * 8, bad, ok ? "OK" : "FAIL");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;
  for (int i = 0; i < 3; ++i) two();
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) two();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  two launches : %.1f us per block\n", ms * 1e3 / iters);
  for (int i = 0; i < 3; ++i) CK(launch_stream<DIL>(nullptr, L1, L2, g, ncu, dx, dy2, nimg, dump));
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) CK(launch_stream<DIL>(nullptr, L1, L2, g, ncu, dx, dy2, nimg, dump));
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const double px = (double)nimg * H * W;
  printf("  stream kernel: %.1f us per block  (%.0f TFLOP/s of 2 x 18432 FLOP/px, %.2f TB/s of x + y)\n", us,
         px * 2 * 18432 / us * 1e-6, px * 128 / us * 1e-6);
#ifdef SN_STREAM_TIMING
  {
    std::vector<unsigned long long> tt(8 * 16 * 8);
    CK(hipMemcpy(tt.data(), reinterpret_cast<char*>(dump) + 1024, tt.size() * 8, hipMemcpyDeviceToHost));
    const unsigned long long base = tt[0];
    printf("  cycle stamps of workgroup 17, super-steps 8..23 (relative to wave 0 / step 8 / stamp 0)\n");
    printf("  conv1 waves 0-3: start | DMA issued | MFMAs done | t epilogue done | vmcnt + lgkm waits over\n");
    printf("  conv2 waves 4-7: start | y epilogue + 4 stores done | MFMAs done | lgkm wait over\n");
    for (int w = 0; w < 8; ++w) {
      printf("  wave %d:", w);
      for (int q = 0; q < 6; ++q) {
        printf(" [");
        for (int k = 0; k < (w < 4 ? 5 : 4); ++k) printf("%s%lld", k ? " " : "", (long long)(tt[(w * 16 + q) * 8 + k] - base));
        printf("]");
      }
      printf("\n");
    }
    {
      std::vector<unsigned long long> wg(4 * 256);
      CK(hipMemcpy(wg.data(), reinterpret_cast<char*>(dump) + 1024 + 1152 * 8, wg.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long w0 = ~0ull, w1 = 0;
      double cyc = 0, wall = 0;
      int n = 0;
      for (int b = 0; b < 256; ++b) {
        if (!wg[4 * b + 1]) continue;
        w0 = std::min(w0, wg[4 * b + 2]);
        w1 = std::max(w1, wg[4 * b + 3]);
        cyc += (double)(wg[4 * b + 1] - wg[4 * b]);
        wall += (double)(wg[4 * b + 3] - wg[4 * b + 2]);
        ++n;
      }
      printf("  %d workgroups: mean lifetime %.0f clock64 cycles = %.1f us (100 MHz wall clock) -> %.2f GHz; first start -> last end %.1f us\n",
             n, cyc / n, wall / n / 100.0, cyc / wall * 0.1, (double)(w1 - w0) / 100.0);
      for (int b : {0, 17, 100, 255})
        printf("    wg %d: start +%.1f us, lifetime %.1f us\n", b, (double)(wg[4 * b + 2] - w0) / 100.0, (double)(wg[4 * b + 3] - wg[4 * b + 2]) / 100.0);
    }
    // mean per-phase durations over the 16 steps
    for (int w = 0; w < 8; w += 4) {
      const int nk = w < 4 ? 5 : 4;
      double d[6] = {0, 0, 0, 0, 0, 0};
      for (int ww = w; ww < w + 4; ++ww)
        for (int q = 0; q < 15; ++q) {
          for (int k = 0; k + 1 < nk; ++k) d[k] += (double)(tt[(ww * 16 + q) * 8 + k + 1] - tt[(ww * 16 + q) * 8 + k]);
          d[nk - 1] += (double)(tt[(ww * 16 + q + 1) * 8] - tt[(ww * 16 + q) * 8 + nk - 1]);     // barrier wait
        }
      printf("  %s mean phase cycles:", w < 4 ? "conv1" : "conv2");
      for (int k = 0; k < nk; ++k) printf(" %.0f", d[k] / 60.0);
      printf("  (last = barrier)\n");
    }
  }
#endif
  return ok ? 0 : 1;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.name, ncu);
  int fails = 0;
  const bool timing_only = getenv("PROBE_TIMING_ONLY") != nullptr;      // experiments that compute wrong results on purpose
  if (!timing_only) {
  fails += check_small<1>(8, 62, ncu, 0, 1);
  fails += check_small<1>(37, 250, ncu, 0, 2);
  fails += check_small<1>(37, 250, ncu, 3, 3);        // few workgroups: several units per workgroup, restarts inside a strip
  fails += check_small<1>(100, 129, ncu, 0, 4);
  fails += check_small<1>(64, 96, ncu, 1, 5);         // one workgroup walks everything
  fails += check_small<2>(40, 70, ncu, 0, 6);
  fails += check_small<2>(45, 131, ncu, 5, 7);
  }
  if (fails) {
    printf("FAILED %d small cases\n", fails);
    return 1;
  }
  fails += full_size<1>(720, 1280, 2, ncu, iters);
  fails += full_size<2>(720, 1280, 2, ncu, iters);
  printf(fails ? "FAILED\n" : "ALL OK\n");
  return fails ? 1 : 0;
}
