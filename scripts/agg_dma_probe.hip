// agg_dma_probe.hip — development harness for k_agg_x3s_dma (hobot_stereonet_amd/csrc/sn_agg_dma.hpp): times the kernel
// and k_conv_x3s<3, 1, 96, ...> on a 16-pair 45 x 80 x 12 volume of random data (random weight fragments: timing only,
// parity is tests/test_gpu_parity.py::test_lowres_split_conv3d) and, built with -DSN_AGG_TIMING, prints where a round goes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSN_AGG_TIMING] -o scripts/build/agg_dma_probe scripts/agg_dma_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../hobot_stereonet_amd/csrc/sn_kernels.hpp"

using namespace sn;

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int npairs = 16, Dl = 12, H = 45, W = 80;
  int ncu = 256;
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  ncu = pr.multiProcessorCount;
  VolPad g{Dl, H, W, VolPad::ph(H), VolPad::pw(W)};
  const size_t nsl = g.planes(npairs) * g.plane_slots();
  std::mt19937 rng(1);
  std::vector<_Float16> hv(nsl * 8, (_Float16)0.f);
  {
    std::normal_distribution<float> nd(0.f, 1.f);
    const size_t phw = (size_t)g.PH * g.PW;
    for (size_t P = 0; P < g.planes(npairs); ++P) {
      if (P % (Dl + 1) == 0) continue;
      for (int s = 0; s < 8; ++s)
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x)
            for (int e = 0; e < 8; ++e) hv[(((P * 8 + s) * phw) + (size_t)(y + 1) * g.PW + x + 1) * 8 + e] = (_Float16)nd(rng);
    }
  }
  uint4 *vin, *vout, *wx3;
  float *bias, *plain_in, *plain_out;
  unsigned long long* stamps;
  CK(hipMalloc(&vin, nsl * 16));
  CK(hipMalloc(&vout, nsl * 16));
  CK(hipMemcpy(vin, hv.data(), nsl * 16, hipMemcpyHostToDevice));
  CK(hipMemset(vout, 0, nsl * 16));
  const size_t wfr = (size_t)54 * 2 * 64;      // [chunk][tap][hi|lo][lane] uint4
  std::vector<_Float16> hw(wfr * 8);
  {
    std::normal_distribution<float> nd(0.f, 0.03f);
    for (auto& v : hw) v = (_Float16)nd(rng);
  }
  CK(hipMalloc(&wx3, wfr * 16));
  CK(hipMemcpy(wx3, hw.data(), wfr * 16, hipMemcpyHostToDevice));
  CK(hipMalloc(&bias, 32 * 4));
  CK(hipMemset(bias, 0, 32 * 4));
  const size_t plain = (size_t)npairs * Dl * 8 * H * W;       // slots
  CK(hipMalloc(&plain_in, plain * 16));
  CK(hipMalloc(&plain_out, plain * 16));
  CK(hipMemcpy(plain_in, hv.data(), plain * 16, hipMemcpyHostToDevice));      // any finite data
  CK(hipMalloc(&stamps, (256 + 4 * 1024) * 8));
  CK(hipMemset(stamps, 0, (256 + 4 * 1024) * 8));

  ConvArgs a{};
  a.wpk = reinterpret_cast<const float*>(wx3);
  a.bias = bias;
  a.nimg = npairs * Dl;
  a.cin_pad = 96;
  a.Ho = H;
  a.Wo = W;
  a.dil = 1;
  a.pad = 1;
  a.lrelu = 1;
  a.tiles_x = (W + 15) / 16;
  a.tiles_y = (H + 7) / 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  {
    auto kern = k_agg_x3s_dma<true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AggDma::LDS_BYTES));
    ConvArgs b = a;
    b.out = reinterpret_cast<float*>(vout);
#ifdef SN_AGG_TIMING
    b.res = reinterpret_cast<const float*>(stamps);
#endif
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(ncu), dim3(256), AggDma::LDS_BYTES, 0, b, vin, g);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(ncu), dim3(256), AggDma::LDS_BYTES, 0, b, vin, g);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int tiles = a.tiles_x * a.tiles_y * a.nimg;
    printf("k_agg_x3s_dma: %.1f us per launch (%d tiles, %.2f per workgroup)\n", ms * 1000.f / iters, tiles, (double)tiles / ncu);
  }
  {
    using T = X3sTile<3, 1, 96, 8, 16, 16>;
    auto kern = k_conv_x3s<3, 1, 96, 8, 16, 16, 1, true, false, SlotIn>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES));
    ConvArgs b = a;
    b.out = plain_out;
    SlotIn ld{reinterpret_cast<const uint4*>(plain_in), Dl, H, W};
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(ncu), dim3(256), T::LDS_BYTES, 0, b, ld);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(ncu), dim3(256), T::LDS_BYTES, 0, b, ld);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_conv_x3s:    %.1f us per launch\n", ms * 1000.f / iters);
  }
#ifdef SN_AGG_TIMING
  {
    std::vector<unsigned long long> st(256 + 4 * 1024);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    const char* names[6] = {"mfma loop", "partial write", "wait DMA", "barrier 1", "partner sum", "barrier 2"};
    for (int w = 0; w < 4; ++w) {
      double ph[7] = {0};
      for (int it = 0; it < 8; ++it) {
        const unsigned long long* s = &st[(w * 8 + it) * 8];
        for (int k = 0; k < 6; ++k) ph[k] += (double)(s[k + 1] - s[k]) / 8;
        if (it < 7) ph[6] += (double)(s[8] - s[6]) / 7;       // stamp 6 -> next round's stamp 0 (tile bookkeeping)
      }
      printf("wave %d:", w);
      for (int k = 0; k < 6; ++k) printf("  %s %.0f", names[k], ph[k]);
      printf("  next-tile setup %.0f  (shader cycles)\n", ph[6]);
    }
    double cyc = 0, wall = 0;
    int nb = 0;
    for (int b = 0; b < ncu; ++b) {
      const unsigned long long* d = &st[256 + 4 * b];
      if (d[1] <= d[0]) continue;
      cyc += (double)(d[1] - d[0]);
      wall += (double)(d[3] - d[2]);
      ++nb;
    }
    printf("workgroups: %d, mean %.0f shader cycles in %.1f us -> %.2f GHz\n", nb, cyc / nb, wall / nb / 100.0, cyc / wall / 10.0);
  }
#endif
  return 0;
}
