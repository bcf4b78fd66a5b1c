// f32_tower_timing.hip — where a step of k_ref_conv_f32 (hobot_stereonet_amd/csrc/sn_tower_f32.hpp) spends its cycles: the
// kernel built with -DSN_F32_TIMING stamps s_memtime at six points of every step of one wave; 30 back-to-back launches of
// one 1280x720 layer keep the clock up.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSN_F32_TIMING -o scripts/build/f32_tower_timing scripts/f32_tower_timing.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../hobot_stereonet_amd/csrc/sn_kernels.hpp"

using namespace sn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int DIL, int CPH, int NW>
static void run(int grid_override) {
  using T = F32Tile<DIL, CPH, 64, NW>;
  const int H = 720, W = 1280;
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  const size_t n = (size_t)32 * H * W;
  float *x, *y, *w, *b;
  CK(hipMalloc(&x, n * 4));
  CK(hipMalloc(&y, n * 4));
  CK(hipMalloc(&w, 32 * 32 * 9 * 4));
  CK(hipMalloc(&b, 32 * 4));
  std::vector<float> hx(n);
  for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 20 & 1023) / 1024.f - 0.5f;
  CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hx.data(), 32 * 32 * 9 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hx.data(), 32 * 4, hipMemcpyHostToDevice));
  auto kern = k_ref_conv_f32<DIL, CPH, false, NW>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
  const int total = ((W + 63) / 64) * ((H + T::TH - 1) / T::TH);
  const int grid = grid_override > 0 ? grid_override : pr.multiProcessorCount;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int warm = getenv("WARM") ? atoi(getenv("WARM")) : 10;
  for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), T::LDS_BYTES, 0, x, y, nullptr, w, b, 1, H, W, 1);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), T::LDS_BYTES, 0, x, y, nullptr, w, b, 1, H, W, 1);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("dil %d cph %d waves %d grid %d: %.1f us per launch (%d tiles, LDS %d B) = %.1f TFLOP/s\n", DIL, CPH, NW, grid, ms / 30 * 1e3, total,
         T::LDS_BYTES, 16.9869e9 / (ms / 30 * 1e-3) / 1e12);
  std::vector<unsigned long long> st(64 * 8 + 8);
  CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(sn_f32_stamps), st.size() * 8));
  printf("workgroup 17: %llu s_memtime ticks in %llu wall ticks (100 MHz): %.3f GHz, %.1f us\n", st[64 * 8 + 2] - st[64 * 8 + 0],
         st[64 * 8 + 3] - st[64 * 8 + 1], (double)(st[64 * 8 + 2] - st[64 * 8 + 0]) / (double)(st[64 * 8 + 3] - st[64 * 8 + 1]) * 0.1,
         (double)(st[64 * 8 + 3] - st[64 * 8 + 1]) * 0.01);
  printf("step: wait barrier issue epilogue mfma | step total (s_memtime ticks; 100 MHz if constant, else shader clocks)\n");
  for (int s = 0; s < 40; ++s) {
    const unsigned long long* t = &st[s * 8];
    if (!t[5]) break;
    printf("%2d: %6llu %6llu %6llu %6llu %6llu | %6llu   gap to next %6llu\n", s, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4],
           t[5] - t[0], st[(s + 1) * 8] ? st[(s + 1) * 8] - t[5] : 0ull);
  }
}

int main(int argc, char** argv) {
  const int dil = argc > 1 ? atoi(argv[1]) : 1;
  const int grid = argc > 2 ? atoi(argv[2]) : 0;
  const int nw = argc > 3 ? atoi(argv[3]) : 8;
  if (nw == 16) {
    if (dil == 8) run<8, 4, 16>(grid);
    else if (dil == 2) run<2, 4, 16>(grid);
    else if (dil == 14) run<1, 4, 16>(grid);
    else run<1, 8, 16>(grid);
  } else {
    if (dil == 8) run<8, 4, 8>(grid);
    else if (dil == 2) run<2, 8, 8>(grid);
    else run<1, 8, 8>(grid);
  }
  return 0;
}
