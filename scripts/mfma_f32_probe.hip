// mfma_f32_probe.hip — what v_mfma_f32_32x32x2_f32 sustains on this part (the exact-fp32 tower, SN_PREC_FP32, is priced
// against the data-sheet 157.3 TFLOP/s = 64 FLOP/clk/SIMD at 2.4 GHz).
//   variant 0: MFMA only, operands in registers, two accumulators per wave
//   variant 1: one ds_read_b32 (B) per MFMA + one (A) per two MFMAs, as k_ref_conv_f32 feeds them
//   variant 2: as 0 with FOUR accumulators per wave
// argv: variant, waves per workgroup (8 or 16: two or four per SIMD), seconds
//   hipcc --offload-arch=gfx950 -O3 -o scripts/build/mfma_f32_probe scripts/mfma_f32_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT, int NT>
__global__ __launch_bounds__(NT) void k_probe(float* out, int iters) {
  __shared__ float lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += NT) lds[i] = 1.0f;
  __syncthreads();
  float a = 0.001f * lane, b = 1.0f;
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  const float* p = lds + lane + (threadIdx.x >> 6) * 64;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      float b0 = b, b1 = b, a0 = a;
      if (VARIANT == 1) {
        b0 = p[(2 * k) * 64 % 4096];
        b1 = p[(2 * k + 1) * 64 % 4096 + 1024];
        a0 = p[k * 64 % 2048 + 4096];
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc1, 0, 0, 0);
      if (VARIANT == 2) {
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc3, 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  if (s == 12345.f) out[threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0;
  const int waves = argc > 2 ? atoi(argv[2]) : 8;
  const double seconds = argc > 3 ? atof(argv[3]) : 3.0;
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount;
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&]() {
#define GO(V, NT) hipLaunchKernelGGL((k_probe<V, NT>), dim3(ncu), dim3(NT), 0, 0, out, iters)
    if (waves == 16) { if (variant == 0) GO(0, 1024); else if (variant == 1) GO(1, 1024); else GO(2, 1024); }
    else if (waves == 4) { if (variant == 0) GO(0, 256); else if (variant == 1) GO(1, 256); else GO(2, 256); }
    else { if (variant == 0) GO(0, 512); else if (variant == 1) GO(1, 512); else GO(2, 512); }
  };
  launch();
  hipDeviceSynchronize();
  double total_ms = 0;
  float last = 0;
  while (total_ms < seconds * 1e3) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&last, e0, e1);
    total_ms += last;
  }
  const double per_wave = (double)iters * 32 * (variant == 2 ? 4 : 2);
  const double flop = (double)ncu * waves * per_wave * 4096.0;      // per launch
  printf("variant %d, %d waves per CU: %.1f TFLOP/s (%.3f ms per launch) after %.1f s of load, %d CUs\n", variant, waves,
         flop * 20 / (last * 1e-3) / 1e12, last / 20, total_ms / 1e3, ncu);
  return 0;
}
