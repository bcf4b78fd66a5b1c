// mfma_power_probe.hip — what the fp16 matrix pipes sustain on this board at its power cap, to put the tower's
// roofline fraction (quoted against the 2.4 GHz data-sheet peak) next to a measured ceiling.
//   variant 0: v_mfma_f32_32x32x16_f16 only, operands in registers, two accumulators per wave
//   variant 1: the same with one conflict-free ds_read_b128 per MFMA feeding the B operand (how the tower kernels feed it)
// One 512-thread workgroup per CU (2 waves per SIMD), `iters` x 64 MFMAs per wave; run long enough for the clock to
// settle, sample rocm-smi beside it (scripts/refresh_profiles.sh).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/build/mfma_power_probe scripts/mfma_power_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void k_probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = uint4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  __syncthreads();
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)1.0f; }
  f32x16 acc0 = {0}, acc1 = {0};
  const uint4* p = lds + lane + (threadIdx.x >> 6) * 64;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      half8 b0 = b, b1 = b;
      if (VARIANT == 1) {
        const uint4 x0 = p[(2 * k) * 8 % 3072], x1 = p[(2 * k + 1) * 8 % 3072];
        b0 = *reinterpret_cast<const half8*>(&x0);
        b1 = *reinterpret_cast<const half8*>(&x1);
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc1, 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 12345.f) out[threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0;
  const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount;
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 2000;                                    // 64 MFMAs per iteration and wave
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&]() {
    if (variant == 0) hipLaunchKernelGGL(k_probe<0>, dim3(ncu), dim3(512), 0, 0, out, iters);
    else hipLaunchKernelGGL(k_probe<1>, dim3(ncu), dim3(512), 0, 0, out, iters);
  };
  launch();
  hipDeviceSynchronize();
  double total_ms = 0;
  int n = 0;
  float last = 0;
  while (total_ms < seconds * 1e3) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&last, e0, e1);
    total_ms += last;
    n += 20;
  }
  const double flop = (double)ncu * 8 * iters * 64 * 32768.0;      // per launch
  printf("variant %d (%s): last 20 launches %.1f TFLOP/s (%.3f ms per launch) after %.1f s of load, %d CUs\n", variant,
         variant ? "MFMA + one ds_read_b128 per MFMA" : "MFMA only", flop * 20 / (last * 1e-3) / 1e12, last / 20, total_ms / 1e3, ncu);
  return 0;
}
