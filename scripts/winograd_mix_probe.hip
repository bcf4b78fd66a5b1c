// winograd_mix_probe.hip — would fewer MFMAs + transform arithmetic beat the streamed block's instruction mix? (VERDICT r3
// item 7; the arithmetic side is scripts/winograd_error_probe.py.)  A synthetic workgroup with the streamed block's shape:
// 512 threads = two roles of four waves, one wave of each role per SIMD, one workgroup per CU (LDS), one barrier per
// super-step; per super-step and wave NM MFMAs (32x32x16 f16), each fed by one conflict-free ds_read_b128 issued AHEAD of it,
// and NV dependent-chain-free VALU instructions standing for the epilogue; role 0 runs [MFMAs, VALU], role 1 [VALU, MFMAs]
// (the kernel's complementary phases).  Compared:
//     direct            NM = 36, NV = 112   (what k_ref_block_stream_f16 issues per wave and super-step; §5c)
//     Winograd F(2,3)   NM = 24, NV = 176   (+64 per role: B^T d with cross-lane packed-fp16 operations / A^T M in fp32)
//     F(2x2,3x3)        NM = 16, NV = 262   (+150 per role: 2-D transforms, the input one through LDS across rows)
// Prints microseconds per 1000 super-steps.   hipcc --offload-arch=gfx950 -O3 -o scripts/build/winograd_mix_probe scripts/winograd_mix_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV>
__global__ __launch_bounds__(512, 1) void k_mix(float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, role = wave >> 2;
  for (int i = tid; i < 8192; i += 512) lds[i] = uint4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  __syncthreads();
  half8 wf[18];
#pragma unroll
  for (int i = 0; i < 18; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) wf[i][e] = (_Float16)(0.001f * (float)(i + e + lane));
  f32x16 acc[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = (float)(lane + r);
  const uint4* bp = lds + (wave & 3) * 1024 + lane;
  auto mfma_phase = [&](int q) {
    half8 b[2][3];
    auto fetch = [&](int batch, half8 (&dst)[3]) {
#pragma unroll
      for (int k = 0; k < 3; ++k) dst[k] = *reinterpret_cast<const half8*>(bp + ((batch * 3 + k + q) & 7) * 64);
    };
    constexpr int NB = NM / 3;
    fetch(0, b[0]);
#pragma unroll
    for (int batch = 0; batch < NB; ++batch) {
      if (batch + 1 < NB) fetch(batch + 1, b[(batch + 1) & 1]);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        acc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(batch * 3 + k) % 18], b[batch & 1][k], acc[k & 1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < NM - 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
  };
  auto valu_phase = [&]() {
    // (inner loop of exactly 16 so that every register index is a compile-time constant whatever the outer loop does)
    for (int i = 0; i < NV / 16; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(acc[0][r]), "v"(acc[1][(r + 3) & 15]));
    }
#pragma unroll
    for (int r = 0; r < NV % 16; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(acc[0][r]), "v"(acc[1][(r + 3) & 15]));
  };
  for (int q = 0; q < steps; ++q) {
    if (role == 0) {
      mfma_phase(q);
      valu_phase();
    } else {
      valu_phase();
      mfma_phase(q);
    }
    __syncthreads();
  }
  float sacc = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sacc += v[r] + acc[0][r] + acc[1][r];
  if (sacc == 12345.678f) out[tid] = sacc;
}

template <int NM, int NV>
static double run(const char* name, int steps) {
  auto kern = k_mix<NM, NV>;
  const int lds = 128 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, out, steps);
  hipDeviceSynchronize();
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms / 20 < best) best = ms / 20;
  }
  const double us_per_k = best * 1e3 / steps * 1000.0;
  printf("%-28s NM=%2d NV=%3d : %8.1f us per 1000 super-steps\n", name, NM, NV, us_per_k);
  hipFree(out);
  return us_per_k;
}

int main() {
  const int steps = 2000;
  const double d = run<36, 112>("direct (the streamed block)", steps);
  const double w1 = run<24, 176>("Winograd F(2,3) along x", steps);
  const double w2 = run<15, 262>("Winograd F(2x2,3x3)", steps);      // 16 MFMAs: the probe issues them in batches of three
  const double m = run<36, 0>("MFMAs + LDS reads only", steps);
  const double v = run<0 + 3, 112>("(3 MFMAs) + the epilogue only", steps);
  printf("per 64 output pixels and wave: F(2,3) %.2fx, F(2x2,3x3) %.2fx the direct form's time (matrix part alone: %.2fx)\n", w1 / d, w2 / d,
         m / d);
  (void)v;
  return 0;
}
