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

// FEAT bits (what else the real kernel does per super-step): 1 = conv1's waves stream 17 KB of x per step into an LDS ring by
// LDS-DMA from a 236 MB tensor, counted vmcnt; 2 = conv2's waves store 16 KB of y per step (four 16-byte stores per lane);
// 4 = conv1's waves write t (8 ds_write_b64 per wave), conv2's read the residual (8 ds_read_b64 per wave)
template <int NM, int NV, int FEAT = 0>
__global__ __launch_bounds__(512, 1) void k_mix(float* out, int steps, const uint4* xin = nullptr, uint4* yout = nullptr) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), role = wave >> 2;
  for (int i = tid; i < 3072; i += 512) lds[i] = uint4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
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
  const uint4* bp = lds + (wave & 3) * 512 + lane;
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
  uint4* ring = lds + 3072;                                  // 6 groups x 1088 slots behind the B-fragment area
  const size_t wg_base = (size_t)blockIdx.x * 60 * 1088;     // this workgroup's 60 groups of the source tensor
  auto dma = [&](int q) {
    if (FEAT & 1) {
      const unsigned dst = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)(ring + (q % 6) * 1088);
      const char* src = reinterpret_cast<const char*>(xin + wg_base + (size_t)(q % 60) * 1088);
      for (int k = 0; k < 5; ++k) {
        const int i = (wave & 3) + 4 * k;
        if (i < 17) {
          unsigned keep;
          const unsigned d2 = dst + (unsigned)i * 1024u, voff = (unsigned)(i * 64 + lane) * 16u;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "s"(d2), "v"(voff), "s"(src) : "memory");
        }
      }
    }
  };
  for (int q = 0; q < steps; ++q) {
    if (role == 0) {
      dma(q + 2);
      mfma_phase(q);
      valu_phase();
      if (FEAT & 4) {
#pragma unroll
        for (int k = 0; k < 8; ++k) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(lds + 2048 + (wave & 3) * 256 + (k & 3) * 64 + (lane & 31)) + (lane >> 5) * 8) = uint2{__float_as_uint(v[k]), __float_as_uint(v[k + 8])};
      }
      if (FEAT & 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      if (FEAT & 4) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(ring + ((q + 3) % 6) * 1088 + k * 66 + (lane & 31)) + (lane >> 5) * 8);
          v[k] += __uint_as_float(r.x);
        }
      }
      valu_phase();
      if (FEAT & 2) {
        uint4* o = yout + ((size_t)blockIdx.x * 60 + (q % 60)) * 1024 + (wave & 3) * 256 + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k * 64] = uint4{__float_as_uint(v[k]), __float_as_uint(v[k + 4]), __float_as_uint(v[k + 8]), __float_as_uint(v[k + 12])};
      }
      mfma_phase(q);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sacc = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sacc += v[r] + acc[0][r] + acc[1][r];
  if (sacc == 12345.678f) out[tid] = sacc;
}

template <int NM, int NV, int FEAT = 0>
static double run(const char* name, int steps) {
  auto kern = k_mix<NM, NV, FEAT>;
  const int lds = (3072 + 6 * 1088 + 64) * 16;      // 154,624 bytes: the real kernel's footprint
  static uint4 *xin = nullptr, *yout = nullptr;
  if (!xin) {
    hipMalloc(&xin, (size_t)256 * 60 * 1088 * 16 + 4096);
    hipMalloc(&yout, (size_t)256 * 60 * 1024 * 16);
    hipMemset(xin, 0x3c, (size_t)256 * 60 * 1088 * 16);
  }
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, out, steps, xin, yout);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
    printf("%s: launch failed\n", name);
    return 0;
  }
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, out, steps, xin, yout);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms / 20 < best) best = ms / 20;
  }
  const double us_per_k = best * 1e3 / steps * 1000.0;
  printf("%-34s NM=%2d NV=%3d FEAT=%d : %8.1f us per 1000 super-steps\n", name, NM, NV, FEAT, us_per_k);
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
  // what the rest of the real kernel's super-step adds to the direct mix (one feature at a time, then all)
  run<36, 112, 1>("direct + x ring by LDS-DMA", steps);
  run<36, 112, 2>("direct + y stores", steps);
  run<36, 112, 4>("direct + t writes / residual reads", steps);
  run<36, 112, 7>("direct + all three", steps);
  return 0;
}
