// mall_probe.hip — how fast are streaming reads / writes / copies when the working set fits the 256 MiB
// Infinity Cache, compared with one that does not?  Decides whether one-pair-per-launch tower chunks (122 MB
// working set) can beat two-pair chunks (244 MB).  Build: hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ a, uint4* __restrict__ sink, size_t n) {
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint4 v = a[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;   // never true in practice
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ b, size_t n, unsigned v) {
  const uint4 w = {v, v + 1, v + 2, v + 3};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = w;
}
// read 2 streams, write 1 (the residual conv's traffic shape)
__global__ __launch_bounds__(256) void k_add(const uint4* __restrict__ a, const uint4* r, uint4* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint4 x = a[i], y = r[i];
    b[i] = uint4{x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w};
  }
}

int main() {
  const size_t MB = 1 << 20;
  const size_t sizes_mb[] = {8, 16, 30, 60, 90, 120, 180, 240, 480, 960};
  uint4 *a, *b;
  CK(hipMalloc(&a, 960 * MB));
  CK(hipMalloc(&b, 960 * MB));
  CK(hipMemset(a, 1, 960 * MB));
  CK(hipMemset(b, 2, 960 * MB));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256 * 8, reps = 20;
  printf("# per-tensor MB | copy a->b, b->a alternating (GB/s of read+write) | read a | write b | a,b->b (3 streams)\n");
  for (size_t mb : sizes_mb) {
    const size_t n = mb * MB / 16;
    float ms;
    double res[4];
    // ping-pong copy: the tensor written by launch i is read by launch i+1 (the tower's pattern)
    for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, b, a, n); }
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, b, a, n); }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    res[0] = 2.0 * reps * 2.0 * mb * MB / (ms * 1e-3) / 1e9;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, n);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 2 * reps; ++r) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    res[1] = 2.0 * reps * mb * MB / (ms * 1e-3) / 1e9;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n, 7u);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 2 * reps; ++r) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n, (unsigned)r);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    res[2] = 2.0 * reps * mb * MB / (ms * 1e-3) / 1e9;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_add, dim3(grid), dim3(256), 0, 0, a, b, b, n);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, b, a, n); hipLaunchKernelGGL(k_add, dim3(grid), dim3(256), 0, 0, a, b, b, n); }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    res[3] = reps * 5.0 * mb * MB / (ms * 1e-3) / 1e9;   // copy (2 streams) + add (3 streams)
    printf("%4zu MB  copy %7.0f  read %7.0f  write %7.0f  conv-pair-shaped %7.0f GB/s\n", mb, res[0], res[1], res[2], res[3]);
    fflush(stdout);
  }
  return 0;
}
