// microbenchmark: out = a + b over 8 x 256 x 256 x 128 floats (268 MB per tensor) - the shape of the UNet's residual add, concat
// and GroupNorm streaming passes at the top resolution - in several launch shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_triad(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * U) {
    f4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * stride;
      if (j < n4) { x[u] = NTL ? __builtin_nontemporal_load(a + j) : a[j]; y[u] = NTL ? __builtin_nontemporal_load(b + j) : b[j]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * stride;
      if (j < n4) { const f4 v = x[u] + y[u]; if (NTS) __builtin_nontemporal_store(v, o + j); else o[j] = v; }
    }
  }
}
template <int U, bool NTL>
__global__ __launch_bounds__(256) void k_read2(const f4* __restrict__ a, const f4* __restrict__ b, float* __restrict__ o, long n4) {
  const long stride = (long)gridDim.x * 256;
  f4 acc = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * stride;
      if (j < n4) acc += (NTL ? __builtin_nontemporal_load(a + j) : a[j]) + (NTL ? __builtin_nontemporal_load(b + j) : b[j]);
    }
  }
  if (acc.x == 1234.5f) o[0] = acc.y;
}
int main() {
  const long n4 = 8L * 256 * 256 * 128 / 4;
  f4 *a, *b, *o;
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&o, n4 * 16));
  CK(hipMemset(a, 1, n4 * 16)); CK(hipMemset(b, 1, n4 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, double bytes, auto launch) {
    for (int w = 0; w < 3; ++w) launch();
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %8.1f us  %6.2f TB/s\n", name, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
    return 0;
  };
  const double b3 = 3.0 * n4 * 16, b2 = 2.0 * n4 * 16;
  run("triad U=1 grid 4096 (k_add_f32 now)", b3, [&]() { hipLaunchKernelGGL((k_triad<1, false, false>), dim3(4096), dim3(256), 0, 0, a, b, o, n4); });
  run("triad U=1 grid 16384", b3, [&]() { hipLaunchKernelGGL((k_triad<1, false, false>), dim3(16384), dim3(256), 0, 0, a, b, o, n4); });
  run("triad U=4 grid 2048", b3, [&]() { hipLaunchKernelGGL((k_triad<4, false, false>), dim3(2048), dim3(256), 0, 0, a, b, o, n4); });
  run("triad U=4 grid 4096", b3, [&]() { hipLaunchKernelGGL((k_triad<4, false, false>), dim3(4096), dim3(256), 0, 0, a, b, o, n4); });
  run("triad U=4 grid 8192", b3, [&]() { hipLaunchKernelGGL((k_triad<4, false, false>), dim3(8192), dim3(256), 0, 0, a, b, o, n4); });
  run("triad U=4 grid 2048 nt loads", b3, [&]() { hipLaunchKernelGGL((k_triad<4, true, false>), dim3(2048), dim3(256), 0, 0, a, b, o, n4); });
  run("triad U=4 grid 2048 nt stores", b3, [&]() { hipLaunchKernelGGL((k_triad<4, false, true>), dim3(2048), dim3(256), 0, 0, a, b, o, n4); });
  run("triad U=4 grid 2048 nt loads + stores", b3, [&]() { hipLaunchKernelGGL((k_triad<4, true, true>), dim3(2048), dim3(256), 0, 0, a, b, o, n4); });
  run("triad U=8 grid 1024 nt loads + stores", b3, [&]() { hipLaunchKernelGGL((k_triad<8, true, true>), dim3(1024), dim3(256), 0, 0, a, b, o, n4); });
  run("read 2 tensors U=4 grid 2048", b2, [&]() { hipLaunchKernelGGL((k_read2<4, false>), dim3(2048), dim3(256), 0, 0, a, b, (float*)o, n4); });
  run("read 2 tensors U=4 grid 2048 nt", b2, [&]() { hipLaunchKernelGGL((k_read2<4, true>), dim3(2048), dim3(256), 0, 0, a, b, (float*)o, n4); });
  return 0;
}
