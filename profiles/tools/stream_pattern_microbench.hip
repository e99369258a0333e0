// microbenchmark: HBM read rate of two access patterns over 8 x 50.3 MB (d = 196608 rows, m = 32 float64 columns per image)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kRows = 768;
struct Ptrs { const double* B[8]; };
// pattern A: column-major [m][d]; workgroup = 768 rows x m columns = m segments of 6 KB
template <bool NT>
__global__ __launch_bounds__(256) void k_cols(Ptrs p, double* out, long d, int m) {
  const double* B = p.B[blockIdx.z];
  const long r0 = (long)blockIdx.x * kRows;
  const int tid = threadIdx.x;
  double acc = 0.0;
  // thread: rows tid*... use double2: 768 rows = 384 double2 -> 256 threads: first 128 threads take 2? simpler: 3 doubles per thread
  for (int j = 0; j < m; j += 8) {
    double v[8][3];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double* a = B + (long)(j + u) * d + r0 + q * 256 + tid;
        v[u][q] = NT ? __builtin_nontemporal_load(a) : *a;
      }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u][0] + v[u][1] + v[u][2];
  }
  if (acc == 12345.678) out[blockIdx.x] = acc;
}
// pattern B: tiled [d/768][m][768]: workgroup reads one contiguous m x 6 KB block
template <bool NT>
__global__ __launch_bounds__(256) void k_tiled(Ptrs p, double* out, long d, int m) {
  const double* B = p.B[blockIdx.z] + (long)blockIdx.x * m * kRows;
  const int tid = threadIdx.x;
  double acc = 0.0;
  for (int j = 0; j < m; j += 8) {
    double v[8][3];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double* a = B + (long)(j + u) * kRows + q * 256 + tid;
        v[u][q] = NT ? __builtin_nontemporal_load(a) : *a;
      }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u][0] + v[u][1] + v[u][2];
  }
  if (acc == 12345.678) out[blockIdx.x] = acc;
}
// pattern C: like A but 16-byte loads (double2), 384 lanes-worth -> threads 0..127 take 3 double2? use 1.5: skip; pattern D: tiled with double2
template <bool NT>
__global__ __launch_bounds__(256) void k_tiled2(Ptrs p, double* out, long d, int m) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2* B = reinterpret_cast<const d2*>(p.B[blockIdx.z] + (long)blockIdx.x * m * kRows);
  const int tid = threadIdx.x;
  double acc = 0.0;
  const int n2 = m * kRows / 2;  // double2 items in the block: 12288 at m = 32 -> 48 per thread
  for (int i = tid; i < n2; i += 256 * 8) {
    d2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const d2* a = B + i + u * 256;
      v[u] = NT ? __builtin_nontemporal_load(a) : *a;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 12345.678) out[blockIdx.x] = acc;
}
template <bool NT, bool REV>
__global__ __launch_bounds__(256) void k_apply_like(Ptrs p, const double* z, const double* D, const double* r, const double* c, double* o, long d, int m) {
  const int img = REV ? (int)gridDim.z - 1 - (int)blockIdx.z : blockIdx.z;
  const long blk = REV ? (long)gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const double* B = p.B[img];
  const int tid = threadIdx.x;
  __shared__ double cs[64];
  if (tid < m) cs[tid] = c[tid];
  __syncthreads();
  const long r0 = blk * kRows;
  double ch[8][3];
  for (int u = 0; u < 8; ++u) for (int q = 0; q < 3; ++q) ch[u][q] = 0.0;
  for (int j = 0; j < m; j += 8) {
    double v[8][3];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double* a = B + (long)(j + u) * d + r0 + q * 256 + tid;
        v[u][q] = NT ? __builtin_nontemporal_load(a) : *a;
      }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < 3; ++q) ch[u][q] = fma(v[u][q], cs[j + u], ch[u][q]);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double acc = ch[0][q];
    for (int u = 1; u < 8; ++u) acc += ch[u][q];
    const long i = (long)img * d + r0 + q * 256 + tid;
    o[i] = fma(r[i], acc, D[i] * z[i]);
  }
}
int main() {
  const long d = 196608; const int m = 32, nimg = 8;
  Ptrs p; double* out;
  for (int i = 0; i < nimg; ++i) { CK(hipMalloc((void**)&p.B[i], sizeof(double) * d * m)); { std::vector<double> h((size_t)d * m); unsigned long long x = 88172645463325252ULL + i; for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0 - 0.5; } CK(hipMemcpy((void*)p.B[i], h.data(), sizeof(double) * d * m, hipMemcpyHostToDevice)); } }
  CK(hipMalloc(&out, 8 * 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 grid(d / kRows, 1, nimg), blk(256);
  const double bytes = (double)nimg * d * m * 8;
  auto run = [&](const char* name, auto launch) {
    for (int w = 0; w < 5; ++w) launch();
    hipEventRecord(e0);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.2f us  %6.2f TB/s\n", name, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
    return 0;
  };
  run("columns (current), plain", [&]() { hipLaunchKernelGGL(k_cols<false>, grid, blk, 0, 0, p, out, d, m); });
  run("columns (current), nt", [&]() { hipLaunchKernelGGL(k_cols<true>, grid, blk, 0, 0, p, out, d, m); });
  run("tiled 8B loads, plain", [&]() { hipLaunchKernelGGL(k_tiled<false>, grid, blk, 0, 0, p, out, d, m); });
  run("tiled 8B loads, nt", [&]() { hipLaunchKernelGGL(k_tiled<true>, grid, blk, 0, 0, p, out, d, m); });
  run("tiled 16B loads, plain", [&]() { hipLaunchKernelGGL(k_tiled2<false>, grid, blk, 0, 0, p, out, d, m); });
  run("tiled 16B loads, nt", [&]() { hipLaunchKernelGGL(k_tiled2<true>, grid, blk, 0, 0, p, out, d, m); });
  double *z, *D, *r, *c, *o;
  CK(hipMalloc(&z, 8 * d * nimg)); CK(hipMalloc(&D, 8 * d * nimg)); CK(hipMalloc(&r, 8 * d * nimg)); CK(hipMalloc(&o, 8 * d * nimg)); CK(hipMalloc(&c, 8 * 64));
  CK(hipMemset(z, 0, 8 * d * nimg)); CK(hipMemset(D, 0, 8 * d * nimg)); CK(hipMemset(r, 0, 8 * d * nimg)); CK(hipMemset(c, 0, 8 * 64));
  run("apply-like fwd, nt", [&]() { hipLaunchKernelGGL((k_apply_like<true, false>), grid, blk, 0, 0, p, z, D, r, c, o, d, m); });
  run("apply-like fwd, plain", [&]() { hipLaunchKernelGGL((k_apply_like<false, false>), grid, blk, 0, 0, p, z, D, r, c, o, d, m); });
  run("read fwd then apply rev, nt (pair)", [&]() { hipLaunchKernelGGL(k_cols<true>, grid, blk, 0, 0, p, out, d, m); hipLaunchKernelGGL((k_apply_like<true, true>), grid, blk, 0, 0, p, z, D, r, c, o, d, m); });
  run("read fwd then apply fwd, nt (pair)", [&]() { hipLaunchKernelGGL(k_cols<true>, grid, blk, 0, 0, p, out, d, m); hipLaunchKernelGGL((k_apply_like<true, false>), grid, blk, 0, 0, p, z, D, r, c, o, d, m); });
  run("read fwd then apply rev, plain (pair)", [&]() { hipLaunchKernelGGL(k_cols<false>, grid, blk, 0, 0, p, out, d, m); hipLaunchKernelGGL((k_apply_like<false, true>), grid, blk, 0, 0, p, z, D, r, c, o, d, m); });
  return 0;
}
