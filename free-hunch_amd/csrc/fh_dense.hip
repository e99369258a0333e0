// Dense-matrix covariance kernels (float64, batched [bs][d][d] row-major): the device side of the reference's dense
// update rules update_covariance / update_bfgs (conditioning_utils/online_update_bfgs.py:377-463).
//
//   fh_dense_matvec   y = alpha * op(A) x + beta * y         one pass over A           (HBM bound: 8 d^2 bytes)
//   fh_dense_rank2    out = scale * (A + a1 u1 v1^T + a2 u2 v2^T) + shift * I          (HBM bound: 16 d^2 bytes)
//
// Layout choices for gfx950: a wave owns 4 consecutive rows and walks them with 16-byte loads (lane stride one
// double2 => each wave-load is one contiguous 1 KiB segment); x is read once per 4 rows and stays in L2.  The transposed
// product keeps the same coalesced row walk (thread = 2 columns, workgroup = a strip of 512 columns x a chunk of rows)
// and reduces the row chunks through caller-provided partials in a fixed order, so the result is deterministic.
#include "fh_common.h"

namespace {

typedef double d2_t __attribute__((ext_vector_type(2)));  // native vector: usable with the nontemporal builtins
constexpr int kRowsPerWave = 4;
constexpr int kDenseBlock = 256;

// y[row] = alpha * sum_j A[row][j] x[j] + beta * y[row];  grid (ceil(d / 16), 1, bs)
__global__ __launch_bounds__(kDenseBlock) void k_dense_mv(const double* __restrict__ A, const double* __restrict__ x,
                                                         double* __restrict__ y, int64_t d, double alpha,
                                                         double beta) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * (kDenseBlock / 64) + wave) * kRowsPerWave;
  if (row0 >= d) return;
  const int64_t img = blockIdx.z;
  const double* Ab = A + img * d * d;
  const double* xb = x + img * d;
  double acc[kRowsPerWave] = {0.0, 0.0, 0.0, 0.0};
  const int nrow = (int)((d - row0) < kRowsPerWave ? (d - row0) : kRowsPerWave);
  if ((d & 1) == 0 && nrow == kRowsPerWave) {  // rows start 16-byte aligned: double2 walk
    const int64_t d2 = d >> 1;
    const d2_t* x2 = reinterpret_cast<const d2_t*>(xb);
    const d2_t* r0 = reinterpret_cast<const d2_t*>(Ab + row0 * d);
    const d2_t* r1 = reinterpret_cast<const d2_t*>(Ab + (row0 + 1) * d);
    const d2_t* r2 = reinterpret_cast<const d2_t*>(Ab + (row0 + 2) * d);
    const d2_t* r3 = reinterpret_cast<const d2_t*>(Ab + (row0 + 3) * d);
    for (int64_t j = lane; j < d2; j += 64) {
      const d2_t xv = x2[j];
      const d2_t a0 = __builtin_nontemporal_load(&r0[j]);
      const d2_t a1 = __builtin_nontemporal_load(&r1[j]);
      const d2_t a2 = __builtin_nontemporal_load(&r2[j]);
      const d2_t a3 = __builtin_nontemporal_load(&r3[j]);
      acc[0] = fma(a0.x, xv.x, fma(a0.y, xv.y, acc[0]));
      acc[1] = fma(a1.x, xv.x, fma(a1.y, xv.y, acc[1]));
      acc[2] = fma(a2.x, xv.x, fma(a2.y, xv.y, acc[2]));
      acc[3] = fma(a3.x, xv.x, fma(a3.y, xv.y, acc[3]));
    }
  } else {
    for (int64_t j = lane; j < d; j += 64) {
      const double xv = xb[j];
#pragma unroll
      for (int r = 0; r < kRowsPerWave; ++r)
        if (r < nrow) acc[r] = fma(Ab[(row0 + r) * d + j], xv, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r) {
    const double s = fh::wave_sum(acc[r]);
    if (lane == 0 && r < nrow) {
      double* yp = y + img * d + row0 + r;
      *yp = beta == 0.0 ? alpha * s : alpha * s + beta * *yp;
    }
  }
}

// partial[img][chunk][j] = sum_{i in chunk} A[i][j] x[i];  grid (ceil(d / 512), nchunk, bs)
__global__ __launch_bounds__(kDenseBlock) void k_dense_mvt_partial(const double* __restrict__ A,
                                                                  const double* __restrict__ x,
                                                                  double* __restrict__ partial, int64_t d,
                                                                  int rows_per_chunk, int nchunk) {
  const int64_t img = blockIdx.z;
  const double* Ab = A + img * d * d;
  const double* xb = x + img * d;
  const int64_t j0 = ((int64_t)blockIdx.x * kDenseBlock + threadIdx.x) * 2;
  const int64_t i0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t i1 = i0 + rows_per_chunk < d ? i0 + rows_per_chunk : d;
  double s0 = 0.0, s1 = 0.0;
  if (j0 + 1 < d && (d & 1) == 0) {
    for (int64_t i = i0; i < i1; ++i) {
      const d2_t a = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(Ab + i * d + j0));
      const double xv = xb[i];
      s0 = fma(a.x, xv, s0);
      s1 = fma(a.y, xv, s1);
    }
  } else if (j0 < d) {
    for (int64_t i = i0; i < i1; ++i) {
      const double xv = xb[i];
      s0 = fma(Ab[i * d + j0], xv, s0);
      if (j0 + 1 < d) s1 = fma(Ab[i * d + j0 + 1], xv, s1);
    }
  }
  double* p = partial + (img * nchunk + blockIdx.y) * d;
  if (j0 < d) p[j0] = s0;
  if (j0 + 1 < d) p[j0 + 1] = s1;
}

__global__ __launch_bounds__(kDenseBlock) void k_dense_mvt_reduce(const double* __restrict__ partial,
                                                                 double* __restrict__ y, int64_t d, int nchunk,
                                                                 double alpha, double beta) {
  const int64_t img = blockIdx.z;
  const int64_t j = (int64_t)blockIdx.x * kDenseBlock + threadIdx.x;
  if (j >= d) return;
  const double* p = partial + img * nchunk * d + j;
  double s = 0.0;
  for (int c = 0; c < nchunk; ++c) s += p[(int64_t)c * d];
  double* yp = y + img * d + j;
  *yp = beta == 0.0 ? alpha * s : alpha * s + beta * *yp;
}

// out[i][j] = scale * (A[i][j] + a1 u1[i] v1[j] + a2 u2[i] v2[j]) + (i == j) * shift
// thread = 2 columns, workgroup = 512 columns x 8 rows; grid (ceil(d / 512), ceil(d / 8), bs)
constexpr int kR2Rows = 8;
__global__ __launch_bounds__(kDenseBlock) void k_dense_rank2(const double* A, double* out,  // out may alias A
                                                            int64_t d, const double* __restrict__ u1,
                                                            const double* __restrict__ v1,
                                                            const double* __restrict__ a1,
                                                            const double* __restrict__ u2,
                                                            const double* __restrict__ v2,
                                                            const double* __restrict__ a2, double scale,
                                                            double shift) {
  const int64_t img = blockIdx.z;
  const int64_t j0 = ((int64_t)blockIdx.x * kDenseBlock + threadIdx.x) * 2;
  if (j0 >= d) return;
  const bool pair = j0 + 1 < d;
  const double c1 = u1 != nullptr ? a1[img] : 0.0, c2 = u2 != nullptr ? a2[img] : 0.0;
  const double v1x = u1 != nullptr ? v1[img * d + j0] : 0.0, v1y = (u1 != nullptr && pair) ? v1[img * d + j0 + 1] : 0.0;
  const double v2x = u2 != nullptr ? v2[img * d + j0] : 0.0, v2y = (u2 != nullptr && pair) ? v2[img * d + j0 + 1] : 0.0;
  const int64_t i0 = (int64_t)blockIdx.y * kR2Rows;
  const bool vec = pair && (d & 1) == 0;
#pragma unroll
  for (int r = 0; r < kR2Rows; ++r) {
    const int64_t i = i0 + r;
    if (i >= d) break;
    const double w1 = u1 != nullptr ? c1 * u1[img * d + i] : 0.0;
    const double w2 = u2 != nullptr ? c2 * u2[img * d + i] : 0.0;
    const int64_t off = (img * d + i) * d + j0;
    double ax, ay = 0.0;
    if (vec) {
      const d2_t a = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(A + off));
      ax = a.x, ay = a.y;
    } else {
      ax = A[off];
      if (pair) ay = A[off + 1];
    }
    double ox = scale * fma(w2, v2x, fma(w1, v1x, ax));
    double oy = scale * fma(w2, v2y, fma(w1, v1y, ay));
    if (i == j0) ox += shift;
    if (i == j0 + 1) oy += shift;
    if (vec) {
      d2_t o;
      o.x = ox, o.y = oy;
      __builtin_nontemporal_store(o, reinterpret_cast<d2_t*>(out + off));
    } else {
      out[off] = ox;
      if (pair) out[off + 1] = oy;
    }
  }
}

}  // namespace

extern "C" {

int64_t fh_dense_matvec_scratch_doubles(int bs, int64_t d) {
  if (bs < 1 || d < 1) return 0;
  const int nchunk = (int)((d + 191) / 192 < 64 ? (d + 191) / 192 : 64);
  return (int64_t)bs * nchunk * d;
}

int fh_dense_matvec(const double* A, const double* x, double* y, double* scratch, int bs, int64_t d, int trans,
                    double alpha, double beta, void* stream) {
  if (!A || !x || !y || bs < 1 || d < 1 || bs > 65535) return FH_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (!trans) {
    const int64_t rows_per_block = (kDenseBlock / 64) * kRowsPerWave;
    hipLaunchKernelGGL(k_dense_mv, dim3((unsigned)((d + rows_per_block - 1) / rows_per_block), 1, bs),
                       dim3(kDenseBlock), 0, s, A, x, y, d, alpha, beta);
    FH_LAUNCH_CHECK();
    return 0;
  }
  if (!scratch) return FH_EINVAL;
  const int nchunk = (int)((d + 191) / 192 < 64 ? (d + 191) / 192 : 64);
  const int rows_per_chunk = (int)((d + nchunk - 1) / nchunk);
  hipLaunchKernelGGL(k_dense_mvt_partial, dim3((unsigned)((d + 2 * kDenseBlock - 1) / (2 * kDenseBlock)), nchunk, bs),
                     dim3(kDenseBlock), 0, s, A, x, scratch, d, rows_per_chunk, nchunk);
  FH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dense_mvt_reduce, dim3((unsigned)((d + kDenseBlock - 1) / kDenseBlock), 1, bs),
                     dim3(kDenseBlock), 0, s, scratch, y, d, nchunk, alpha, beta);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_dense_rank2(const double* A, double* out, int bs, int64_t d, const double* u1, const double* v1,
                   const double* a1, const double* u2, const double* v2, const double* a2, double scale, double shift,
                   void* stream) {
  if (!A || !out || bs < 1 || d < 1 || bs > 65535) return FH_EINVAL;
  if ((u1 && (!v1 || !a1)) || (u2 && (!v2 || !a2))) return FH_EINVAL;
  const int64_t gy = (d + kR2Rows - 1) / kR2Rows;
  if (gy > 65535) return FH_ESIZE;
  hipLaunchKernelGGL(k_dense_rank2, dim3((unsigned)((d + 2 * kDenseBlock - 1) / (2 * kDenseBlock)), (unsigned)gy, bs),
                     dim3(kDenseBlock), 0, (hipStream_t)stream, A, out, d, u1, v1, a1, u2, v2, a2, scale, shift);
  FH_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
