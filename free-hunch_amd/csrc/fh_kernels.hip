// Free Hunch covariance / DCT / operator / CG kernels for gfx950 (CDNA4, wave64), float64.
//
// Everything on this side of the path is HBM- or latency-bound vector work on d = 3*S*S elements
// plus the d x m factor base; the kernels are organised around coalesced 16-byte row-pair accesses,
// wave shuffles + LDS for reductions, and deterministic two-stage sums (no float atomics on this side, so a
// solve is bitwise reproducible).  Every kernel that runs inside the CG loop takes a device `done` flag and
// returns at once when it is set, which lets the host enqueue iterations in chunks without changing
// the reference's stopping rule.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "fh_common.h"

using namespace fh;

// Kernels inside the CG loop are gated per image: `states` is the array of device-resident CG control blocks (null
// outside the loop) and a workgroup returns at once when its image has met its stopping rule.
#define IMG_GUARD(states, img) \
  if ((states) != nullptr && (states)[(img)].done != 0) return

struct RtolArr {
  double v[FH_MAX_BATCH];
};

// Per-image pointer table of a lock-step batch (passed by value in the kernel arguments; the image is a grid dimension).
// The covariance-update kernels below take their per-image operands this way, so that ONE launch sequence updates all
// images of a batch; the single-image entry points pass a table with one entry (same kernels, same arithmetic).
struct PtrTab {
  double* p[FH_MAX_BATCH];
};
static inline PtrTab tab1(const double* q) {
  PtrTab t;
  memset(&t, 0, sizeof(t));
  t.p[0] = const_cast<double*>(q);
  return t;
}

// ------------------------------------------------------------------------------------------------
// float64 GEMM for the two DCT passes on the matrix cores (v_mfma_f64_16x16x4_f64):
//   C = A * op(B),  A [M][K] (lda), C [M][N] (ldc),  BT ? B [N][K] : B [K][N]  (ldb).
// 32x32 output tile per workgroup, one 16x16 MFMA tile per wave, K staged through LDS in chunks of
// 32 with both operands k-contiguous (row stride 34 doubles: conflict-free ds_read_b64 for the
// A[i=l&15][k=l>>4] / B[k=l>>4][j=l&15] operand maps).  f64 C/D map: col = l&15, row = (l>>4) + 4*reg.
// ------------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

template <bool BT, int TM, int TN = TM>  // TM x TN MFMA tiles (16x16) per wave: workgroup tile 32 TM x 32 TN
__global__ __launch_bounds__(256) void k_gemm_f64(const double* __restrict__ A, const double* __restrict__ B,
                                                  double* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                  int ldc, int64_t sA, int64_t sB, int64_t sC,
                                                  const fh_cg_state* __restrict__ states, int rows_per_plane,
                                                  const double* __restrict__ add, double add_scale) {
  constexpr int BM = 32 * TM, BN = 32 * TN, BK = 32, LD = BK + 2;
  IMG_GUARD(states, ((int)blockIdx.z + (int)(blockIdx.y * BM) / rows_per_plane) / 3);
  __shared__ __align__(16) double As[2][BM][LD];
  __shared__ __align__(16) double Bs[2][BN][LD];
  A += sA * blockIdx.z;
  B += sB * blockIdx.z;
  C += sC * blockIdx.z;
  if (add != nullptr) add += sC * blockIdx.z;  // epilogue C = acc + add_scale * add, same layout as C
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 16 * TM, wn = (wave & 1) * 16 * TN;
  const int li = lane & 15, lk = lane >> 4;
  double4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
  // staging: BM rows x 32 k per operand = 4*T doubles per thread: row = (tid >> 3) + 32 * q, k segment (tid & 7) * 4
  const int srow = tid >> 3, sseg = (tid & 7) * 4;
  double ra[TM][4], rb[TN][4];
  // full tiles (every DCT size that is a multiple of 64/32) take 16-byte loads; ragged edges fall back to guarded scalars
  const bool full = (m0 + BM <= M) && (n0 + BN <= N) && (K % BK == 0) && ((lda | ldb) % 2 == 0);
  auto load_chunk = [&](int k0) {  // global -> registers (issued one chunk ahead of its use)
#pragma unroll
    for (int q = 0; q < TM; ++q) {
      const int gm = m0 + srow + 32 * q;
      if (full) {
        const double2* pa = reinterpret_cast<const double2*>(A + (int64_t)gm * lda + k0 + sseg);
        const double2 a0 = pa[0], a1 = pa[1];
        ra[q][0] = a0.x, ra[q][1] = a0.y, ra[q][2] = a1.x, ra[q][3] = a1.y;
        continue;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gk = k0 + sseg + e;
        ra[q][e] = (gm < M && gk < K) ? A[(int64_t)gm * lda + gk] : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < TN; ++q) {
      if (full) {
        const double2* pb = BT ? reinterpret_cast<const double2*>(B + (int64_t)(n0 + srow + 32 * q) * ldb + k0 + sseg)
                               : reinterpret_cast<const double2*>(B + (int64_t)(k0 + srow) * ldb + n0 + sseg + 32 * q);
        const double2 b0 = pb[0], b1 = pb[1];
        rb[q][0] = b0.x, rb[q][1] = b0.y, rb[q][2] = b1.x, rb[q][3] = b1.y;
        continue;
      }
      if (BT) {
        const int gn = n0 + srow + 32 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int gk = k0 + sseg + e;
          rb[q][e] = (gn < N && gk < K) ? B[(int64_t)gn * ldb + gk] : 0.0;
        }
      } else {
        const int gk = k0 + srow;  // srow indexes k here, (sseg + 32 q) the n segment
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int gn = n0 + sseg + 32 * q + e;
          rb[q][e] = (gn < N && gk < K) ? B[(int64_t)gk * ldb + gn] : 0.0;
        }
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int q = 0; q < TM; ++q) {
      *reinterpret_cast<double2*>(&As[buf][srow + 32 * q][sseg]) = make_double2(ra[q][0], ra[q][1]);
      *reinterpret_cast<double2*>(&As[buf][srow + 32 * q][sseg + 2]) = make_double2(ra[q][2], ra[q][3]);
    }
#pragma unroll
    for (int q = 0; q < TN; ++q) {
      if (BT) {
        *reinterpret_cast<double2*>(&Bs[buf][srow + 32 * q][sseg]) = make_double2(rb[q][0], rb[q][1]);
        *reinterpret_cast<double2*>(&Bs[buf][srow + 32 * q][sseg + 2]) = make_double2(rb[q][2], rb[q][3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[buf][sseg + 32 * q + e][srow] = rb[q][e];
      }
    }
  };
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
    const bool more = k0 + BK < K;
    if (more) load_chunk(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][wm + 16 * i + li][kk + lk];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[buf][wn + 16 * j + li][kk + lk];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int gn = n0 + wn + 16 * j + li;
    if (gn >= N) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm + 16 * i + lk + 4 * r;
        if (gm < M) {
          const int64_t o = (int64_t)gm * ldc + gn;
          C[o] = add != nullptr ? fma(add_scale, add[o], acc[i][j][r]) : acc[i][j][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Symmetric-basis form of the DCT passes (k_dct_sym).  Both dense passes are written as the SAME per-plane product with a
// transposed result,  out[k][r] = sum_n P[k][n] in[r][n]:  pass A gives (X P_w^T)^T, pass B applied to that gives
// P_h X P_w^T in the natural orientation.  The DCT basis - and a symmetric blur folded into it - satisfies
// P[k][S-1-n] = (-1)^k P[k][n], so with the butterflies s = x_n + x_{S-1-n}, d = x_n - x_{S-1-n}
//     forward  (INV = 0):  out[2j][r] = sum_{n<S/2} Pe[j][n] s[r][n],   out[2j+1][r] = sum_{n<S/2} Po[j][n] d[r][n]
//     inverse  (INV = 1, basis Q = P^T, mirror symmetry in k):  E = sum_j Qe[k][j] x[r][2j],  O = sum_j Qo[k][j] x[r][2j+1],
//                          out[k][r] = E + O,  out[S-1-k][r] = E - O          (k < S/2)
// half the multiply-adds of the dense pass.  A workgroup owns 32 (j or k) x 32 (r) outputs of each parity, keeps its two
// half-basis slices (2 x 32 x S/2 doubles, 66 KB at S = 256) RESIDENT in LDS and walks over its share of the planes, so per
// output tile it pulls 85 KB from L2 instead of the 192 KB of the 64 x 32 dense tile (the dense kernel is bound by the
// ~12 B/clk a CU pulls from L2, not by the f64 matrix cores).  Same MFMA (v_mfma_f64_16x16x4_f64), same LDS row pitch rule.
// ------------------------------------------------------------------------------------------------
// per-image diagonal applied to the result of a pass (the m = 0 covariance apply C z = D .* z folded into the forward DCT)
struct fh_diag_tab {
  const double* D[FH_MAX_BATCH];  // null table (D[0] == nullptr): no scaling
};

template <bool INV>
__global__ __launch_bounds__(256) void k_dct_sym(const double* __restrict__ Ah, const double* __restrict__ X,
                                                 double* __restrict__ C, int S, int planes,
                                                 const fh_cg_state* __restrict__ states, const double* __restrict__ add,
                                                 double add_scale, fh_diag_tab dg, double* __restrict__ dot_part,
                                                 int dot_stride) {
  // dot_part != null (with add): the workgroup also leaves sum(add .* out) of every (plane, tile) it completes in
  // dot_part[image * dot_stride + (plane % 3) * tiles + tile] - the p.Ap reduction of the CG iteration rides in the pass
  // that produces Ap (A p = sigma_y^2 p + ..., add = p), summed later in a fixed order
  // K chunks of 64 (two per plane at S = 256): the per-chunk cost besides the 16 MFMA pairs - LDS stores, the barrier, the
  // wait for the prefetched rows - was ~65 % of a 32-wide chunk's time (measured 14.4 us per pass at BK = 32, 36 % of the f64
  // MFMA rate); halving the chunk count halves it.  Two LDS buffers, the next chunk prefetched into registers.
  constexpr int TJ = 32, TR = 32, BK = 64, LDB = BK + 2, NBUF = 2, NP8 = BK / 32;  // NP8 pieces of 8 doubles per thread
  extern __shared__ __align__(16) double smem[];
  const int H = S >> 1, LDA = H + 2;
  double* As_e = smem;                      // [TJ][LDA] resident half-basis slices
  double* As_o = As_e + TJ * LDA;           // [TJ][LDA]
  double* Bs = As_o + TJ * LDA;             // [NBUF buffers][2 parities][TR][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int jw = (wave >> 1) * 16, rw = (wave & 1) * 16;
  const int j0 = blockIdx.y * TJ, r0 = blockIdx.x * TR;
  for (int i = tid; i < TJ * (H / 2); i += 256) {
    const int row = i / (H / 2), c2 = (i % (H / 2)) * 2;
    const double2 ve = *reinterpret_cast<const double2*>(Ah + (int64_t)(j0 + row) * H + c2);
    const double2 vo = *reinterpret_cast<const double2*>(Ah + (int64_t)H * H + (int64_t)(j0 + row) * H + c2);
    *reinterpret_cast<double2*>(&As_e[row * LDA + c2]) = ve;
    *reinterpret_cast<double2*>(&As_o[row * LDA + c2]) = vo;
  }
  const int srow = tid >> 3, sq = tid & 7;  // staging: row of the r tile, lane within the row's 8-thread group
  const int nkc = H / BK;
  // planes of this workgroup: p = blockIdx.z, + gridDim.z, ...  (finished images of a CG batch are skipped)
  int pl[8], npl = 0;
  for (int p = blockIdx.z; p < planes && npl < 8; p += gridDim.z)
    if (states == nullptr || states[p / 3].done == 0) pl[npl++] = p;
  const int total = npl * nkc;
  if (total == 0) return;
  double ra[NP8][8];
  auto load_chunk = [&](int t) {
    const double* row = X + (int64_t)pl[t / nkc] * S * S + (int64_t)(r0 + srow) * S;
    const int n0 = (t % nkc) * BK;
#pragma unroll
    for (int h = 0; h < NP8; ++h) {
      if (INV) {  // (even, odd) sample pairs j = n0 + 32 h + 8 e + sq: 8 lanes of a row read 128 contiguous bytes per load
        const double2* p2 = reinterpret_cast<const double2*>(row + 2 * (n0 + 32 * h)) + sq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double2 v = p2[8 * e];
          ra[h][2 * e] = v.x, ra[h][2 * e + 1] = v.y;
        }
      } else {    // 4 samples at n = n0 + 32 h + 4 sq + i and their mirror images S - 1 - n
        const int nb = n0 + 32 * h + 4 * sq;
        const double2* lo = reinterpret_cast<const double2*>(row + nb);
        const double2* hi = reinterpret_cast<const double2*>(row + S - 4 - nb);
        const double2 l0 = lo[0], l1 = lo[1], h0 = hi[0], h1 = hi[1];
        ra[h][0] = l0.x, ra[h][1] = l0.y, ra[h][2] = l1.x, ra[h][3] = l1.y;
        ra[h][4] = h1.y, ra[h][5] = h1.x, ra[h][6] = h0.y, ra[h][7] = h0.x;  // mirrors of samples 0 .. 3
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int h = 0; h < NP8; ++h) {
      if (INV) {  // pair j = 32 h + 8 e + sq of the chunk: even sample to parity 0, odd sample to parity 1
        double* c0 = Bs + (buf * 2 + 0) * TR * LDB + srow * LDB + 32 * h + sq;
        double* c1 = Bs + (buf * 2 + 1) * TR * LDB + srow * LDB + 32 * h + sq;
#pragma unroll
        for (int e = 0; e < 4; ++e) c0[8 * e] = ra[h][2 * e], c1[8 * e] = ra[h][2 * e + 1];
      } else {
        double* b0 = Bs + (buf * 2 + 0) * TR * LDB + srow * LDB + 32 * h + 4 * sq;
        double* b1 = Bs + (buf * 2 + 1) * TR * LDB + srow * LDB + 32 * h + 4 * sq;
        *reinterpret_cast<double2*>(b0) = make_double2(ra[h][0] + ra[h][4], ra[h][1] + ra[h][5]);
        *reinterpret_cast<double2*>(b0 + 2) = make_double2(ra[h][2] + ra[h][6], ra[h][3] + ra[h][7]);
        *reinterpret_cast<double2*>(b1) = make_double2(ra[h][0] - ra[h][4], ra[h][1] - ra[h][5]);
        *reinterpret_cast<double2*>(b1 + 2) = make_double2(ra[h][2] - ra[h][6], ra[h][3] - ra[h][7]);
      }
    }
  };
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  double4_t acc_e = {0.0, 0.0, 0.0, 0.0}, acc_o = {0.0, 0.0, 0.0, 0.0};
  int buf = 0;
  for (int t = 0; t < total; ++t, buf ^= 1) {
    const bool more = t + 1 < total;
    if (more) load_chunk(t + 1);
    const int kc = t % nkc;
    const double* be = Bs + (buf * 2 + 0) * TR * LDB + (rw + li) * LDB + lk;
    const double* bo = Bs + (buf * 2 + 1) * TR * LDB + (rw + li) * LDB + lk;
    const double* ae = As_e + (jw + li) * LDA + kc * BK + lk;
    const double* ao = As_o + (jw + li) * LDA + kc * BK + lk;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      acc_e = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[kk], be[kk], acc_e, 0, 0, 0);
      acc_o = __builtin_amdgcn_mfma_f64_16x16x4f64(ao[kk], bo[kk], acc_o, 0, 0, 0);
    }
    if (kc == nkc - 1) {  // a plane is complete: write its two output row families
      const int plane = pl[t / nkc];
      double* Cp = C + (int64_t)plane * S * S;
      const double* Ap = add != nullptr ? add + (int64_t)plane * S * S : nullptr;
      const double* Dp = dg.D[0] != nullptr ? dg.D[plane / 3] + (int64_t)(plane % 3) * S * S : nullptr;
      const int col = r0 + rw + li;
      double dsum = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jj = j0 + jw + lk + 4 * q;
        int64_t o0, o1;
        double v0, v1;
        if (INV) {
          o0 = (int64_t)jj * S + col, o1 = (int64_t)(S - 1 - jj) * S + col;
          v0 = acc_e[q] + acc_o[q], v1 = acc_e[q] - acc_o[q];
        } else {
          o0 = (int64_t)(2 * jj) * S + col, o1 = o0 + S;
          v0 = acc_e[q], v1 = acc_o[q];
        }
        if (Dp != nullptr) v0 = Dp[o0] * v0, v1 = Dp[o1] * v1;  // same product as k_rep_apply2 forms for m = 0
        if (Ap != nullptr) {
          const double a0 = Ap[o0], a1 = Ap[o1];
          v0 = fma(add_scale, a0, v0), v1 = fma(add_scale, a1, v1);
          dsum = fma(a0, v0, fma(a1, v1, dsum));
        }
        Cp[o0] = v0;
        Cp[o1] = v1;
      }
      if (dot_part != nullptr) {  // (uniform: every thread of the workgroup completes the plane in this iteration)
        __shared__ double dred[4];
        dsum = block_sum_256(dsum, dred);
        if (tid == 0)
          dot_part[(int64_t)(plane / 3) * dot_stride + (plane % 3) * (int)(gridDim.x * gridDim.y) + blockIdx.y * gridDim.x +
                   blockIdx.x] = dsum;
      }
      acc_e = double4_t{0.0, 0.0, 0.0, 0.0};
      acc_o = double4_t{0.0, 0.0, 0.0, 0.0};
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }
}

// Two dense S x S passes over `planes` images: T = X b_w^T (along W), out = b_h T (along H) [+ add_scale * add].
// b_w = b_h = the DCT basis (or its transpose) gives the 2-D DCT-II / DCT-III; a separable blur folded into the bases
// (fh_problem.fold_*) makes the same two passes compute dct2(A^T x) or A(idct2(x)).
static int dct2d_launch_bases(fh_context* ctx, const double* in, double* out, int planes, const double* b_w, const double* b_h,
                              const double* add, double add_scale, const fh_cg_state* states, hipStream_t st);

// the two symmetric passes: tmp = (X P_w^T)^T, out = P_h X P_w^T (+ add_scale * add); sym_* = packed half bases [2][S/2][S/2]
static int dct2d_launch_sym(fh_context* ctx, const double* in, double* out, int planes, const double* sym_w,
                            const double* sym_h, int inverse, const double* add, double add_scale,
                            const fh_cg_state* states, hipStream_t st, const fh_batch* diag = nullptr,
                            double* dot_part = nullptr, int dot_stride = 0, int* dot_nparts = nullptr) {
  const int S = ctx->S, H = S / 2;
  if (planes > ctx->planes_max || S % 128 != 0) return FH_ESIZE;
  const int gx = S / 32, gy = H / 32;
  int gz = 256 / (gx * gy);
  if (gz < (planes + 7) / 8) gz = (planes + 7) / 8;
  if (gz > planes) gz = planes;
  if (gz < 1) gz = 1;
  if (H % 64 != 0) return FH_ESIZE;  // K chunks of 64
  const size_t lds = ((size_t)2 * 32 * (H + 2) + (size_t)2 * 2 * 32 * 66) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    FH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dct_sym<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 140 * 1024));
    FH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dct_sym<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 140 * 1024));
    attr_set = true;
  }
  const dim3 grid(gx, gy, gz);
  fh_diag_tab none, dg;
  memset(&none, 0, sizeof(none));
  memset(&dg, 0, sizeof(dg));
  if (diag != nullptr)
    for (int i = 0; i < diag->nimg && i < FH_MAX_BATCH; ++i) dg.D[i] = diag->D[i];
  if (add == nullptr || 3 * gx * gy > 256) dot_part = nullptr;  // (the consumer sums <= 256 partials per image)
  if (dot_nparts != nullptr) *dot_nparts = dot_part != nullptr ? 3 * gx * gy : 0;
  if (inverse) {
    hipLaunchKernelGGL(k_dct_sym<true>, grid, dim3(256), lds, st, sym_w, in, ctx->tmp_img, S, planes, states,
                       (const double*)nullptr, 0.0, none, (double*)nullptr, 0);
    hipLaunchKernelGGL(k_dct_sym<true>, grid, dim3(256), lds, st, sym_h, (const double*)ctx->tmp_img, out, S, planes, states,
                       add, add_scale, dg, dot_part, dot_stride);
  } else {
    hipLaunchKernelGGL(k_dct_sym<false>, grid, dim3(256), lds, st, sym_w, in, ctx->tmp_img, S, planes, states,
                       (const double*)nullptr, 0.0, none, (double*)nullptr, 0);
    hipLaunchKernelGGL(k_dct_sym<false>, grid, dim3(256), lds, st, sym_h, (const double*)ctx->tmp_img, out, S, planes, states,
                       add, add_scale, dg, dot_part, dot_stride);
  }
  FH_LAUNCH_CHECK();
  return 0;
}

static int dct2d_launch(fh_context* ctx, const double* in, double* out, int planes, int inverse,
                        const fh_cg_state* states, hipStream_t st) {
  static const bool no_sym = getenv("FH_DCT_NOSYM") != nullptr;  // A/B switch: the dense passes
  if (ctx->sym_fwd != nullptr && !no_sym) {
    const double* sb = inverse ? ctx->sym_inv : ctx->sym_fwd;
    return dct2d_launch_sym(ctx, in, out, planes, sb, sb, inverse, nullptr, 0.0, states, st);
  }
  const double* b1 = inverse ? ctx->basis_t : ctx->basis;
  return dct2d_launch_bases(ctx, in, out, planes, b1, b1, nullptr, 0.0, states, st);
}

static int dct2d_launch_bases(fh_context* ctx, const double* in, double* out, int planes, const double* b1, const double* b2,
                              const double* add, double add_scale, const fh_cg_state* states, hipStream_t st) {
  const int S = ctx->S;
  if (planes > ctx->planes_max) return FH_ESIZE;
  // Tile choice by workgroup count (256 CUs, the GEMMs are MFMA-bound: what matters is an even number of tile-units per
  // CU).  64x64 tiles halve the LDS/global traffic per flop but S = 256, 24 planes gives 384 of them = 1.5 per CU
  // (half the CUs carry two: 24 us); 64x32 tiles give 768 = exactly 3 half-size units per CU.
  static const int force = getenv("FH_DCT_TILE") ? atoi(getenv("FH_DCT_TILE")) : 0;  // 1: 64x64, 2: 64x32, 3: 32x32
  const int64_t n64 = (int64_t)((S + 63) / 64) * ((S + 63) / 64) * planes;
  int tile = n64 >= 1024 ? 1 : ((int64_t)((S + 31) / 32) * ((S + 63) / 64) * planes >= 512 ? 2 : 3);
  if (force) tile = force;
  // pass 1 (along W): T[r][k] = sum_n X[r][n] * b1[k][n]
  if (tile == 1) {
    dim3 grid((S + 63) / 64, (planes * S + 63) / 64, 1);
    hipLaunchKernelGGL((k_gemm_f64<true, 2, 2>), grid, dim3(256), 0, st, in, b1, ctx->tmp_img, planes * S, S, S, S, S, S,
                       (int64_t)0, (int64_t)0, (int64_t)0, states, S, (const double*)nullptr, 0.0);
  } else if (tile == 2) {
    dim3 grid((S + 31) / 32, (planes * S + 63) / 64, 1);
    hipLaunchKernelGGL((k_gemm_f64<true, 2, 1>), grid, dim3(256), 0, st, in, b1, ctx->tmp_img, planes * S, S, S, S, S, S,
                       (int64_t)0, (int64_t)0, (int64_t)0, states, S, (const double*)nullptr, 0.0);
  } else {
    dim3 grid((S + 31) / 32, (planes * S + 31) / 32, 1);
    hipLaunchKernelGGL((k_gemm_f64<true, 1, 1>), grid, dim3(256), 0, st, in, b1, ctx->tmp_img, planes * S, S, S, S, S, S,
                       (int64_t)0, (int64_t)0, (int64_t)0, states, S, (const double*)nullptr, 0.0);
  }
  // pass 2 (along H), per plane: Y[k][w] = sum_n b1[k][n] * T[n][w]
  if (tile == 1) {
    dim3 grid((S + 63) / 64, (S + 63) / 64, planes);
    hipLaunchKernelGGL((k_gemm_f64<false, 2, 2>), grid, dim3(256), 0, st, b2, (const double*)ctx->tmp_img, out, S, S, S, S,
                       S, S, (int64_t)0, (int64_t)S * S, (int64_t)S * S, states, 1 << 30, add, add_scale);
  } else if (tile == 2) {
    dim3 grid((S + 31) / 32, (S + 63) / 64, planes);
    hipLaunchKernelGGL((k_gemm_f64<false, 2, 1>), grid, dim3(256), 0, st, b2, (const double*)ctx->tmp_img, out, S, S, S, S,
                       S, S, (int64_t)0, (int64_t)S * S, (int64_t)S * S, states, 1 << 30, add, add_scale);
  } else {
    dim3 grid((S + 31) / 32, (S + 31) / 32, planes);
    hipLaunchKernelGGL((k_gemm_f64<false, 1, 1>), grid, dim3(256), 0, st, b2, (const double*)ctx->tmp_img, out, S, S, S, S,
                       S, S, (int64_t)0, (int64_t)S * S, (int64_t)S * S, states, 1 << 30, add, add_scale);
  }
  FH_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Representation apply:  out = D.*z + r.*(B (M (B^T (r.*z))))
//   pass 1  k_rep_dots   : block partials of t = B^T (r.*z)
//   pass 1b k_rep_coef   : t = sum of partials (fixed order), c = M t        (one workgroup)
//   pass 2  k_rep_apply2 : out = D.*z + r.*(B c)
// B is read twice; pass 2 walks the rows in the opposite order so that the tail of pass 1 is still
// in the Infinity Cache / L2 when it is needed again.
// ------------------------------------------------------------------------------------------------
typedef double d2v_t __attribute__((ext_vector_type(2)));  // native vector type: usable with the nontemporal builtins
constexpr int kDotRows = 768;   // rows per workgroup (d = 196608 -> 256 workgroups, one per CU)
constexpr int kColChunk = 4;    // columns held in registers per wave and sweep

// pass 1: the 4 waves of a workgroup split the columns (wave w owns columns w, w+4, ...), lanes run along the
// rows with 16-byte loads.  The column loads are branch-free (out-of-range columns re-read column m-1 and are
// dropped at the end) so that a wave has (768/128) x 4 independent 1-KiB loads in flight (8 per wave cost 248 VGPRs,
// two waves per SIMD and 71 us per 403 MB sweep at 8 images; 4 -> four waves per SIMD, 66 us; a pure streaming read of
// the same bytes with this access pattern takes 61 us, profiles/r02_cov_apply_single_sweep.md).  Each wave writes its sums as block
// partials; k_rep_coef (one 1024-thread workgroup) adds them in a fixed order and forms c = M t.  (float64 atomics
// into a shared m-vector were measured 2x SLOWER than this: 8192 same-address adds serialise at the memory side.)
// NT: non-temporal loads of B.  A base that does not fit the 256 MB Infinity Cache anyway (8 images x 32 columns = 403 MB)
// streams 1-4 % faster without allocating there; a smaller one (8 x 16 columns = 201 MB) is re-read from the cache by pass 2
// and by the next CG iteration, and non-temporal loads cost 15-20 % (measured 97 vs 81 us per batched apply at m = 16).
template <bool NT>
__device__ __forceinline__ d2v_t ld_base(const double* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2v_t*>(p));
  return *reinterpret_cast<const d2v_t*>(p);
}

template <bool NT>
__device__ __forceinline__ double ld_base1(const double* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

template <bool NT>
__global__ __launch_bounds__(256) void k_rep_dots(fh_batch per, const double* __restrict__ z,
                                                  double* __restrict__ partial, int64_t d, int m,
                                                  const fh_cg_state* __restrict__ states) {
  const int img = blockIdx.z;
  IMG_GUARD(states, img);
  const double* __restrict__ B = per.B[img];
  const double* __restrict__ r = per.r[img];
  z += (int64_t)img * d;
  partial += (int64_t)img * kPartialRows * FH_MAX_COLS;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kDotRows;
  constexpr int kIter = kDotRows / 128;  // row pairs per lane
  double2 rz[kIter];
  int64_t row[kIter];
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    int64_t i = r0 + it * 128 + 2 * lane;
    const bool ok = i + 1 < d;
    i = ok ? i : 0;  // clamped rows contribute zero
    row[it] = i;
    const double2 zz = *reinterpret_cast<const double2*>(z + i);
    const double2 rr = *reinterpret_cast<const double2*>(r + i);
    rz[it] = ok ? make_double2(zz.x * rr.x, zz.y * rr.y) : make_double2(0.0, 0.0);
  }
  for (int q0 = 0; w + 4 * q0 < m; q0 += kColChunk) {
    double acc[kColChunk];
#pragma unroll
    for (int q = 0; q < kColChunk; ++q) acc[q] = 0.0;
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
      for (int q = 0; q < kColChunk; ++q) {
        const int j = w + 4 * (q0 + q);
        const int jc = j < m ? j : m - 1;
        const d2v_t b = ld_base<NT>(B + (int64_t)jc * d + row[it]);
        acc[q] = fma(b.x, rz[it].x, fma(b.y, rz[it].y, acc[q]));
      }
    }
#pragma unroll
    for (int q = 0; q < kColChunk; ++q) {
      const double v = wave_sum(acc[q]);
      const int j = w + 4 * (q0 + q);
      if (lane == 0 && j < m) partial[(int64_t)blockIdx.x * FH_MAX_COLS + j] = v;
    }
  }
}

// pass 1b: t[j] = sum_b partial[b][j] (fixed order -> deterministic), c = M t.  One workgroup of 1024 threads:
// 32 row-groups x 32 columns keep nparts/32 coalesced loads per thread in flight.
__global__ __launch_bounds__(1024) void k_rep_coef(fh_batch per, const double* __restrict__ partial, int nparts,
                                                   int ldm, int m, double* __restrict__ coef,
                                                   const fh_cg_state* __restrict__ states) {
  const int img = blockIdx.z;
  IMG_GUARD(states, img);
  const double* __restrict__ M = per.M[img];
  partial += (int64_t)img * kPartialRows * FH_MAX_COLS;
  coef += (int64_t)img * 2 * FH_MAX_COLS;
  __shared__ double t[FH_MAX_COLS];
  __shared__ double red[32][33];
  const int tid = threadIdx.x, jj = tid & 31, rg = tid >> 5;
  for (int j0 = 0; j0 < m; j0 += 32) {
    const int j = j0 + jj;
    double s0 = 0.0, s1 = 0.0;
    if (j < m) {
      int b = rg;
      for (; b + 32 < nparts; b += 64) {
        s0 += partial[(int64_t)b * FH_MAX_COLS + j];
        s1 += partial[(int64_t)(b + 32) * FH_MAX_COLS + j];
      }
      for (; b < nparts; b += 32) s0 += partial[(int64_t)b * FH_MAX_COLS + j];
    }
    red[rg][jj] = s0 + s1;
    __syncthreads();
    if (rg == 0 && j < m) {
      double sum = 0.0;
#pragma unroll
      for (int g = 0; g < 32; ++g) sum += red[g][jj];
      t[j] = sum;
      coef[FH_MAX_COLS + j] = sum;  // t = B^T (r.*z) itself (read by the closed-form BFGS inverse of the space update)
    }
    __syncthreads();
  }
  // c = M t : 32 threads per output row (rows in groups of 32), shuffle-reduced
  for (int r0 = 0; r0 < m; r0 += 32) {
    const int rowi = r0 + rg;
    double sum = 0.0;
    if (rowi < m)
      for (int l = jj; l < m; l += 32) sum = fma(M[(int64_t)rowi * ldm + l], t[l], sum);
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    sum += __shfl_xor(sum, 8, 64);
    sum += __shfl_xor(sum, 16, 64);
    if (rowi < m && jj == 0) coef[rowi] = sum;
  }
}

// pass 2: out = D.*z + r.*(B c), rows swept in the opposite order of pass 1 (Infinity-Cache reuse).
template <bool NT>
__global__ __launch_bounds__(256) void k_rep_apply2(fh_batch per, const double* __restrict__ coef,
                                                    const double* __restrict__ z, double* __restrict__ out,
                                                    int64_t d, int m, const fh_cg_state* __restrict__ states) {
  const int img = (int)gridDim.z - 1 - (int)blockIdx.z;  // images too in reverse: the last bases of pass 1 are still in the 256 MB Infinity Cache
  IMG_GUARD(states, img);
  const double* __restrict__ D = per.D[img];
  const double* __restrict__ r = per.r[img];
  const double* __restrict__ B = per.B[img];
  coef += (int64_t)img * 2 * FH_MAX_COLS;
  z += (int64_t)img * d;
  out += (int64_t)img * d;
  const int tid = threadIdx.x;
  __shared__ double c[FH_MAX_COLS];
  for (int j = tid; j < m; j += 256) c[j] = coef[j];
  __syncthreads();
  const int64_t blk = (int64_t)gridDim.x - 1 - blockIdx.x;  // reverse sweep
  // every thread owns three rows of the block (row = block start + 256 q + tid: coalesced 8-byte loads): equal work for the
  // 256 threads, any d (the former row-pair mapping needed an even d and left half of the lanes idle in a second trip)
  constexpr int kR = kDotRows / 256;
  int64_t row[kR];
  bool ok[kR];
#pragma unroll
  for (int q = 0; q < kR; ++q) {
    const int64_t i = blk * kDotRows + 256 * q + tid;
    ok[q] = i < d;
    row[q] = ok[q] ? i : 0;
  }
  // canonical summation order of (B c)_i, shared with the single-sweep kernel k_rep_fused (whose 8 waves each own the
  // columns j = w (mod 8)): eight chains over j = w, w + 8, ... then the chains added in the order w = 0 .. 7
  double ch[8][kR];
#pragma unroll
  for (int w8 = 0; w8 < 8; ++w8)
#pragma unroll
    for (int q = 0; q < kR; ++q) ch[w8][q] = 0.0;
  // the three per-row vectors are requested before the base: their latency hides behind the sweep instead of its tail
  double zq[kR], dq[kR], rq[kR];
#pragma unroll
  for (int q = 0; q < kR; ++q) zq[q] = z[row[q]], dq[q] = D[row[q]], rq[q] = m > 0 ? r[row[q]] : 0.0;
  const int mfull = m & ~7;
  for (int j0 = 0; j0 < mfull; j0 += 8) {  // 24 independent streaming loads per step
    double b[8][kR];
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8)
#pragma unroll
      for (int q = 0; q < kR; ++q) b[w8][q] = ld_base1<NT>(B + (int64_t)(j0 + w8) * d + row[q]);
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) {
      const double cj = c[j0 + w8];
#pragma unroll
      for (int q = 0; q < kR; ++q) ch[w8][q] = fma(b[w8][q], cj, ch[w8][q]);
    }
  }
#pragma unroll
  for (int w8 = 0; w8 < 8; ++w8) {
    const int j = mfull + w8;
    if (j < m) {
      const double cj = c[j];
#pragma unroll
      for (int q = 0; q < kR; ++q) ch[w8][q] = fma(ld_base1<NT>(B + (int64_t)j * d + row[q]), cj, ch[w8][q]);
    }
  }
#pragma unroll
  for (int q = 0; q < kR; ++q) {
    double acc = ch[0][q];
#pragma unroll
    for (int w8 = 1; w8 < 8; ++w8) acc += ch[w8][q];
    const double o = m > 0 ? fma(rq[q], acc, dq[q] * zq[q]) : dq[q] * zq[q];
    if (ok[q]) out[row[q]] = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Single-sweep representation apply (k_rep_fused): B is read ONCE.
// The two-pass form above is bound by its second sweep of the factor base (2 x 8 d m bytes; the base of 8 lock-step
// images, 400 MB, does not fit the 256 MB Infinity Cache).  Here a 512-thread workgroup keeps its 768-row x m slice of
// B in REGISTERS (wave w owns the columns j = w (mod 8): 6 row pairs x MC columns = 48 / 96 doubles per lane at
// m <= 32 / 64) between the reduction t = B^T (r.*z) and the product B (M t):
//   phase 1  slice -> registers, block partials of t (same per-column arithmetic as k_rep_dots), sc1 stores
//   arrive   every wave drains its stores, workgroup barrier, one relaxed agent-scope add on the image's counter
//   wait     thread 0 polls the counter with relaxed sc1 loads + s_sleep (BOUNDED: on time-out the error word of the
//            context is set and the workgroup leaves - every wave reaches its exit)
//   phase 2  every workgroup re-reduces the nb x m partials with sc1 loads (64 KB from L2 / fabric; same fixed order
//            as k_rep_coef), forms c = M t, multiplies its register slice, adds the 8 per-wave chains through LDS in the
//            canonical order and writes out = D.*z + r.*(B c)
//   depart   the last workgroup to finish reading the partials resets the image's two counters
// The hand-off follows the MI355X guide's measured counter form: sc1 (write-through) stores -> s_waitcnt vmcnt(0) in
// every storing wave -> workgroup barrier -> one lane's agent-scope atomic add; consumer: relaxed sc1 poll by one wave,
// workgroup barrier, sc1 loads of every handed-off byte.  No fence, no placement assumption.
// Residency: images are dispatched in order (blockIdx.z slowest), an image's nb <= #CU workgroups fit the chip at once
// (1 or 2 workgroups per CU by register budget) and finished workgroups never wait on later images, so every counter
// completes PROVIDED no other grid-synchronising kernel runs concurrently - the launcher takes this path only on a
// context the caller declared exclusive (fh_context_set_exclusive); everything else keeps the two-pass kernels.
// ------------------------------------------------------------------------------------------------
constexpr int kFusedSpinLimit = 1 << 21;  // x s_sleep(2) ~ 64 clocks: ~ 0.1 s before a workgroup gives up
constexpr int kSyncStride = 64;           // uint32 per image: arrive counter at +0, depart counter at +32 (own 128-B lines)

template <int MC>
__global__ __launch_bounds__(512, MC <= 4 ? 4 : 2) void k_rep_fused(fh_batch per, const double* __restrict__ z,
                                                                    double* __restrict__ out, double* __restrict__ partial,
                                                                    unsigned int* __restrict__ sync, int64_t d, int m,
                                                                    int ldm, const fh_cg_state* __restrict__ states,
                                                                    unsigned long long* __restrict__ dbg) {
  const int img = blockIdx.z;
  IMG_GUARD(states, img);
  // dbg != null (profiling builds of the launcher only): thread 0 records the 100 MHz wall clock at the phase boundaries
#define FUSED_STAMP(k)                                                                                          \
  if (dbg != nullptr && threadIdx.x == 0)                                                                       \
  dbg[((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime()
  FUSED_STAMP(0);
  const double* __restrict__ B = per.B[img];
  const double* __restrict__ r = per.r[img];
  const double* __restrict__ D = per.D[img];
  const double* __restrict__ M = per.M[img];
  z += (int64_t)img * d;
  out += (int64_t)img * d;
  partial += (int64_t)img * kPartialRows * FH_MAX_COLS;
  unsigned int* arrive = sync + img * kSyncStride;
  unsigned int* depart = arrive + 32;
  unsigned int* err = sync + FH_MAX_BATCH * kSyncStride;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nb = gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * kDotRows + 2 * lane;  // d is a multiple of kDotRows on this path
  constexpr int kIter = kDotRows / 128;

  constexpr int MS = 8 * MC;                          // columns this instantiation can hold
  __shared__ __align__(16) double psum[8][kDotRows];  // phase 1: r.*z; phase 2a: reduction scratch; 2c: per-wave chains of (B c)
  __shared__ __align__(16) double dzs[kDotRows], rrs[kDotRows];  // D.*z and r of the workgroup's rows (read before the barrier)
  __shared__ double Msh[MS][MS + 1];                  // the inner matrix, staged before the barrier as well
  __shared__ double tsh[64], csh[64];
  __shared__ int ok_s;
  double(*red)[33] = reinterpret_cast<double(*)[33]>(&psum[1][0]);  // [32][33] doubles inside psum (free in phase 2a)

  // ---- phase 1: r.*z of the workgroup's rows goes through LDS (every wave needs all of it; holding it in registers next
  // to the slice would spill at the 128-VGPR budget of two workgroups per CU)
  double* rzs = &psum[0][0];
  if (tid < kDotRows / 2) {
    const int64_t i0 = (int64_t)blockIdx.x * kDotRows + 2 * tid;
    const double2 zz = *reinterpret_cast<const double2*>(z + i0);
    const double2 rr = *reinterpret_cast<const double2*>(r + i0);
    const double2 dd = *reinterpret_cast<const double2*>(D + i0);
    *reinterpret_cast<double2*>(rzs + 2 * tid) = make_double2(zz.x * rr.x, zz.y * rr.y);
    *reinterpret_cast<double2*>(dzs + 2 * tid) = make_double2(dd.x * zz.x, dd.y * zz.y);
    *reinterpret_cast<double2*>(rrs + 2 * tid) = rr;
  } else {
    // the other two waves stage M (everything the latency chain after the barrier would otherwise have to fetch)
    for (int i = tid - kDotRows / 2; i < m * m; i += 512 - kDotRows / 2) Msh[i / m][i % m] = M[(int64_t)(i / m) * ldm + i % m];
  }
  double2 breg[MC][kIter];
#pragma unroll
  for (int q = 0; q < MC; ++q) {
    const int j = w + 8 * q;  // wave-uniform
    if (j < m) {
      const double* bp = B + (int64_t)j * d + r0;
#pragma unroll
      for (int it = 0; it < kIter; ++it) breg[q][it] = *reinterpret_cast<const double2*>(bp + it * 128);
    } else {
#pragma unroll
      for (int it = 0; it < kIter; ++it) breg[q][it] = make_double2(0.0, 0.0);
    }
  }
  __syncthreads();  // r.*z, D.*z, r and M are in LDS
#pragma unroll
  for (int q = 0; q < MC; ++q) {
    double acc = 0.0;
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      const double2 rz = *reinterpret_cast<const double2*>(rzs + it * 128 + 2 * lane);
      acc = fma(breg[q][it].x, rz.x, fma(breg[q][it].y, rz.y, acc));
    }
    const double v = wave_sum(acc);
    const int j = w + 8 * q;
    if (lane == 0 && j < m)
      __hip_atomic_store(partial + (int64_t)blockIdx.x * FH_MAX_COLS + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory");  // re-read r.*z from LDS for the next column instead of keeping it live
  }
  // ---- arrive + wait
  FUSED_STAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  FUSED_STAMP(2);
  if (tid == 0) {
    __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1, spins = 0;
    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nb) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > kFusedSpinLimit) {
        ok = 0;
        break;
      }
    }
    if (!ok) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok_s = ok;
  }
  __syncthreads();
  if (!ok_s) return;  // uniform: every wave of the workgroup leaves
  FUSED_STAMP(3);

  // ---- phase 2a: t = sum of the block partials, in k_rep_coef's order: row group rg = 0..31 adds the blocks
  // rg, rg + 64, ... into one chain and rg + 32, rg + 96, ... into a second one.  The 16 sc1 loads of a thread are issued
  // back to back (inline asm: the compiler would wait after every atomic load) and drained by ONE s_waitcnt that the
  // results are tied to.
  const int jj = tid & 31, rg2 = tid >> 5;
  for (int j0 = 0; j0 < m; j0 += 32) {
    const int j = j0 + jj;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // scalar base + 32-bit lane offset: one address VGPR per load keeps the batch inside the 128-VGPR budget
      double v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int b = rg2 + 16 * h + 32 * i;
        v[i] = 0.0;
        if (j < m && b < nb) {
          const unsigned int off = (unsigned int)((b * FH_MAX_COLS + j) * (int)sizeof(double));
          asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(v[i]) : "v"(off), "s"(partial) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                   :
                   : "memory");
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < 8; i += 2) s0 += v[i], s1 += v[i + 1];
      red[rg2 + 16 * h][jj] = s0 + s1;
    }
    __syncthreads();
    if (rg2 == 0 && j < m) {
      double sum = 0.0;
#pragma unroll
      for (int g = 0; g < 32; ++g) sum += red[g][jj];
      tsh[j] = sum;
    }
    __syncthreads();
  }
  FUSED_STAMP(4);
  // every partial this workgroup needs has been read: depart; the last one re-arms the image's counters for the next launch
  if (tid == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (unsigned)nb - 1u) {
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- phase 2b: c = M t (32 threads per row, as k_rep_coef)
  for (int q0 = 0; q0 < m; q0 += 16) {
    const int rowi = q0 + rg2;
    double sum = 0.0;
    if (rowi < m)
      for (int l = jj; l < m; l += 32) sum = fma(Msh[rowi][l], tsh[l], sum);
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    sum += __shfl_xor(sum, 8, 64);
    sum += __shfl_xor(sum, 16, 64);
    if (rowi < m && jj == 0) csh[rowi] = sum;
  }
  __syncthreads();
  // ---- phase 2c: this wave's chain over its columns, from registers
  {
    double cq[MC];
#pragma unroll
    for (int q = 0; q < MC; ++q) cq[q] = (w + 8 * q < m) ? csh[w + 8 * q] : 0.0;
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      double2 p = make_double2(0.0, 0.0);
#pragma unroll
      for (int q = 0; q < MC; ++q) {
        p.x = fma(breg[q][it].x, cq[q], p.x);
        p.y = fma(breg[q][it].y, cq[q], p.y);
      }
      *reinterpret_cast<double2*>(&psum[w][it * 128 + 2 * lane]) = p;
    }
  }
  __syncthreads();
  FUSED_STAMP(5);
  // ---- phase 2d: out = D.*z + r.*(chains added in the order w = 0 .. 7)
  if (tid < kDotRows / 2) {
    double2 acc = *reinterpret_cast<const double2*>(&psum[0][2 * tid]);
#pragma unroll
    for (int w8 = 1; w8 < 8; ++w8) {
      const double2 p = *reinterpret_cast<const double2*>(&psum[w8][2 * tid]);
      acc.x += p.x, acc.y += p.y;
    }
    const int64_t i0 = (int64_t)blockIdx.x * kDotRows + 2 * tid;
    const double2 dz = *reinterpret_cast<const double2*>(dzs + 2 * tid);
    const double2 rr = *reinterpret_cast<const double2*>(rrs + 2 * tid);
    *reinterpret_cast<double2*>(out + i0) = make_double2(fma(rr.x, acc.x, dz.x), fma(rr.y, acc.y, dz.y));
  }
  FUSED_STAMP(6);
#undef FUSED_STAMP
}

static int rep_apply_launch(fh_context* ctx, const fh_batch& per, int ldm, const double* z, double* out, int64_t d,
                            int m, const fh_cg_state* states, hipStream_t st) {
  if (m < 0 || m > FH_MAX_COLS || (d & 1) || per.nimg < 1 || per.nimg > ctx->nimg_max) return FH_ESIZE;
  const int nb = (int)((d + kDotRows - 1) / kDotRows);
  if (nb > kPartialRows) return FH_ESIZE;
  // All images of a batch go through pass 1, then pass 2 in reverse image (and row) order, so that the bases read last
  // are re-read first while they are still in the 256 MB Infinity Cache.  Measured at 8 images, m = 32 (algorithmic
  // TB/s): same order 2.75, reverse order 2.97; splitting the batch into cache-sized groups of 4 (each group both
  // passes back to back) 2.80 - the extra dependent launches cost more than the additional hits return.
  const unsigned Z = (unsigned)per.nimg;
  // Single-sweep kernel: only on explicit request (fh_context_set_exclusive(ctx, 2): tests, profiling).  It reads the base once
  // and is bitwise equal, but since the two sweeps run at the per-CU ingest limit it is no longer faster anywhere: one image
  // per launch 26.0-27.0 us vs 25.5 us, eight images 161-177 us vs 139.5 us (the hand-off of one image goes through the same
  // per-CU memory queues as the next image's streaming loads and stretches from 13 us to 19 us, see
  // profiles/r02_cov_apply_single_sweep.md).
  if (out != nullptr && m > 0 && m <= 64 && ctx->exclusive >= 2 && !ctx->fused_disabled && d % kDotRows == 0 &&
      nb <= ctx->num_cus && nb <= 256) {
    unsigned long long* dbg = getenv("FH_FUSED_DEBUG") ? (unsigned long long*)ctx->gpartial : nullptr;  // profiling only
    if (m <= 32)
      hipLaunchKernelGGL((k_rep_fused<4>), dim3(nb, 1, Z), dim3(512), 0, st, per, z, out, ctx->partial, ctx->sync, d, m, ldm,
                         states, dbg);
    else
      hipLaunchKernelGGL((k_rep_fused<8>), dim3(nb, 1, Z), dim3(512), 0, st, per, z, out, ctx->partial, ctx->sync, d, m, ldm,
                         states, dbg);
    FH_LAUNCH_CHECK();
    return 0;
  }
  const bool nt = (int64_t)per.nimg * m * d * (int64_t)sizeof(double) > ((int64_t)220 << 20);  // beyond the Infinity Cache
  if (m > 0) {
    if (nt)
      hipLaunchKernelGGL(k_rep_dots<true>, dim3(nb, 1, Z), dim3(256), 0, st, per, z, ctx->partial, d, m, states);
    else
      hipLaunchKernelGGL(k_rep_dots<false>, dim3(nb, 1, Z), dim3(256), 0, st, per, z, ctx->partial, d, m, states);
    hipLaunchKernelGGL(k_rep_coef, dim3(1, 1, Z), dim3(1024), 0, st, per, (const double*)ctx->partial, nb, ldm, m,
                       ctx->coef, states);
  }
  if (out != nullptr) {  // out == null: only t = B^T (r.*z) and c = M t are wanted (left in ctx->coef[FH_MAX_COLS..] / [0..])
    if (nt)
      hipLaunchKernelGGL(k_rep_apply2<true>, dim3(nb, 1, Z), dim3(256), 0, st, per, (const double*)ctx->coef, z, out, d, m,
                         states);
    else
      hipLaunchKernelGGL(k_rep_apply2<false>, dim3(nb, 1, Z), dim3(256), 0, st, per, (const double*)ctx->coef, z, out, d, m,
                         states);
  }
  FH_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Representation inverse (Woodbury) - diagonal part + weighted Gram  G = B^T diag(rx^2/Dx) B.
//   k_invert_diag : Dx += shift, Dy = 1/Dx, ry = rx/Dx
//   k_gram        : grid (row blocks, 64x64 tile pairs ta<=tb); each workgroup streams its rows in chunks of 64
//                   through LDS; wave w accumulates the 16-row strip w of the pair on v_mfma_f64_16x16x4_f64
//                   (one LDS read per operand per 1024 FMAs; sub-tiles beyond column m are skipped)
//   k_gram_reduce : sums the row-block partials, writes both triangles
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_invert_diag(PtrTab Dxt, PtrTab rxt, double shift, PtrTab Dyt, PtrTab ryt,
                                                     int64_t d) {
  double* __restrict__ Dx = Dxt.p[blockIdx.z];
  const double* __restrict__ rx = rxt.p[blockIdx.z];
  double* __restrict__ Dy = Dyt.p[blockIdx.z];
  double* __restrict__ ry = ryt.p[blockIdx.z];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= d) return;
  const double v = Dx[i] + shift;
  Dx[i] = v;
  Dy[i] = 1.0 / v;
  if (rx != nullptr) ry[i] = rx[i] / v;
}

// forward time shift of a representation and of its inverse: D <- D / (1 + s D), r <- r / (1 + s D), Dinv <- Dinv + s
__global__ __launch_bounds__(256) void k_forward_diag(PtrTab Dt, PtrTab rt_, PtrTab Dinvt, double s, int64_t d) {
  double* __restrict__ D = Dt.p[blockIdx.z];
  double* __restrict__ r = rt_.p[blockIdx.z];
  double* __restrict__ Dinv = Dinvt.p[blockIdx.z];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= d) return;
  const double e = 1.0 / fma(s, D[i], 1.0);
  D[i] *= e;
  r[i] *= e;
  Dinv[i] += s;
}

constexpr int kGramRowBlocks = 512;  // 384 rows each at d = 196608: two workgroups per CU
constexpr int kGT = 64;  // Gram tile edge and row-chunk length

// fwd = 0: weights rx^2 / Dx (Woodbury inverse of the representation); fwd = 1: rx^2 / (1 + fshift Dx) (forward time
// shift of the representation itself, see cov_shift_forward)
__global__ __launch_bounds__(256) void k_gram(PtrTab Bt, PtrTab rxt, PtrTab Dxt, double* __restrict__ gpartial,
                                              int64_t gstride, int64_t d, int m, int ntiles, int fwd, double fshift) {
  const double* __restrict__ B = Bt.p[blockIdx.z];   // grid z = image
  const double* __restrict__ rx = rxt.p[blockIdx.z];
  const double* __restrict__ Dx = Dxt.p[blockIdx.z];
  gpartial += (int64_t)blockIdx.z * gstride;
  __shared__ double As[kGT][kGT + 1];  // [row i][col of tile a], weighted
  __shared__ double Bs[kGT][kGT + 1];  // [row i][col of tile b]
  // decode tile pair
  int ta = 0, tb = 0, p = blockIdx.y;
  for (ta = 0; ta < ntiles; ++ta) {
    const int cnt = ntiles - ta;
    if (p < cnt) {
      tb = ta + p;
      break;
    }
    p -= cnt;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  // 16 x 16 sub-tiles of the 64 x 64 pair that hold columns < m; wave w owns sub-tile row w (f64 MFMA 16x16x4:
  // a = A[i = l & 15][k = l >> 4], b = B[k = l >> 4][j = l & 15], C: col = l & 15, row = (l >> 4) + 4 reg)
  const int ma = m - ta * kGT < kGT ? m - ta * kGT : kGT, mb = m - tb * kGT < kGT ? m - tb * kGT : kGT;
  const int na = (ma + 15) / 16, nbt = (mb + 15) / 16;
  const bool active = wave < na;
  const int64_t rows_per = ((d + gridDim.x - 1) / gridDim.x + kGT - 1) / kGT * kGT;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per;
  const int64_t r1 = r0 + rows_per < d ? r0 + rows_per : d;
  double4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  // On the diagonal pair only sub-tiles on or above the diagonal are formed (G is symmetric; k_gram_reduce mirrors).
  const int j_lo = ta == tb ? wave : 0;
  const int row = tid & 63;
  for (int64_t c0 = r0; c0 < r1; c0 += kGT) {
    // all loads of the chunk are issued before the first LDS store; columns beyond the last live sub-tile are not
    // fetched and the diagonal pair loads its columns once.  (Prefetching chunk c + 1 into registers across the MFMA
    // phase was measured slower: the allocator then needs > 256 VGPRs and the kernel drops to one workgroup per CU.)
    const int64_t gi = c0 + row;
    const bool rok = gi < r1;
    const double wgt = rok ? rx[gi] * rx[gi] / (fwd ? fma(fshift, Dx[gi], 1.0) : Dx[gi]) : 0.0;
    double va[16], vb[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int col = e * 4 + (tid >> 6);
      const int ja = ta * kGT + col, jb = tb * kGT + col;
      vb[e] = (col < 16 * nbt && rok && jb < m) ? B[(int64_t)jb * d + gi] : 0.0;
      va[e] = ta == tb ? vb[e] : ((col < 16 * na && rok && ja < m) ? B[(int64_t)ja * d + gi] : 0.0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int col = e * 4 + (tid >> 6);
      if (col < 16 * na) As[row][col] = va[e] * wgt;
      if (col < 16 * nbt) Bs[row][col] = vb[e];
    }
    __syncthreads();
    if (active) {
#pragma unroll 4
      for (int kk = 0; kk < kGT; kk += 4) {
        const double av = As[kk + lk][16 * wave + li];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j >= j_lo && j < nbt)
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[kk + lk][16 * j + li], acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (!active) return;
  double* dst = gpartial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (kGT * kGT);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < j_lo || j >= nbt) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(16 * wave + lk + 4 * r) * kGT + 16 * j + li] = acc[j][r];
  }
}
__global__ __launch_bounds__(256) void k_gram_reduce(const double* __restrict__ gpartial, int64_t gstride, int nrb,
                                                     int ntiles, int m, PtrTab Gt, int ldg) {
  double* __restrict__ G = Gt.p[blockIdx.z];
  gpartial += (int64_t)blockIdx.z * gstride;
  // grid (tile pairs, 64): a workgroup owns 64 consecutive tile outputs; its 4 waves each sum a quarter of the nrb
  // row-block partials (coalesced 512-byte rows, 4 independent chains), then the quarters are added in a fixed order.
  int ta = 0, tb = 0, p = blockIdx.x;
  for (ta = 0; ta < ntiles; ++ta) {
    const int cnt = ntiles - ta;
    if (p < cnt) {
      tb = ta + p;
      break;
    }
    p -= cnt;
  }
  __shared__ double red[4][64];
  const double* src = gpartial + (int64_t)blockIdx.x * nrb * (kGT * kGT);
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int o_out = blockIdx.y * 64 + lane;
  const int a = ta * kGT + o_out / kGT, c = tb * kGT + o_out % kGT;
  const bool live = a < m && c < m;  // k_gram writes only sub-tiles that hold columns < m
  // diagonal pair: sub-tiles below the diagonal were not formed - read the mirrored entry
  const bool mirror = ta == tb && (o_out / kGT) / 16 > (o_out % kGT) / 16;
  const int o = mirror ? (o_out % kGT) * kGT + o_out / kGT : o_out;
  const int per = (nrb + 3) / 4;
  const int b0 = q * per, b1 = b0 + per < nrb ? b0 + per : nrb;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (live) {
    int b = b0;
    for (; b + 3 < b1; b += 4) {
      s0 += src[(int64_t)b * (kGT * kGT) + o];
      s1 += src[(int64_t)(b + 1) * (kGT * kGT) + o];
      s2 += src[(int64_t)(b + 2) * (kGT * kGT) + o];
      s3 += src[(int64_t)(b + 3) * (kGT * kGT) + o];
    }
    for (; b < b1; ++b) s0 += src[(int64_t)b * (kGT * kGT) + o];
  }
  red[q][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && live) {
    const double s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    G[(int64_t)a * ldg + c] = s;
    if (ta != tb) G[(int64_t)c * ldg + a] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// Elementwise helpers and scalar reductions
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_axpby(double alpha, const double* __restrict__ a, double beta,
                                               const double* __restrict__ b, double* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double v = alpha * a[i];
    if (b != nullptr) v = fma(beta, b[i], v);
    out[i] = v;
  }
}

// out = mask .* in  (add == null)   |   out = add_scale*add + mask .* in      (grid z = image)
__global__ __launch_bounds__(256) void k_mask(fh_batch per, const double* __restrict__ in,
                                              const double* __restrict__ add, double add_scale,
                                              double* __restrict__ out, int64_t n,
                                              const fh_cg_state* __restrict__ states) {
  const int img = blockIdx.z;
  IMG_GUARD(states, img);
  const double* __restrict__ mask = per.mask[img];
  in += (int64_t)img * n;
  out += (int64_t)img * n;
  if (add != nullptr) add += (int64_t)img * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double v = mask[i] * in[i];
    if (add != nullptr) v = fma(add_scale, add[i], v);
    out[i] = v;
  }
}

constexpr int kDotBlocks = 256;
constexpr int kCgBlocks = 192;
constexpr int kCgScratch = 4 * kDotBlocks + 16;  // doubles of w2 per image: [pAp | b.b | r.r | - | rzbuf[2]]

// partial[b] = sum over block b of a.*b ; optional second product a2.*b2 -> partial[kDotBlocks + b]   (grid z = image)
__global__ __launch_bounds__(256) void k_dot_partial(const double* __restrict__ a, const double* __restrict__ b,
                                                     const double* __restrict__ a2, const double* __restrict__ b2,
                                                     double* __restrict__ part, int64_t n, int part_stride,
                                                     const fh_cg_state* __restrict__ states) {
  const int img = blockIdx.z;
  IMG_GUARD(states, img);
  a += (int64_t)img * n;
  b += (int64_t)img * n;
  part += (int64_t)img * part_stride;
  __shared__ double red[4];
  double s = 0.0, s2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    s = fma(a[i], b[i], s);
    if (a2 != nullptr) s2 = fma(a2[i], b2[i], s2);
  }
  s = block_sum_256(s, red);
  if (a2 != nullptr) s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = s;
    if (a2 != nullptr) part[kDotBlocks + blockIdx.x] = s2;
  }
}

// grid x = image: dst[img][slot] = sum of that image's partials
__global__ __launch_bounds__(256) void k_scalar_reduce(const double* __restrict__ part, int64_t part_stride, int nparts,
                                                       PtrTab dst, int slot) {
  __shared__ double red[4];
  part += (int64_t)blockIdx.x * part_stride;
  double s = threadIdx.x < nparts ? part[threadIdx.x] : 0.0;
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) dst.p[blockIdx.x][slot] = s;
}

__global__ __launch_bounds__(256) void k_space_prep(const double* __restrict__ dm, double s2,
                                                    const double* __restrict__ dx, double* __restrict__ de,
                                                    double* __restrict__ part, int64_t part_stride, int64_t n) {
  __shared__ double red[4];
  dm += (int64_t)blockIdx.z * n;  // grid z = image: the vectors of a batch are contiguous [image][n]
  dx += (int64_t)blockIdx.z * n;
  de += (int64_t)blockIdx.z * n;
  part += (int64_t)blockIdx.z * part_stride;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double e = s2 * dm[i];
    de[i] = e;
    s = fma(dx[i], e, s);
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// scal != null: gamma = 1 / scal[0] (dx.de) and q = scal[1] (dx.C dx) are read on the device, and workgroup 0 appends the
// pair to the inner matrices: Mc gets diag(gamma, -1/q) at (mc, mc+1) unless projected, Mh diag(gamma, -1/q) / s2^2 at
// (mh, mh+1); the new rows / columns are cleared first.  No scalar returns to the host.
__global__ __launch_bounds__(256) void k_space_commit(const double* __restrict__ de, const double* __restrict__ cdx,
                                                      double gamma, double q, double s2, PtrTab Dct, PtrTab rct, PtrTab bc0t,
                                                      PtrTab bc1t, PtrTab Dht, PtrTab rht, PtrTab bh0t, PtrTab bh1t,
                                                      int project, int64_t n, int use_scal, PtrTab scalt, PtrTab Mct, int ldc,
                                                      int mc, PtrTab Mht, int ldh, int mh) {
  const int img = blockIdx.z;  // grid z = image; de, cdx contiguous [image][n]
  de += (int64_t)img * n;
  cdx += (int64_t)img * n;
  double* __restrict__ Dc = Dct.p[img];
  const double* __restrict__ rc = rct.p[img];
  double* __restrict__ bc0 = bc0t.p[img];
  double* __restrict__ bc1 = bc1t.p[img];
  double* __restrict__ Dh = Dht.p[img];
  const double* __restrict__ rh = rht.p[img];
  double* __restrict__ bh0 = bh0t.p[img];
  double* __restrict__ bh1 = bh1t.p[img];
  if (use_scal) {
    const double* __restrict__ scal = scalt.p[img];
    double* __restrict__ Mc = Mct.p[img];
    double* __restrict__ Mh = Mht.p[img];
    gamma = 1.0 / scal[0];
    q = scal[1];
    if (blockIdx.x == 0) {
      const int t = threadIdx.x;
      if (!project && Mc != nullptr) {
        for (int i = t; i < mc + 2; i += 256)
          for (int e = 0; e < 2; ++e) Mc[(int64_t)(mc + e) * ldc + i] = 0.0, Mc[(int64_t)i * ldc + mc + e] = 0.0;
      }
      if (Mh != nullptr) {
        for (int i = t; i < mh + 2; i += 256)
          for (int e = 0; e < 2; ++e) Mh[(int64_t)(mh + e) * ldh + i] = 0.0, Mh[(int64_t)i * ldh + mh + e] = 0.0;
      }
      __syncthreads();
      if (t == 0) {
        if (!project && Mc != nullptr) {
          Mc[(int64_t)mc * ldc + mc] = gamma;
          Mc[(int64_t)(mc + 1) * ldc + mc + 1] = -1.0 / q;
        }
        if (Mh != nullptr) {
          Mh[(int64_t)mh * ldh + mh] = gamma / (s2 * s2);
          Mh[(int64_t)(mh + 1) * ldh + mh + 1] = -1.0 / (q * s2 * s2);
        }
      }
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double e = de[i], c = cdx[i];
    double dc = Dc[i];
    if (project) {
      dc = dc + gamma * e * e - c * c / q;
      Dc[i] = dc;
    } else {
      const double s = (rc != nullptr) ? rc[i] : 1.0;
      bc0[i] = e / s;
      bc1[i] = c / s;
    }
    Dh[i] = (dc / s2 - 1.0) / s2;
    const double sh = (rh != nullptr) ? rh[i] : 1.0;
    bh0[i] = e / sh;
    bh1[i] = c / sh;
  }
}

constexpr int kWbMaxCols = 64;  // = kWbMax below: the device-side m x m algebra handles up to 64 columns

// part[b] = sum over block b of a^2 .* w   (de^T D^-1 de of the closed-form BFGS inverse)
__global__ __launch_bounds__(256) void k_wnorm_partial(const double* __restrict__ a, PtrTab wt,
                                                       double* __restrict__ part, int64_t part_stride, int64_t n) {
  __shared__ double red[4];
  const double* __restrict__ w = wt.p[blockIdx.z];  // grid z = image; a contiguous [image][n]
  a += (int64_t)blockIdx.z * n;
  part += (int64_t)blockIdx.z * part_stride;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s = fma(a[i] * a[i], w[i], s);
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_copy_small(const double* __restrict__ src, int64_t src_stride, PtrTab dstt, int n) {
  src += (int64_t)blockIdx.x * src_stride;  // grid x = image
  double* __restrict__ dst = dstt.p[blockIdx.x];
  for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// Inverse covariance after a BFGS pair, in closed form (no inversion, no Gram):
//   C'^-1 = (I - g dx de^T) C^-1 (I - g de dx^T) + g dx dx^T,   g = 1 / (dx.de)
// In the coordinates of the shared base (columns 0..mc-1 old, mc = de / r, mc+1 = C dx / r; C^-1 = 1/D + Wi Mi Wi^T with
// Wi = (r / D) .* B):  dx = Wi' a,  C^-1 de = Wi' b  with  a = [-M t; 0; 1],  b = [Mi s; 1; 0],  t = W^T dx,  s = Wi^T de, so
//   Mi' = pad(Mi) - g (a b^T + b a^T) + (g^2 rho + g) a a^T,   rho = de^T C^-1 de = de^T D^-1 de + s^T Mi s.
// The generic Woodbury route passes through the singular matrix C - v v^T (v = C dx / sqrt(dx^T C dx)) and loses
// cond(I + G M) ~ 1e9 of the 1e16 at d = 196608 with the DCT prior; this form has no cancellation.
// scal: [0] dx.de, [2] de^T D^-1 de;  cC = M t (mc);  sCi = s, cCi = Mi s (mc).  One workgroup.
__global__ __launch_bounds__(256) void k_bfgs_inverse_commit(PtrTab scalt, PtrTab cCt, const double* __restrict__ coef,
                                                             int64_t coef_stride, PtrTab Mit, int ld, int mc) {
  const double* __restrict__ scal = scalt.p[blockIdx.x];  // grid x = image
  const double* __restrict__ cC = cCt.p[blockIdx.x];
  const double* __restrict__ cCi = coef + (int64_t)blockIdx.x * coef_stride;  // c = Mi s of the dots pass ...
  const double* __restrict__ sCi = cCi + FH_MAX_COLS;                         // ... and s = Wi^T de itself
  double* __restrict__ Mi = Mit.p[blockIdx.x];
  __shared__ double a[kWbMaxCols + 2], b[kWbMaxCols + 2];
  __shared__ double rho_s;
  const int tid = threadIdx.x, n = mc + 2;
  for (int j = tid; j < n; j += 256) {
    a[j] = j < mc ? -cC[j] : (j == mc ? 0.0 : 1.0);
    b[j] = j < mc ? cCi[j] : (j == mc ? 1.0 : 0.0);
  }
  if (tid == 0) {
    double rho = scal[2];
    for (int j = 0; j < mc; ++j) rho = fma(sCi[j], cCi[j], rho);  // fixed order
    rho_s = rho;
  }
  __syncthreads();
  const double g = 1.0 / scal[0], k2 = fma(g * g, rho_s, g);
  for (int i = tid; i < n * n; i += 256) {
    const int r = i / n, c = i % n;
    const double old = (r < mc && c < mc) ? Mi[(int64_t)r * ld + c] : 0.0;
    Mi[(int64_t)r * ld + c] = old - g * (a[r] * b[c] + b[r] * a[c]) + k2 * a[r] * a[c];
  }
}

// ------------------------------------------------------------------------------------------------
// Circular convolution with a sparse tap list.
//   k_conv_tile   : stride-1 output grid (blur, blur^T, and blur^T of a zero-inserted LR image):
//                   16x32 output tile + halo staged in LDS (<= 56 KiB at halo 30), 2 outputs per
//                   thread, taps read through the scalar path (uniform index).
//   k_conv_direct : decimated forward (SR: out[i][j] sampled at (i*stride, j*stride)).
// Epilogue: out = acc + add_scale * add  when add != null  (the sigma_y^2 u term of A_mm).
// ------------------------------------------------------------------------------------------------
constexpr int kMaxTaps = 1024;

__global__ __launch_bounds__(256) void k_conv_tile(const double* __restrict__ in, double* __restrict__ out,
                                                   const int* __restrict__ tdy, const int* __restrict__ tdx,
                                                   const double* __restrict__ tw, int ntaps, int S, int halo,
                                                   int halo_x, int adjoint, int up, const double* __restrict__ add,
                                                   double add_scale, const fh_cg_state* __restrict__ states) {
  IMG_GUARD(states, blockIdx.z / 3);
  extern __shared__ __align__(16) double tile[];  // [sh*sw] tile | [ntaps] weights | [ntaps] int offsets
  const int sw = 32 + 2 * halo_x, sh = 16 + 2 * halo;  // halo = rows (dy extent), halo_x = columns (dx extent)
  double* s_w = tile + sw * sh;
  int* s_off = reinterpret_cast<int*>(s_w + ntaps);
  const int plane = blockIdx.z;
  const int oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 32;
  const int Sin = S / up;
  const double* src = in + (int64_t)plane * Sin * Sin;
  const int sgn = adjoint ? 1 : -1;
  for (int t = threadIdx.x; t < ntaps; t += 256) {  // taps -> LDS: the inner loop then needs no scalar-memory waits
    s_w[t] = tw[t];
    s_off[t] = sgn * (tdy[t] * sw + tdx[t]);
  }
  for (int idx = threadIdx.x; idx < sw * sh; idx += 256) {
    const int ly = idx / sw, lx = idx % sw;
    int gy = (oy0 + ly - halo) % S, gx = (ox0 + lx - halo_x) % S;
    gy += gy < 0 ? S : 0;
    gx += gx < 0 ? S : 0;
    double v;
    if (up == 1) {
      v = src[(int64_t)gy * S + gx];
    } else {
      v = ((gy % up) == 0 && (gx % up) == 0) ? src[(int64_t)(gy / up) * Sin + gx / up] : 0.0;
    }
    tile[idx] = v;
  }
  __syncthreads();
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;  // ly in [0,8): rows ly, ly+8
  double a0 = 0, a1 = 0;
  const double* q0 = tile + (ly + halo) * sw + lx + halo_x;
  const double* q1 = q0 + 8 * sw;
  int t = 0;
  for (; t + 3 < ntaps; t += 4) {
    const int o0 = s_off[t], o1 = s_off[t + 1], o2 = s_off[t + 2], o3 = s_off[t + 3];
    const double w0 = s_w[t], w1 = s_w[t + 1], w2 = s_w[t + 2], w3 = s_w[t + 3];
    a0 = fma(w0, q0[o0], a0);
    a1 = fma(w0, q1[o0], a1);
    a0 = fma(w1, q0[o1], a0);
    a1 = fma(w1, q1[o1], a1);
    a0 = fma(w2, q0[o2], a0);
    a1 = fma(w2, q1[o2], a1);
    a0 = fma(w3, q0[o3], a0);
    a1 = fma(w3, q1[o3], a1);
  }
  for (; t < ntaps; ++t) {
    a0 = fma(s_w[t], q0[s_off[t]], a0);
    a1 = fma(s_w[t], q1[s_off[t]], a1);
  }
  const int ox = ox0 + lx;
  if (ox >= S) return;
  double acc[2] = {a0, a1};
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int oy = oy0 + ly + 8 * e;
    if (oy < S) {
      const int64_t o = (int64_t)plane * S * S + (int64_t)oy * S + ox;
      double v = acc[e];
      if (add != nullptr) v = fma(add_scale, add[o], v);
      out[o] = v;
    }
  }
}

// k_conv_tile8: the stride-1 case with 8 outputs per thread - a 64 x 32 output tile, thread = 2 columns (16 apart) x 4 rows
// (16 apart).  k_conv_tile issues 4 LDS instructions per 2 multiply-adds (offset, weight, two inputs) and is bound by
// the LDS instruction rate (10 % of the f64 FMA rate on the 217-tap motion PSF); here a tap costs one 16-byte broadcast
// read (weight + offset packed) and four two-address reads for 8 multiply-adds, and the halo is amortised over 4x the
// outputs (2.9x the tile instead of 6.9x at the motion PSF's 58 x 16 extent).  Same tap order per output: same result.
constexpr int kT8W = 32, kT8H = 64;

__global__ __launch_bounds__(256) void k_conv_tile8(const double* __restrict__ in, double* __restrict__ out,
                                                    const int* __restrict__ tdy, const int* __restrict__ tdx,
                                                    const double* __restrict__ tw, int ntaps, int S, int halo,
                                                    int halo_x, int adjoint, const double* __restrict__ add,
                                                    double add_scale, const fh_cg_state* __restrict__ states) {
  IMG_GUARD(states, blockIdx.z / 3);
  extern __shared__ __align__(16) double tile[];  // [sh*sw (+1)] tile | [ntaps] (weight, offset) pairs
  const int sw = kT8W + 2 * halo_x, sh = kT8H + 2 * halo;
  double2* s_tap = reinterpret_cast<double2*>(tile + ((sw * sh + 1) & ~1));
  const int plane = blockIdx.z;
  const int oy0 = blockIdx.y * kT8H, ox0 = blockIdx.x * kT8W;
  const double* src = in + (int64_t)plane * S * S;
  const int sgn = adjoint ? 1 : -1;
  for (int t = threadIdx.x; t < ntaps; t += 256)
    s_tap[t] = make_double2(tw[t], __longlong_as_double((long long)(sgn * (tdy[t] * sw + tdx[t]))));
  for (int ly = threadIdx.x >> 6; ly < sh; ly += 4) {  // a wave per tile row: no division by the run-time row pitch
    int gy = (oy0 + ly - halo) % S;
    gy += gy < 0 ? S : 0;
    const double* srow = src + (int64_t)gy * S;
    for (int lx = threadIdx.x & 63; lx < sw; lx += 64) {
      int gx = (ox0 + lx - halo_x) % S;
      gx += gx < 0 ? S : 0;
      tile[ly * sw + lx] = srow[gx];
    }
  }
  __syncthreads();
  // thread = columns tx and tx + 16, rows ty + 16 r: 16 lanes read 16 consecutive doubles (all 32 banks of a half), the next
  // 16 lanes the next tile row - adjacent column PAIRS per lane would leave half of the banks idle in every pass
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const double* q = tile + (ty + halo) * sw + tx + halo_x;
  const int rstep = 16 * sw;
  double a[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r][0] = a[r][1] = 0.0;
#pragma unroll 2
  for (int t = 0; t < ntaps; ++t) {
    const double2 tp = s_tap[t];
    const double* p = q + (int)__double_as_longlong(tp.y);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a[r][0] = fma(tp.x, p[r * rstep], a[r][0]);
      a[r][1] = fma(tp.x, p[r * rstep + 16], a[r][1]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oy = oy0 + ty + 16 * r;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ox = ox0 + tx + 16 * e;
      if (oy < S && ox < S) {
        const int64_t o = (int64_t)plane * S * S + (int64_t)oy * S + ox;
        double v = a[r][e];
        if (add != nullptr) v = fma(add_scale, add[o], v);
        out[o] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_conv_direct(const double* __restrict__ in, double* __restrict__ out,
                                                     const int* __restrict__ tdy, const int* __restrict__ tdx,
                                                     const double* __restrict__ tw, int ntaps, int S, int stride,
                                                     int planes, const double* __restrict__ add, double add_scale,
                                                     const fh_cg_state* __restrict__ states) {
  const int So = S / stride;
  const int64_t total = (int64_t)planes * So * So;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= total) return;
  const int plane = (int)(o / ((int64_t)So * So));
  IMG_GUARD(states, plane / 3);
  const int rem = (int)(o % ((int64_t)So * So));
  const int oy = (rem / So) * stride, ox = (rem % So) * stride;
  const double* src = in + (int64_t)plane * S * S;
  double acc = 0.0;
  for (int t = 0; t < ntaps; ++t) {
    int gy = (oy - tdy[t]) % S, gx = (ox - tdx[t]) % S;
    gy += gy < 0 ? S : 0;
    gx += gx < 0 ? S : 0;
    acc = fma(tw[t], src[(int64_t)gy * S + gx], acc);
  }
  if (add != nullptr) acc = fma(add_scale, add[o], acc);
  out[o] = acc;
}

// ------------------------------------------------------------------------------------------------
// Super-resolution operator pair at full rate (stride s = scale factor, 25 x 25 bicubic PSF, not separable):
//   k_conv_dec : out[i][j] = sum_t w_t in[s i - dy_t][s j - dx_t]          (blur, then keep every s-th sample)
//   k_conv_up  : out[y][x] = sum_t w_t z[y + dy_t][x + dx_t],  z = u with s - 1 zeros inserted   (its adjoint)
// k_conv_direct fetched every operand of its 625 taps per output through the vector memory path (131 us per call over 24
// planes), and k_conv_tile with `up` multiplied the inserted zeros (15 of 16 taps at s = 4; 107 us).  Here the input tile
// of k_conv_dec sits in LDS de-interleaved by column phase (x = s X + px), so that 16 neighbouring outputs read 16
// consecutive doubles for every tap; k_conv_up sorts the taps by output phase (py, px) = (-dy mod s, -dx mod s) once per
// workgroup - one wave, ballots + prefix counts, list order preserved - and every output then visits only the ~ 1/s^2 of
// the taps that meet a sample of u.  Both keep the tap order of the kernels they replace (skipping exact zeros): the
// results are bitwise the same.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_dec(const double* __restrict__ in, double* __restrict__ out,
                                                  const int* __restrict__ tdy, const int* __restrict__ tdx,
                                                  const double* __restrict__ tw, int ntaps, int S, int s, int hy, int hx,
                                                  const double* __restrict__ add, double add_scale,
                                                  const fh_cg_state* __restrict__ states) {
  IMG_GUARD(states, blockIdx.z / 3);
  extern __shared__ __align__(16) double tile[];  // [s][R][Wp] input tile by column phase | [ntaps] (weight, offset)
  const int So = S / s, hq = (hx + s - 1) / s, hx4 = hq * s;
  const int R = 15 * s + 1 + 2 * hy, Wp = 16 + 2 * hq + 1;
  double2* s_tap = reinterpret_cast<double2*>(tile + ((s * R * Wp + 1) & ~1));
  const int plane = blockIdx.z, oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
  const double* src = in + (int64_t)plane * S * S;
  for (int t = threadIdx.x; t < ntaps; t += 256) {
    const int c = hx4 - tdx[t];  // column of the tap's operand for output column 0, >= 0
    s_tap[t] = make_double2(tw[t], __longlong_as_double((long long)(((c % s) * R + hy - tdy[t]) * Wp + c / s)));
  }
  const int gy0 = oy0 * s - hy, gx0 = ox0 * s - hx4, ncol = s * (16 + 2 * hq);
  for (int r = threadIdx.x >> 6; r < R; r += 4) {
    int gy = (gy0 + r) % S;
    gy += gy < 0 ? S : 0;
    const double* srow = src + (int64_t)gy * S;
    for (int c = threadIdx.x & 63; c < ncol; c += 64) {
      int gx = (gx0 + c) % S;
      gx += gx < 0 ? S : 0;
      tile[((c % s) * R + r) * Wp + c / s] = srow[gx];
    }
  }
  __syncthreads();
  const int tj = threadIdx.x & 15, ti = threadIdx.x >> 4;
  const double* q = tile + (s * ti) * Wp + tj;
  double acc = 0.0;
#pragma unroll 4
  for (int t = 0; t < ntaps; ++t) {
    const double2 tp = s_tap[t];
    acc = fma(tp.x, q[(int)__double_as_longlong(tp.y)], acc);
  }
  const int oy = oy0 + ti, ox = ox0 + tj;
  if (oy < So && ox < So) {
    const int64_t o = (int64_t)plane * So * So + (int64_t)oy * So + ox;
    if (add != nullptr) acc = fma(add_scale, add[o], acc);
    out[o] = acc;
  }
}

constexpr int kUpMaxPhases = 16;  // s <= 4

__global__ __launch_bounds__(256) void k_conv_up(const double* __restrict__ in, double* __restrict__ out,
                                                 const int* __restrict__ tdy, const int* __restrict__ tdx,
                                                 const double* __restrict__ tw, int ntaps, int S, int s, int hy, int hx,
                                                 int cap, const double* __restrict__ add, double add_scale,
                                                 const fh_cg_state* __restrict__ states) {
  IMG_GUARD(states, blockIdx.z / 3);
  extern __shared__ __align__(16) double lds_up[];  // [s*s][cap] (weight, offset) | [Hu][Wu] tile of u | [16 s][16 s + 1] outputs
  const int Sin = S / s, hqy = (hy + s - 1) / s, hqx = (hx + s - 1) / s;
  const int Hu = 16 + 2 * hqy, Wu = 16 + 2 * hqx + 1, T = 16 * s, SW = T + 1;
  double2* lists = reinterpret_cast<double2*>(lds_up);
  double* utile = lds_up + 2 * s * s * cap;
  double* stage = utile + ((Hu * Wu + 1) & ~1);
  __shared__ int cnt_s[kUpMaxPhases];
  const int plane = blockIdx.z, tid = threadIdx.x;
  const double* src = in + (int64_t)plane * Sin * Sin;
  const int a0 = blockIdx.y * 16 - hqy, b0 = blockIdx.x * 16 - hqx;
  for (int idx = tid; idx < Hu * (Wu - 1); idx += 256) {
    const int r = idx / (Wu - 1), c = idx % (Wu - 1);
    int ga = (a0 + r) % Sin, gb = (b0 + c) % Sin;
    ga += ga < 0 ? Sin : 0;
    gb += gb < 0 ? Sin : 0;
    utile[r * Wu + c] = src[(int64_t)ga * Sin + gb];
  }
  if (tid < 64) {  // one wave sorts the taps by phase, keeping the list order inside every phase
    const int lane = tid;
    int cnt[kUpMaxPhases];
#pragma unroll
    for (int p = 0; p < kUpMaxPhases; ++p) cnt[p] = 0;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int t0 = 0; t0 < ntaps; t0 += 64) {
      const int t = t0 + lane;
      const bool valid = t < ntaps;
      const int dy = valid ? tdy[t] : 0, dx = valid ? tdx[t] : 0;
      const int py = ((-dy) % s + s) % s, px = ((-dx) % s + s) % s;
      const int ph = valid ? py * s + px : -1;
      const double2 item = make_double2(valid ? tw[t] : 0.0,
                                        __longlong_as_double((long long)(((py + dy) / s) * Wu + (px + dx) / s)));
#pragma unroll
      for (int p = 0; p < kUpMaxPhases; ++p) {
        const unsigned long long mask = __ballot(ph == p);
        if (ph == p) {
          const int pos = cnt[p] + __popcll(mask & lt);
          if (pos < cap) lists[p * cap + pos] = item;
        }
        cnt[p] += __popcll(mask);
      }
    }
    if (lane < kUpMaxPhases) {
      int c = 0;
#pragma unroll
      for (int p = 0; p < kUpMaxPhases; ++p) c = lane == p ? cnt[p] : c;
      cnt_s[lane] = c < cap ? c : cap;
    }
  }
  __syncthreads();
  const int la = tid >> 4, lb = tid & 15;
  const double* q = utile + (la + hqy) * Wu + lb + hqx;
  for (int p = 0; p < s * s; ++p) {
    const int n = cnt_s[p];
    const double2* lp = lists + p * cap;
    double acc = 0.0;
    for (int t = 0; t < n; ++t) {
      const double2 tp = lp[t];
      acc = fma(tp.x, q[(int)__double_as_longlong(tp.y)], acc);
    }
    stage[(s * la + p / s) * SW + s * lb + p % s] = acc;
  }
  __syncthreads();
  const int oy0 = blockIdx.y * T, ox0 = blockIdx.x * T;
  for (int idx = tid; idx < T * T; idx += 256) {
    const int r = idx / T, c = idx % T;
    const int oy = oy0 + r, ox = ox0 + c;
    if (oy < S && ox < S) {
      const int64_t o = (int64_t)plane * S * S + (int64_t)oy * S + ox;
      double v = stage[r * SW + c];
      if (add != nullptr) v = fma(add_scale, add[o], v);
      out[o] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The m x m part of a Woodbury step / forward time shift on the device (online_update_bfgs.py:87-119 in the real form of
// this build):   Mdst = sym( sign * Msrc (I + alpha G Msrc)^-1 ),   (alpha, sign) = (1, -1): Woodbury inverse; (s, +1): shift.
// One workgroup, matrices in LDS.  A = I + alpha G Msrc is NOT well conditioned: the factor columns of a trajectory are
// nearly dependent (balanced Gram matrices with eigenvalues 1e-10 .. 7 at k = 8) and G Msrc is far from normal - measured
// cond(A) ~ 1e12 at 256 x 256, i.e. a plain float64 elimination leaves 1e-4 in Mdst and 1e-6 .. 1e-4 in the covariance at
// sigma < 0.2.  So: A^-1 by Gauss-Jordan with partial pivoting in float64, X0 = sign Msrc A^-1, then up to kWbRefine steps of
// iterative refinement  X <- X + (sign Msrc - X A) A^-1  with A = I + alpha G Msrc and the residual accumulated in
// double-double (error-free products by fma, two-sum accumulation): converges to the float64 rounding of the exact
// m x m result (contraction ~ cond(A) x 1e-16 per step), 3e-6 -> 6e-9 in the covariance on the measured states.
// No host round trip: the covariance updates of a guidance call stay on the device.
// ------------------------------------------------------------------------------------------------
constexpr int kWbMax = 64;
constexpr int kWbRefine = 12;
constexpr int kWbLdsMax = 48;  // (m (2m+1) + 2 m (m+1) + 3 m^2) doubles = 129 KB at m = 48

struct dd_acc {
  double hi, lo;
};
// s += a * b, exactly up to the final rounding of hi + lo (Ogita-Rump-Oishi Dot2 step)
__device__ __forceinline__ void dd_fma(dd_acc& s, double a, double b) {
#pragma clang fp contract(off)
  const double p = a * b;
  const double ep = __builtin_fma(a, b, -p);
  const double t = s.hi + p;
  const double z = t - s.hi;
  const double es = (s.hi - (t - z)) + (p - z);
  s.hi = t;
  s.lo += es + ep;
}

// scratch: 4 m^2 doubles of global memory (A as hi / lo pairs, the residual, the last correction), private to this launch
__global__ __launch_bounds__(256) void k_woodbury_inner(PtrTab Msrct, int lds_, PtrTab Gt, int ldg, PtrTab Mdstt, int ldd, int m,
                                                        double alpha, double sign, double* __restrict__ scratch,
                                                        int64_t scratch_stride) {
  const double* __restrict__ Msrc = Msrct.p[blockIdx.x];  // grid x = image (one workgroup each)
  const double* __restrict__ G = Gt.p[blockIdx.x];
  double* __restrict__ Mdst = Mdstt.p[blockIdx.x];
  scratch += (int64_t)blockIdx.x * scratch_stride;
  extern __shared__ __align__(16) double wb[];
  const int w = 2 * m + 1;          // row pitch of the augmented matrix (odd: conflict-free column walks)
  const int q = m + 1;
  double* aug = wb;                 // [m][w]   left: A^T, later X;  right: I, later A^-T
  double* Ms = wb + m * w;          // [m][m+1] Msrc
  double* Gs = Ms + m * q;          // [m][m+1] alpha G
  // A (hi / lo) and the residual live in LDS while they fit beside the elimination (m <= kWbLdsMax), else in the scratch
  double* extra = Gs + m * q;
  double* Ahi = m <= kWbLdsMax ? extra : scratch;  // [m][m]
  double* Alo = m <= kWbLdsMax ? extra + m * m : scratch + m * m;
  double* Rs = m <= kWbLdsMax ? extra + 2 * m * m : scratch + 2 * m * m;
  double* Dl = scratch + 3 * m * m;  // the last applied correction
  __shared__ int piv;
  __shared__ double pval;
  __shared__ double rnorm[4];
  const int tid = threadIdx.x;
  for (int i = tid; i < m * m; i += 256) {
    const int r = i / m, c = i % m;
    Ms[r * q + c] = Msrc[(int64_t)r * lds_ + c];
    Gs[r * q + c] = alpha * G[(int64_t)r * ldg + c];
  }
  __syncthreads();
  // A = I + G Msrc in double-double;  aug[r][c] = A^T[r][c] = A[c][r] (rounded);  aug[r][m + c] = I
  for (int i = tid; i < m * m; i += 256) {
    const int r = i / m, c = i % m;  // element A[r][c]
    dd_acc a{r == c ? 1.0 : 0.0, 0.0};
    for (int k = 0; k < m; ++k) dd_fma(a, Gs[r * q + k], Ms[k * q + c]);
    const double hi = a.hi + a.lo;
    Ahi[r * m + c] = hi;
    Alo[r * m + c] = (a.hi - hi) + a.lo;
    aug[c * w + r] = hi;
    aug[r * w + m + c] = r == c ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int k = 0; k < m; ++k) {
    // pivot search in column k, rows k..m-1 (one wave, deterministic: the first maximum wins)
    if (tid < 64) {
      double best = -1.0;
      int bi = k;
      for (int r = k + tid; r < m; r += 64) {
        const double v = fabs(aug[r * w + k]);
        if (v > best) best = v, bi = r;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(best, o, 64);
        const int oi = __shfl_down(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) best = ov, bi = oi;
      }
      if (tid == 0) piv = bi;
    }
    __syncthreads();
    const int p = piv;
    if (p != k)
      for (int c = tid; c < 2 * m; c += 256) {
        const double t = aug[k * w + c];
        aug[k * w + c] = aug[p * w + c];
        aug[p * w + c] = t;
      }
    __syncthreads();
    if (tid == 0) pval = aug[k * w + k];
    __syncthreads();
    const double inv = 1.0 / pval;
    for (int c = tid; c < 2 * m; c += 256) aug[k * w + c] *= inv;
    __syncthreads();
    // eliminate column k from every other row: thread -> (row, column slice)
    for (int i = tid; i < m * 2 * m; i += 256) {
      const int r = i / (2 * m), c = i % (2 * m);
      if (r == k || c == k) continue;  // column k itself is cleared after the sweep (its old value is the multiplier)
      aug[r * w + c] = fma(-aug[r * w + k], aug[k * w + c], aug[r * w + c]);
    }
    __syncthreads();
    for (int r = tid; r < m; r += 256)
      if (r != k) aug[r * w + k] = 0.0;
    __syncthreads();
  }
  // right half = A^-T, i.e. Ainv[k][c] = aug[c][m + k].  X0 = sign Msrc Ainv into the left half
  for (int i = tid; i < m * m; i += 256) {
    const int r = i / m, c = i % m;
    double x = 0.0;
    for (int k = 0; k < m; ++k) x = fma(Ms[r * q + k], aug[c * w + m + k], x);
    aug[r * w + c] = sign * x;
  }
  __syncthreads();
  // refinement until the residual stops shrinking (contraction ~ cond(A) x 1e-16 per step: 2 steps at cond 1e12, more
  // towards 1e15), at most kWbRefine steps; a step after which the residual grew is taken back
  double prev = 1.0e300;
  for (int it = 0; it < kWbRefine; ++it) {
    // R = sign Msrc - X A  (double-double accumulation, A as hi + lo)
    double rmax = 0.0;
    for (int i = tid; i < m * m; i += 256) {
      const int r = i / m, c = i % m;
      dd_acc a{sign * Ms[r * q + c], 0.0};
      double lo = 0.0;
      for (int k = 0; k < m; ++k) {
        const double x = -aug[r * w + k];
        dd_fma(a, x, Ahi[k * m + c]);
        lo = fma(x, Alo[k * m + c], lo);
      }
      const double rv = a.hi + (a.lo + lo);
      Rs[r * m + c] = rv;
      rmax = fmax(rmax, fabs(rv));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rmax = fmax(rmax, __shfl_down(rmax, o, 64));
    if ((tid & 63) == 0) rnorm[tid >> 6] = rmax;
    __syncthreads();
    const double cur = fmax(fmax(rnorm[0], rnorm[1]), fmax(rnorm[2], rnorm[3]));
    __syncthreads();
    if (!(cur < prev)) {  // uniform: every thread read the same four values.  The last step did not help: take it back
      for (int i = tid; i < m * m; i += 256) aug[(i / m) * w + i % m] -= Dl[i];
      __syncthreads();
      break;
    }
    // X += R Ainv
    for (int i = tid; i < m * m; i += 256) {
      const int r = i / m, c = i % m;
      double dlt = 0.0;
      for (int k = 0; k < m; ++k) dlt = fma(Rs[r * m + k], aug[c * w + m + k], dlt);
      Dl[i] = dlt;
      aug[r * w + c] += dlt;
    }
    __syncthreads();
    if (cur == 0.0) break;
    prev = cur;
  }
  // Mdst = 0.5 (X + X^T)
  for (int i = tid; i < m * m; i += 256) {
    const int r = i / m, c = i % m;
    Mdst[(int64_t)r * ldd + c] = 0.5 * (aug[r * w + c] + aug[c * w + r]);
  }
}

// ------------------------------------------------------------------------------------------------
// 1-D pass of a separable PSF (column kernel: AXIS 0, row kernel: AXIS 1), stride 1, circular.
// k_conv_tile spends two LDS reads per FMA; here a thread owns 8 consecutive outputs ALONG the filter axis, so one LDS
// read of an input feeds up to 8 FMAs and the weights of a 16-tap block sit in registers: 0.3 LDS reads per FMA.
// Lanes always run ACROSS the filter axis (conflict-free LDS reads of consecutive doubles); for the row kernel the tile
// is therefore staged transposed ([x][y], row pad 1) and the result goes back through LDS for coalesced stores.
// The tap list (ascending offsets, possibly with gaps) is expanded to a dense, zero-padded coefficient array in LDS:
// e[k + h] multiplies in[p + k]; forward uses k = -offset, adjoint k = +offset (as k_conv_tile).
// ------------------------------------------------------------------------------------------------
constexpr int k1dR = 8;        // outputs per thread along the filter axis
constexpr int k1dTB = 16;      // taps per register block
constexpr int k1dAcross = 64;  // lanes across the filter axis
constexpr int k1dAlong = 32;   // outputs along the filter axis per workgroup (4 waves x 8)

template <int AXIS>
__global__ __launch_bounds__(256) void k_conv1d(const double* __restrict__ in, double* __restrict__ out,
                                                const int* __restrict__ toff, const double* __restrict__ tw,
                                                int ntaps, int S, int h, int adjoint, const double* __restrict__ add,
                                                double add_scale, const fh_cg_state* __restrict__ states) {
  IMG_GUARD(states, blockIdx.z / 3);
  extern __shared__ __align__(16) double lds1d[];
  const int span = k1dAlong + 2 * h;                   // staged extent along the filter axis
  const int ld = k1dAcross + 1;                        // row pitch (doubles)
  double* tile = lds1d;                                // [span][ld]
  const int nblk = (2 * h + 1 + k1dTB - 1) / k1dTB;
  double* e = tile + span * ld;                        // [nblk * 16 + 8] dense coefficients, zero padded
  const int tid = threadIdx.x;
  const int plane = blockIdx.z;
  const double* src = in + (int64_t)plane * S * S;
  // block origin: `a0` along the filter axis, `c0` across it
  const int a0 = (AXIS == 0 ? blockIdx.y : blockIdx.x) * k1dAlong;
  const int c0 = (AXIS == 0 ? blockIdx.x : blockIdx.y) * k1dAcross;
  for (int t = tid; t < nblk * k1dTB + k1dR; t += 256) e[t] = 0.0;
  __syncthreads();
  for (int t = tid; t < ntaps; t += 256) e[(adjoint ? toff[t] : -toff[t]) + h] = tw[t];
  // stage: tile[a][c] = in at (along = a0 - h + a, across = c0 + c), circular.  All loads of a batch of 8 elements per
  // thread are issued before the first LDS store (the staging is latency-bound otherwise); the wrap is a compare + add.
  const int total = span * k1dAcross;
  for (int base = 0; base < total; base += 256 * 8) {
    double v[8];
    int dst[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = base + q * 256 + tid;
      int a, c;
      if (AXIS == 0) a = idx / k1dAcross, c = idx % k1dAcross;  // consecutive lanes: consecutive x = across (coalesced)
      else c = idx / span, a = idx % span;                       // consecutive lanes: consecutive x = along (coalesced)
      int ga = a0 - h + a;
      ga += ga < 0 ? S : 0;
      ga -= ga >= S ? S : 0;
      const int gc = c0 + c;
      const bool ok = idx < total && gc < S;
      const int64_t off = AXIS == 0 ? (int64_t)ga * S + gc : (int64_t)gc * S + ga;
      v[q] = ok ? src[off] : 0.0;
      dst[q] = idx < total ? a * ld + c : -1;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (dst[q] >= 0) tile[dst[q]] = v[q];
  }
  __syncthreads();
  const int lc = tid & 63, g = tid >> 6;  // lane across, wave = group of 8 outputs along
  double acc[k1dR];
#pragma unroll
  for (int r = 0; r < k1dR; ++r) acc[r] = 0.0;
  const double* col = tile + (g * k1dR) * ld + lc;
  for (int tb = 0; tb < nblk; ++tb) {
    double wreg[k1dTB];
#pragma unroll
    for (int t = 0; t < k1dTB; ++t) wreg[t] = e[tb * k1dTB + t];
    // output r uses input index (r + k) with k = tap index in [0, 2h]; within this block taps 16 tb .. 16 tb + 15
#pragma unroll
    for (int jj = 0; jj < k1dTB + k1dR - 1; ++jj) {
      const int a = tb * k1dTB + jj;
      const double v = a < span - g * k1dR ? col[a * ld] : 0.0;
#pragma unroll
      for (int r = 0; r < k1dR; ++r) {
        const int t = jj - r;
        if (t >= 0 && t < k1dTB) acc[r] = fma(wreg[t], v, acc[r]);
      }
    }
  }
  __syncthreads();  // tile is dead: reuse it to turn the results so that stores are coalesced along x
  if (AXIS == 0) {
    const int gx = c0 + lc;
#pragma unroll
    for (int r = 0; r < k1dR; ++r) {
      const int gy = a0 + g * k1dR + r;
      if (gx < S && gy < S) {
        const int64_t o = (int64_t)plane * S * S + (int64_t)gy * S + gx;
        double v = acc[r];
        if (add != nullptr) v = fma(add_scale, add[o], v);
        out[o] = v;
      }
    }
  } else {
    double* turn = tile;  // [64 rows y][33]
#pragma unroll
    for (int r = 0; r < k1dR; ++r) turn[lc * (k1dAlong + 1) + g * k1dR + r] = acc[r];
    __syncthreads();
    for (int idx = tid; idx < k1dAcross * k1dAlong; idx += 256) {
      const int yy = idx / k1dAlong, xx = idx % k1dAlong;
      const int gy = c0 + yy, gx = a0 + xx;
      if (gx < S && gy < S) {
        const int64_t o = (int64_t)plane * S * S + (int64_t)gy * S + gx;
        double v = turn[yy * (k1dAlong + 1) + xx];
        if (add != nullptr) v = fma(add_scale, add[o], v);
        out[o] = v;
      }
    }
  }
}

static int conv_launch(fh_context* ctx, const double* in, double* out, const int32_t* dy, const int32_t* dx,
                       const double* w, int ntaps, int halo, int planes, int stride, int adjoint, const double* add,
                       double add_scale, const fh_cg_state* done, hipStream_t st) {
  const int S = ctx->S;
  // halo encodings: h = max(|dy|,|dx|) <= 32;  -(h+1) / -(h+101): 1-D column / row tap list;  1000 + 64 hy + hx: both extents
  int hy2 = -1, hx2 = -1;
  if (halo >= 1000) {
    hy2 = (halo - 1000) / 64, hx2 = (halo - 1000) % 64;
    if (hy2 > 32 || hx2 > 32) return FH_EINVAL;
    halo = hy2 > hx2 ? hy2 : hx2;
  }
  if (stride < 1 || S % stride != 0 || halo < -133 || halo > 32 || ntaps > kMaxTaps) return FH_EINVAL;
  if (stride > 1 && stride <= 4 && halo >= 0) {  // SR pair on the LDS-tiled kernels when the sizes allow
    const int hyq = hy2 >= 0 ? hy2 : halo, hxq = hy2 >= 0 ? hx2 : halo, s_ = stride;
    static bool attr_sr = false;
    if (!attr_sr) {
      FH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_dec), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   100 * 1024));
      FH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_up), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   100 * 1024));
      attr_sr = true;
    }
    if (!adjoint && (S / s_) % 16 == 0) {
      const int hq = (hxq + s_ - 1) / s_, R = 15 * s_ + 1 + 2 * hyq, Wp = 16 + 2 * hq + 1;
      const size_t lds = (((size_t)s_ * R * Wp + 1) & ~(size_t)1) * sizeof(double) + (size_t)ntaps * 16;
      if (lds <= 100 * 1024) {
        const int So = S / s_;
        hipLaunchKernelGGL(k_conv_dec, dim3(So / 16, So / 16, planes), dim3(256), lds, st, in, out, dy, dx, w, ntaps, S, s_,
                           hyq, hxq, add, add_scale, done);
        FH_LAUNCH_CHECK();
        return 0;
      }
    }
    if (adjoint && S % (16 * s_) == 0) {
      const int hqy = (hyq + s_ - 1) / s_, hqx = (hxq + s_ - 1) / s_;
      const int cap = ((2 * hyq + s_) / s_ + 1) * ((2 * hxq + s_) / s_ + 1);  // >= taps of one phase
      const int Hu = 16 + 2 * hqy, Wu = 16 + 2 * hqx + 1, T = 16 * s_;
      const size_t lds = ((size_t)2 * s_ * s_ * cap + (((size_t)Hu * Wu + 1) & ~(size_t)1) + (size_t)T * (T + 1)) * sizeof(double);
      if (lds <= 100 * 1024) {
        hipLaunchKernelGGL(k_conv_up, dim3(S / T, S / T, planes), dim3(256), lds, st, in, out, dy, dx, w, ntaps, S, s_, hyq,
                           hxq, cap, add, add_scale, done);
        FH_LAUNCH_CHECK();
        return 0;
      }
    }
  }
  if (!adjoint && stride > 1) {
    const int So = S / stride;
    const int64_t total = (int64_t)planes * So * So;
    hipLaunchKernelGGL(k_conv_direct, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, out, dy, dx, w,
                       ntaps, S, stride, planes, add, add_scale, done);
  } else {
    // halo = max(|dy|,|dx|); negative values encode 1-D tap lists: -(h+1) = column kernel (dx = 0), -(h+101) = row kernel
    int hy = halo, hx = halo;
    if (hy2 >= 0) hy = hy2, hx = hx2;
    if (halo <= -101) hy = 0, hx = -halo - 101;
    else if (halo < 0) hy = -halo - 1, hx = 0;
    if (halo < 0 && stride == 1 && S % 2 == 0) {  // 1-D pass at full resolution: register-blocked kernel
      const int h = hy + hx;
      const int nblk = (2 * h + 1 + k1dTB - 1) / k1dTB;
      const size_t lds1 = ((size_t)(k1dAlong + 2 * h) * (k1dAcross + 1) + nblk * k1dTB + k1dR) * sizeof(double);
      if (lds1 <= 64 * 1024) {
        if (hx == 0) {
          dim3 grid((S + k1dAcross - 1) / k1dAcross, (S + k1dAlong - 1) / k1dAlong, planes);
          hipLaunchKernelGGL(k_conv1d<0>, grid, dim3(256), lds1, st, in, out, dy, w, ntaps, S, h, adjoint, add, add_scale,
                             done);
        } else {
          dim3 grid((S + k1dAlong - 1) / k1dAlong, (S + k1dAcross - 1) / k1dAcross, planes);
          hipLaunchKernelGGL(k_conv1d<1>, grid, dim3(256), lds1, st, in, out, dx, w, ntaps, S, h, adjoint, add, add_scale,
                             done);
        }
        FH_LAUNCH_CHECK();
        return 0;
      }
    }
    if (stride == 1) {  // 8 outputs per thread when the 64 x 32 tile with its halo fits
      const size_t lds8 = (((size_t)(kT8W + 2 * hx) * (kT8H + 2 * hy) + 1) & ~(size_t)1) * sizeof(double) + (size_t)ntaps * 16;
      if (lds8 <= 64 * 1024) {
        dim3 grid8((S + kT8W - 1) / kT8W, (S + kT8H - 1) / kT8H, planes);
        hipLaunchKernelGGL(k_conv_tile8, grid8, dim3(256), lds8, st, in, out, dy, dx, w, ntaps, S, hy, hx, adjoint, add,
                           add_scale, done);
        FH_LAUNCH_CHECK();
        return 0;
      }
    }
    const size_t lds = (size_t)(32 + 2 * hx) * (16 + 2 * hy) * sizeof(double) + (size_t)ntaps * 12;
    if (lds > 64 * 1024) return FH_ESIZE;
    dim3 grid((S + 31) / 32, (S + 15) / 16, planes);
    hipLaunchKernelGGL(k_conv_tile, grid, dim3(256), lds, st, in, out, dy, dx, w, ntaps, S, hy, hx, adjoint,
                       adjoint ? stride : 1, add, add_scale, done);
  }
  FH_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// A_mm(u) = sigma_y^2 u + A C A^T u       (conditioning_mechanisms.py:395-400, 505-511, 653-659)
// ------------------------------------------------------------------------------------------------
// dot_part / dot_nparts (CG only): where the operator's last pass can also reduce u . (A u) per image it writes *dot_nparts
// block partials per image to dot_part (stride kCgScratch) - the caller then skips its own dot kernel; else *dot_nparts = 0
static int amm_launch(fh_context* ctx, const fh_problem* p, const fh_batch& per, const double* u, double* out,
                      const fh_cg_state* states, hipStream_t st, double* dot_part = nullptr, int* dot_nparts = nullptr) {
  if (dot_nparts != nullptr) *dot_nparts = 0;
  static const bool no_dot_fuse = getenv("FH_CG_NO_DOT_FUSE") != nullptr;  // A/B switch
  if (no_dot_fuse) dot_part = nullptr;
  const int S = ctx->S;
  const int64_t d = p->d;  // per image
  const int nimg = per.nimg;
  const int planes = p->planes * nimg;
  if (d != (int64_t)p->planes * S * S || p->planes != 3 || nimg < 1 || nimg > ctx->nimg_max) return FH_EINVAL;
  int rc;
  double *w0 = ctx->w0, *w1 = ctx->w1;
  const int halo = p->halo;
  const dim3 egrid(512, 1, (unsigned)nimg);
  const bool sep = p->ntaps2 > 0;  // separable PSF: two 1-D passes (blur only, stride 1)
  if (sep && p->stride != 1) return FH_EINVAL;
  if (p->op == 1 && p->use_dct && p->fold_fwd_w != nullptr) {
    // separable blur folded into the DCT bases: out = sigma_y^2 u + A idct2( C dct2(A^T u) ) in 4 dense passes + the
    // apply, instead of 4 blur passes + 4 DCT passes + the apply + an axpy epilogue
    if (!p->fold_fwd_h || !p->fold_inv_w || !p->fold_inv_h) return FH_EINVAL;
    if (p->fold_sym) {  // symmetric PSF: the fold pointers are packed half bases (k_dct_sym)
      if (p->m == 0) {
        // no factor columns yet (every call above sigma = 10, i.e. three quarters of all CG iterations of a Heun-30 run): C z =
        // D .* z rides in the epilogue of the forward pass - two kernels per apply instead of three
        rc = dct2d_launch_sym(ctx, u, w0, planes, p->fold_fwd_w, p->fold_fwd_h, 0, nullptr, 0.0, states, st, &per);
        if (rc) return rc;
        return dct2d_launch_sym(ctx, w0, out, planes, p->fold_inv_w, p->fold_inv_h, 1, u, p->sigma_y2, states, st, nullptr,
                                dot_part, kCgScratch, dot_nparts);
      }
      rc = dct2d_launch_sym(ctx, u, w1, planes, p->fold_fwd_w, p->fold_fwd_h, 0, nullptr, 0.0, states, st);
      if (rc) return rc;
      rc = rep_apply_launch(ctx, per, p->ldm, w1, w0, d, p->m, states, st);
      if (rc) return rc;
      return dct2d_launch_sym(ctx, w0, out, planes, p->fold_inv_w, p->fold_inv_h, 1, u, p->sigma_y2, states, st, nullptr,
                              dot_part, kCgScratch, dot_nparts);
    }
    rc = dct2d_launch_bases(ctx, u, w1, planes, p->fold_fwd_w, p->fold_fwd_h, nullptr, 0.0, states, st);
    if (rc) return rc;
    rc = rep_apply_launch(ctx, per, p->ldm, w1, w0, d, p->m, states, st);
    if (rc) return rc;
    return dct2d_launch_bases(ctx, w0, out, planes, p->fold_inv_w, p->fold_inv_h, u, p->sigma_y2, states, st);
  }
  // w0 = A^T u
  if (p->op == 0) {
    hipLaunchKernelGGL(k_mask, egrid, dim3(256), 0, st, per, u, (const double*)nullptr, 0.0, w0, d, states);
  } else if (sep) {
    rc = conv_launch(ctx, u, w1, p->tap2_dy, p->tap2_dx, p->tap2_w, p->ntaps2, p->halo2, planes, 1, 1, nullptr, 0.0,
                     states, st);
    if (rc) return rc;
    rc = conv_launch(ctx, w1, w0, p->tap_dy, p->tap_dx, p->tap_w, p->ntaps, halo, planes, 1, 1, nullptr, 0.0, states, st);
    if (rc) return rc;
  } else {
    rc = conv_launch(ctx, u, w0, p->tap_dy, p->tap_dx, p->tap_w, p->ntaps, halo, planes, p->stride, 1, nullptr, 0.0,
                     states, st);
    if (rc) return rc;
  }
  // w0 <- C w0  (through the DCT basis when the covariance lives there)
  static const bool no_sym_amm = getenv("FH_DCT_NOSYM") != nullptr;
  if (p->use_dct && p->m == 0 && ctx->sym_fwd != nullptr && !no_sym_amm) {
    // as above: the diagonal-only covariance apply inside the forward DCT's second pass
    rc = dct2d_launch_sym(ctx, w0, w1, planes, ctx->sym_fwd, ctx->sym_fwd, 0, nullptr, 0.0, states, st, &per);
    if (rc) return rc;
    rc = dct2d_launch_sym(ctx, w1, w0, planes, ctx->sym_inv, ctx->sym_inv, 1, nullptr, 0.0, states, st);
    if (rc) return rc;
    { double* t_ = w0; w0 = w1; w1 = t_; }  // the result is expected in w1 below
  } else if (p->use_dct) {
    rc = dct2d_launch(ctx, w0, w1, planes, 0, states, st);
    if (rc) return rc;
    rc = rep_apply_launch(ctx, per, p->ldm, w1, w0, d, p->m, states, st);
    if (rc) return rc;
    rc = dct2d_launch(ctx, w0, w1, planes, 1, states, st);
    if (rc) return rc;
  } else {
    rc = rep_apply_launch(ctx, per, p->ldm, w0, w1, d, p->m, states, st);
    if (rc) return rc;
  }
  // out = sigma_y^2 u + A w1
  if (p->op == 0) {
    hipLaunchKernelGGL(k_mask, egrid, dim3(256), 0, st, per, (const double*)w1, u, p->sigma_y2, out, d, states);
  } else if (sep) {
    rc = conv_launch(ctx, w1, w0, p->tap_dy, p->tap_dx, p->tap_w, p->ntaps, halo, planes, 1, 0, nullptr, 0.0, states, st);
    if (rc) return rc;
    rc = conv_launch(ctx, w0, out, p->tap2_dy, p->tap2_dx, p->tap2_w, p->ntaps2, p->halo2, planes, 1, 0, u, p->sigma_y2,
                     states, st);
    if (rc) return rc;
  } else {
    rc = conv_launch(ctx, w1, out, p->tap_dy, p->tap_dx, p->tap_w, p->ntaps, halo, planes, p->stride, 0, u, p->sigma_y2,
                     states, st);
    if (rc) return rc;
  }
  FH_LAUNCH_CHECK();
  return 0;
}

static fh_batch batch_of(const fh_problem* p) {
  fh_batch per;
  memset(&per, 0, sizeof(per));
  per.nimg = 1;
  per.D[0] = p->D, per.r[0] = p->r, per.B[0] = p->B, per.M[0] = p->M, per.mask[0] = p->mask;
  return per;
}

// ------------------------------------------------------------------------------------------------
// Conjugate gradients, conditioning_utils/cg.py:232-282 (M = I, x0 = b).  Scalars stay on the device;
// the two reductions per iteration are block partials re-summed by every workgroup of the next kernel.
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ double sum_partials(const double* __restrict__ part, int n, double* red) {
  double s = threadIdx.x < n ? part[threadIdx.x] : 0.0;
  return block_sum_256(s, red);
}

// r = b - Ab ; p = r ; x = b ; partial r.r and b.b
__global__ __launch_bounds__(256) void k_cg_init(const double* __restrict__ b, const double* __restrict__ Ab,
                                                 double* __restrict__ x, double* __restrict__ r,
                                                 double* __restrict__ p, double* __restrict__ part, int64_t n) {
  {
    const int64_t off = (int64_t)blockIdx.z * n;
    b += off, x += off, r += off, p += off;
    if (Ab != nullptr) Ab += off;
    part += (int64_t)blockIdx.z * kCgScratch;
  }
  __shared__ double red[4];
  double srr = 0.0, sbb = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double bi = b[i], ri = Ab != nullptr ? bi - Ab[i] : bi;  // Ab == null: x0 = 0 (scipy's start), r = b
    x[i] = Ab != nullptr ? bi : 0.0;
    r[i] = ri;
    p[i] = ri;
    srr = fma(ri, ri, srr);
    sbb = fma(bi, bi, sbb);
  }
  srr = block_sum_256(srr, red);
  sbb = block_sum_256(sbb, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = srr;
    part[kDotBlocks + blockIdx.x] = sbb;
  }
}

__global__ __launch_bounds__(256) void k_cg_init_fin(const double* __restrict__ part, int nparts, RtolArr rtols,
                                                     double atol, int maxiter, fh_cg_state* __restrict__ stt,
                                                     double* __restrict__ rzbuf, int scipy_mode) {
  const double rtol = rtols.v[blockIdx.z];
  part += (int64_t)blockIdx.z * kCgScratch;
  rzbuf += (int64_t)blockIdx.z * kCgScratch;
  stt += blockIdx.z;
  __shared__ double red[4];
  const double srr = sum_partials(part, nparts, red);
  const double sbb = sum_partials(part + kDotBlocks, nparts, red);
  if (threadIdx.x == 0) {
    rzbuf[0] = srr;
    stt->rz = srr;
    stt->rnorm = sqrt(srr);
    stt->bnorm = sqrt(sbb);
    const double s = rtol * sqrt(sbb);
    stt->stop = s > atol ? s : atol;
    stt->done = 0;
    stt->niter = 0;
    stt->optimal = 0;
    stt->it = 0;
    stt->k_cur = 0;
    stt->maxiter = maxiter;
    stt->scipy_mode = scipy_mode;
    if (scipy_mode && sqrt(srr) <= stt->stop) {  // scipy tests the initial residual before the first iteration
      stt->optimal = 1;
      stt->done = 1;
    }
  }
}

// iteration k, first half: pAp check, x += alpha p, r -= alpha Ap, partial r.r
__global__ __launch_bounds__(256) void k_cg_step1(const double* __restrict__ p, const double* __restrict__ ap,
                                                  double* __restrict__ x, double* __restrict__ r,
                                                  const double* __restrict__ part_pap, int nparts,
                                                  double* __restrict__ part_rr, const double* __restrict__ rzbuf,
                                                  fh_cg_state* __restrict__ stt, int64_t n) {
  {
    const int64_t off = (int64_t)blockIdx.z * n, so = (int64_t)blockIdx.z * kCgScratch;
    p += off, ap += off, x += off, r += off;
    part_pap += so, part_rr += so, rzbuf += so;
    stt += blockIdx.z;
  }
  if (stt->done) return;
  const int it = stt->it, k = it + 1;
  if (it >= stt->maxiter) {  // cg.py:245 loop bound; the host reports niter = maxiter
    if (blockIdx.x == 0 && threadIdx.x == 0) stt->done = 3;
    return;
  }
  __shared__ double red[4];
  const double pAp = sum_partials(part_pap, nparts, red);
  if (stt->scipy_mode ? pAp == 0.0 : pAp <= 1e-16) {  // cg.py:250 (scipy has no such test: only an exact zero stops it here)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      stt->pAp = pAp;
      stt->niter = k;
      stt->done = 2;
    }
    return;
  }
  const double alpha = rzbuf[(k - 1) & 1] / pAp;
  double srr = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    x[i] = fma(alpha, p[i], x[i]);
    const double ri = fma(-alpha, ap[i], r[i]);
    r[i] = ri;
    srr = fma(ri, ri, srr);
  }
  srr = block_sum_256(srr, red);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = srr;
  if (blockIdx.x == 0 && threadIdx.x == 0) stt->k_cur = k;
}

// iteration k, second half: stopping test, p = r + beta p
__global__ __launch_bounds__(256) void k_cg_step2(const double* __restrict__ r, double* __restrict__ p,
                                                  const double* __restrict__ part_rr, int nparts,
                                                  double* __restrict__ rzbuf, fh_cg_state* __restrict__ stt,
                                                  int64_t n) {
  {
    const int64_t off = (int64_t)blockIdx.z * n, so = (int64_t)blockIdx.z * kCgScratch;
    r += off, p += off;
    part_rr += so, rzbuf += so;
    stt += blockIdx.z;
  }
  if (stt->done) return;
  const int k = stt->k_cur;
  __shared__ double red[4];
  const double rz_new = sum_partials(part_rr, nparts, red);
  const double rnorm = sqrt(rz_new);
  if (rnorm <= stt->stop) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      stt->rnorm = rnorm;
      stt->niter = k;
      stt->optimal = 1;
      stt->done = 1;
    }
    return;
  }
  const double beta = rz_new / rzbuf[(k - 1) & 1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    p[i] = fma(beta, p[i], r[i]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    rzbuf[k & 1] = rz_new;
    stt->rnorm = rnorm;
    stt->niter = k;
    stt->it = k;
  }
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int fh_version(void) { return 100; }

// serialises hipGraph captures (cg_graph_for) against the legacy-stream work of context creation / destruction, see there
static std::mutex g_capture_mu;

int fh_context_create(fh_context** out, int S, int planes_max, int m_cap) {
  if (out == nullptr || S < 2 || S > 256 || (S & 1) || planes_max < 1 || m_cap < 0 || m_cap > FH_MAX_COLS)
    return FH_EINVAL;
  std::lock_guard<std::mutex> capture_lock(g_capture_mu);
  fh_context* c = new fh_context();
  memset(c, 0, sizeof(*c));
  c->S = S;
  c->planes_max = planes_max;
  c->nimg_max = planes_max / 3 > 0 ? planes_max / 3 : 1;
  if (c->nimg_max > FH_MAX_BATCH) c->nimg_max = FH_MAX_BATCH;
  c->m_cap = m_cap;
  const size_t nimg = (size_t)planes_max * S * S;
  std::vector<double> bas((size_t)S * S), bast((size_t)S * S);
  for (int k = 0; k < S; ++k) {
    const long double sk = k == 0 ? sqrtl(1.0L / S) : sqrtl(2.0L / S);
    for (int n = 0; n < S; ++n) {
      const long double v = sk * cosl(M_PIl * (2 * n + 1) * k / (2.0L * S));
      bas[(size_t)k * S + n] = (double)v;
      bast[(size_t)n * S + k] = (double)v;
    }
  }
  const int ntiles = (m_cap + kGT - 1) / kGT;
  const int npairs = ntiles * (ntiles + 1) / 2;
  // Gram partials: one image at the context's column capacity, or (batched covariance updates, <= 64 columns: one tile
  // pair) every image of a batch side by side
  const int pairs_batched = m_cap > 0 ? c->nimg_max : 0;
  const int pairs = npairs > pairs_batched ? npairs : pairs_batched;
  c->gpartial_elems = (int64_t)(pairs > 0 ? pairs : 1) * kGramRowBlocks * kGT * kGT;
  FH_CHECK(hipMalloc(&c->basis, sizeof(double) * S * S));
  FH_CHECK(hipMalloc(&c->basis_t, sizeof(double) * S * S));
  FH_CHECK(hipMemcpy(c->basis, bas.data(), sizeof(double) * S * S, hipMemcpyHostToDevice));
  FH_CHECK(hipMemcpy(c->basis_t, bast.data(), sizeof(double) * S * S, hipMemcpyHostToDevice));
  if (S % 128 == 0) {  // packed half bases of the symmetric passes: forward [Pe; Po], inverse [Qe; Qo] (k_dct_sym)
    const int H = S / 2;
    std::vector<double> sf((size_t)2 * H * H), si((size_t)2 * H * H);
    for (int j = 0; j < H; ++j)
      for (int n = 0; n < H; ++n) {
        sf[(size_t)j * H + n] = bas[(size_t)(2 * j) * S + n];                      // Pe[j][n] = C[2j][n]
        sf[(size_t)H * H + (size_t)j * H + n] = bas[(size_t)(2 * j + 1) * S + n];  // Po[j][n] = C[2j+1][n]
        si[(size_t)j * H + n] = bas[(size_t)(2 * n) * S + j];                      // Qe[k][j'] = C[2j'][k]  (here k = j, j' = n)
        si[(size_t)H * H + (size_t)j * H + n] = bas[(size_t)(2 * n + 1) * S + j];  // Qo[k][j'] = C[2j'+1][k]
      }
    FH_CHECK(hipMalloc(&c->sym_fwd, sizeof(double) * 2 * H * H));
    FH_CHECK(hipMalloc(&c->sym_inv, sizeof(double) * 2 * H * H));
    FH_CHECK(hipMemcpy(c->sym_fwd, sf.data(), sizeof(double) * 2 * H * H, hipMemcpyHostToDevice));
    FH_CHECK(hipMemcpy(c->sym_inv, si.data(), sizeof(double) * 2 * H * H, hipMemcpyHostToDevice));
  }
  FH_CHECK(hipMalloc(&c->tmp_img, sizeof(double) * nimg));
  FH_CHECK(hipMalloc(&c->partial, sizeof(double) * kPartialRows * FH_MAX_COLS * c->nimg_max));
  FH_CHECK(hipMalloc(&c->gpartial, sizeof(double) * c->gpartial_elems));
  FH_CHECK(hipMalloc(&c->sync, sizeof(unsigned int) * (FH_MAX_BATCH * kSyncStride + 32)));
  FH_CHECK(hipMemset(c->sync, 0, sizeof(unsigned int) * (FH_MAX_BATCH * kSyncStride + 32)));
  {
    int dev = 0, cus = 0;
    FH_CHECK(hipGetDevice(&dev));
    FH_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    c->num_cus = cus;
  }
  FH_CHECK(hipMalloc(&c->coef, sizeof(double) * 2 * FH_MAX_COLS * c->nimg_max));
  FH_CHECK(hipMemset(c->coef, 0, sizeof(double) * 2 * FH_MAX_COLS));
  FH_CHECK(hipMalloc(&c->cg_r, sizeof(double) * nimg));
  FH_CHECK(hipMalloc(&c->cg_p, sizeof(double) * nimg));
  FH_CHECK(hipMalloc(&c->cg_ap, sizeof(double) * nimg));
  FH_CHECK(hipMalloc(&c->cg_x, sizeof(double) * nimg));
  FH_CHECK(hipMalloc(&c->w0, sizeof(double) * nimg));
  FH_CHECK(hipMalloc(&c->w1, sizeof(double) * nimg));
  FH_CHECK(hipMalloc(&c->w2, sizeof(double) * kCgScratch * c->nimg_max));
  FH_CHECK(hipMalloc(&c->cg_state, sizeof(fh_cg_state) * c->nimg_max));
  FH_CHECK(hipMemset(c->cg_state, 0, sizeof(fh_cg_state) * c->nimg_max));
  FH_CHECK(hipHostMalloc((void**)&c->h_state, sizeof(fh_cg_state) * c->nimg_max, hipHostMallocDefault));
  FH_CHECK(hipHostMalloc((void**)&c->h_scal, sizeof(double) * 64, hipHostMallocDefault));
  *out = c;
  return 0;
}

int fh_context_destroy(fh_context* c) {
  if (c == nullptr) return 0;
  std::lock_guard<std::mutex> capture_lock(g_capture_mu);
  void* bufs[] = {c->basis, c->basis_t, c->tmp_img, c->partial, c->gpartial, c->coef, c->cg_r,
                  c->cg_p,  c->cg_ap,   c->w0,      c->w1,      c->w2,       c->cg_state, c->sync,
                  c->sym_fwd, c->sym_inv};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (auto& g : c->graphs)
    if (g.exec) {
      (void)hipGraphExecDestroy(g.exec);
      (void)hipGraphDestroy(g.graph);
    }
  if (c->cg_x) (void)hipFree(c->cg_x);
  if (c->h_state) (void)hipHostFree(c->h_state);
  if (c->h_scal) (void)hipHostFree(c->h_scal);
  delete c;
  return 0;
}

// a captured CG chunk bakes in the apply kernel that `exclusive` / `fused_disabled` selected at capture time: drop the cached
// graphs whenever that selection can change, so a graph never replays the single-sweep kernel on a context that has since
// lost its exclusivity (or had a time-out reported)
static void drop_graphs(fh_context* ctx) {
  for (auto& g : ctx->graphs) {
    if (g.exec != nullptr) {
      (void)hipGraphExecDestroy(g.exec);
      (void)hipGraphDestroy(g.graph);
      g.exec = nullptr;
      g.graph = nullptr;
    }
  }
}

int fh_context_set_exclusive(fh_context* ctx, int exclusive) {
  if (!ctx) return FH_EINVAL;
  const int e = exclusive < 0 ? 0 : (exclusive > 2 ? 2 : exclusive);
  if ((e >= 2) != (ctx->exclusive >= 2)) drop_graphs(ctx);  // (levels 0 and 1 select the same kernels)
  ctx->exclusive = e;
  return 0;
}

int fh_debug_read_stamps(fh_context* ctx, unsigned long long* out_host, int count, void* stream) {
  if (!ctx || !out_host || count < 1 || (int64_t)count > ctx->gpartial_elems) return FH_EINVAL;
  FH_CHECK(hipMemcpyAsync(out_host, ctx->gpartial, sizeof(unsigned long long) * count, hipMemcpyDeviceToHost, (hipStream_t)stream));
  FH_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int fh_context_status(fh_context* ctx, void* stream) {
  if (!ctx) return FH_EINVAL;
  unsigned int flag = 0;
  FH_CHECK(hipMemcpyAsync(&flag, ctx->sync + FH_MAX_BATCH * kSyncStride, sizeof(flag), hipMemcpyDeviceToHost,
                          (hipStream_t)stream));
  FH_CHECK(hipStreamSynchronize((hipStream_t)stream));
  if (flag == 0) return 0;
  // a single-sweep apply timed out waiting for its peers (another grid-synchronising kernel shared the GPU): its output
  // is invalid.  Re-arm the counters, keep this context on the two-pass kernels and report.
  ctx->fused_disabled = 1;
  drop_graphs(ctx);
  FH_CHECK(hipMemsetAsync(ctx->sync, 0, sizeof(unsigned int) * (FH_MAX_BATCH * kSyncStride + 32), (hipStream_t)stream));
  FH_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return FH_ESYNC;
}

int fh_dct2d(fh_context* ctx, const double* in, double* out, int planes, int inverse, void* stream) {
  if (!ctx || !in || !out || planes < 1) return FH_EINVAL;
  return dct2d_launch(ctx, in, out, planes, inverse, nullptr, (hipStream_t)stream);
}

int fh_rep_apply(fh_context* ctx, const double* D, const double* r, const double* B, const double* M, int ldm,
                 const double* z, double* out, int64_t d, int m, void* stream) {
  if (!ctx || !D || !z || !out || d < 2) return FH_EINVAL;
  if (m > 0 && (!r || !B || !M || ldm < m)) return FH_EINVAL;
  fh_batch per;
  memset(&per, 0, sizeof(per));
  per.nimg = 1;
  per.D[0] = D, per.r[0] = r, per.B[0] = B, per.M[0] = M;
  return rep_apply_launch(ctx, per, ldm, z, out, d, m, nullptr, (hipStream_t)stream);
}

int fh_rep_apply_batched(fh_context* ctx, const fh_batch* per, int ldm, const double* z, double* out, int64_t d,
                         int m, void* stream) {
  if (!ctx || !per || !z || !out || d < 2 || per->nimg < 1 || per->nimg > FH_MAX_BATCH) return FH_EINVAL;
  for (int i = 0; i < per->nimg; ++i) {
    if (!per->D[i]) return FH_EINVAL;
    if (m > 0 && (!per->r[i] || !per->B[i] || !per->M[i] || ldm < m)) return FH_EINVAL;
  }
  return rep_apply_launch(ctx, *per, ldm, z, out, d, m, nullptr, (hipStream_t)stream);
}

// ---- batched internals: every per-image operand arrives as a PtrTab, vectors of a batch are contiguous [image][d] ----
static inline int gram_pairs(int m) {
  const int ntiles = (m + kGT - 1) / kGT;
  return ntiles * (ntiles + 1) / 2;
}

static int rep_invert_b(fh_context* ctx, int nimg, PtrTab Dx, PtrTab rx, PtrTab B, double shift, PtrTab Dy, PtrTab ry,
                        PtrTab G, int ldg, int64_t d, int m, hipStream_t st) {
  const unsigned Z = (unsigned)nimg;
  // Gram uses the shifted diagonal: apply the shift first, then read Dx
  hipLaunchKernelGGL(k_invert_diag, dim3((unsigned)((d + 255) / 256), 1, Z), dim3(256), 0, st, Dx, rx, shift, Dy, ry, d);
  if (m > 0) {
    const int ntiles = (m + kGT - 1) / kGT;
    const int npairs = gram_pairs(m);
    const int64_t gstride = (int64_t)npairs * kGramRowBlocks * kGT * kGT;
    if (gstride * nimg > ctx->gpartial_elems) return FH_ESIZE;
    hipLaunchKernelGGL(k_gram, dim3(kGramRowBlocks, npairs, Z), dim3(256), 0, st, B, rx, Dx, ctx->gpartial, gstride, d, m, ntiles, 0,
                       0.0);
    hipLaunchKernelGGL(k_gram_reduce, dim3(npairs, kGT * kGT / 64, Z), dim3(256), 0, st, (const double*)ctx->gpartial, gstride,
                       kGramRowBlocks, ntiles, m, G, ldg);
  }
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_rep_invert(fh_context* ctx, double* Dx, const double* rx, const double* B, double shift, double* Dy,
                  double* ry, double* G, int ldg, int64_t d, int m, void* stream) {
  if (!ctx || !Dx || !Dy || d < 1 || m < 0 || m > ctx->m_cap) return FH_EINVAL;
  if (m > 0 && (!rx || !ry || !B || !G || ldg < m)) return FH_EINVAL;
  return rep_invert_b(ctx, 1, tab1(Dx), tab1(rx), tab1(B), shift, tab1(Dy), tab1(ry), tab1(G), ldg, d, m, (hipStream_t)stream);
}

// de <- s2 dm; scal[img][0] = dx.de
static int space_prep_b(fh_context* ctx, int nimg, const double* dm, double s2, const double* dx, double* de, PtrTab scal,
                        int64_t d, hipStream_t st) {
  hipLaunchKernelGGL(k_space_prep, dim3(kDotBlocks, 1, (unsigned)nimg), dim3(256), 0, st, dm, s2, dx, de, ctx->w2,
                     (int64_t)kCgScratch, d);
  hipLaunchKernelGGL(k_scalar_reduce, dim3((unsigned)nimg), dim3(256), 0, st, (const double*)ctx->w2, (int64_t)kCgScratch,
                     kDotBlocks, scal, 0);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_space_prep(fh_context* ctx, const double* dm, double s2, const double* dx, double* de, double* scal,
                  int64_t d, void* stream) {
  if (!ctx || !dm || !dx || !de || !scal) return FH_EINVAL;
  return space_prep_b(ctx, 1, dm, s2, dx, de, tab1(scal), d, (hipStream_t)stream);
}

// scal[img][slot] = a[img] . b[img]
static int dot_b(fh_context* ctx, int nimg, const double* a, const double* b, PtrTab scal, int slot, int64_t d, hipStream_t st) {
  hipLaunchKernelGGL(k_dot_partial, dim3(kDotBlocks, 1, (unsigned)nimg), dim3(256), 0, st, a, b, (const double*)nullptr,
                     (const double*)nullptr, ctx->w2, d, kCgScratch, (const fh_cg_state*)nullptr);
  hipLaunchKernelGGL(k_scalar_reduce, dim3((unsigned)nimg), dim3(256), 0, st, (const double*)ctx->w2, (int64_t)kCgScratch,
                     kDotBlocks, scal, slot);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_dot(fh_context* ctx, const double* a, const double* b, double* scal, int slot, int64_t d, void* stream) {
  if (!ctx || !a || !b || !scal || slot < 0) return FH_EINVAL;
  return dot_b(ctx, 1, a, b, tab1(scal), slot, d, (hipStream_t)stream);
}

int fh_space_commit(fh_context* ctx, const double* de, const double* cdx, double gamma, double q, double s2,
                    double* Dc, const double* rc, double* Bc_col0, double* Bc_col1, double* Dh, const double* rh,
                    double* Bh_col0, double* Bh_col1, int project, int64_t d, void* stream) {
  if (!ctx || !de || !cdx || !Dc || !Dh || !Bh_col0 || !Bh_col1) return FH_EINVAL;
  if (!project && (!Bc_col0 || !Bc_col1)) return FH_EINVAL;
  const PtrTab none = tab1(nullptr);
  hipLaunchKernelGGL(k_space_commit, dim3(512), dim3(256), 0, (hipStream_t)stream, de, cdx, gamma, q, s2, tab1(Dc), tab1(rc),
                     tab1(Bc_col0), tab1(Bc_col1), tab1(Dh), tab1(rh), tab1(Bh_col0), tab1(Bh_col1), project, d, 0, none, none, 0,
                     0, none, 0, 0);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_space_commit_dev(fh_context* ctx, const double* de, const double* cdx, const double* scal, double s2, double* Dc,
                        const double* rc, double* Bc_col0, double* Bc_col1, double* Dh, const double* rh,
                        double* Bh_col0, double* Bh_col1, double* Mc, int ldc, int mc, double* Mh, int ldh, int mh,
                        int project, int64_t d, void* stream) {
  if (!ctx || !de || !cdx || !scal || !Dc || !Dh || !Bh_col0 || !Bh_col1 || !Mh || mh < 0 || ldh < mh + 2) return FH_EINVAL;
  if (!project && (!Bc_col0 || !Bc_col1 || !Mc || mc < 0 || ldc < mc + 2)) return FH_EINVAL;
  hipLaunchKernelGGL(k_space_commit, dim3(512), dim3(256), 0, (hipStream_t)stream, de, cdx, 0.0, 0.0, s2, tab1(Dc), tab1(rc),
                     tab1(Bc_col0), tab1(Bc_col1), tab1(Dh), tab1(rh), tab1(Bh_col0), tab1(Bh_col1), project, d, 1, tab1(scal),
                     tab1(Mc), ldc, mc, tab1(Mh), ldh, mh);
  FH_LAUNCH_CHECK();
  return 0;
}

static int inner_update_b(fh_context* ctx, int nimg, PtrTab Msrc, int ld_src, PtrTab G, int ldg, PtrTab Mdst, int ld_dst, int m,
                          double alpha, double sign, hipStream_t st) {
  if (m == 0) return 0;
  if (m > kWbMax) return FH_ESIZE;  // the caller falls back to its host path
  const size_t lds = ((size_t)m * (2 * m + 1) + 2 * (size_t)m * (m + 1) + (m <= kWbLdsMax ? 3 * (size_t)m * m : 0)) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    FH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_woodbury_inner),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    attr_set = true;
  }
  const int64_t sstride = (int64_t)4 * kWbMax * kWbMax;
  if (ctx->gpartial_elems < sstride * nimg) return FH_ESIZE;
  // ctx->gpartial: the Gram partials were reduced into G before this launch (stream order); free until the next k_gram
  hipLaunchKernelGGL(k_woodbury_inner, dim3((unsigned)nimg), dim3(256), lds, st, Msrc, ld_src, G, ldg, Mdst, ld_dst, m, alpha, sign,
                     ctx->gpartial, sstride);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_woodbury_inner(fh_context* ctx, const double* Msrc, int ld_src, const double* G, int ldg, double* Mdst, int ld_dst,
                      int m, void* stream) {
  if (!ctx || m < 0 || (m > 0 && (!Msrc || !G || !Mdst || ld_src < m || ldg < m || ld_dst < m))) return FH_EINVAL;
  return inner_update_b(ctx, 1, tab1(Msrc), ld_src, tab1(G), ldg, tab1(Mdst), ld_dst, m, 1.0, -1.0, (hipStream_t)stream);
}

int fh_axpby(double alpha, const double* a, double beta, const double* b, double* out, int64_t n, void* stream) {
  if (!a || !out || n < 1) return FH_EINVAL;
  hipLaunchKernelGGL(k_axpby, dim3(512), dim3(256), 0, (hipStream_t)stream, alpha, a, beta, b, out, n);
  FH_LAUNCH_CHECK();
  return 0;
}

// ---- whole covariance updates (see fh_hip.h): the step-by-step entry points, enqueued from C in covariance.py's order, for
// ONE image (fh_cov_time_update / fh_cov_space_update) or for all images of a lock-step batch in one launch sequence
// (fh_cov_*_update_batched: same kernels with grid z = image, so an image's result does not depend on the batch it is in).
struct CovB {
  int nimg;
  const fh_cov_state* st[FH_MAX_BATCH];
};
enum CovField { F_D, F_R, F_M, F_BC, F_BH, F_G, F_SCAL };
static PtrTab cov_tab(const CovB& cb, CovField f, int rep = 0, int64_t offset = 0) {
  PtrTab t;
  memset(&t, 0, sizeof(t));
  for (int i = 0; i < cb.nimg; ++i) {
    const fh_cov_state* s = cb.st[i];
    double* q = f == F_D ? s->D[rep] : f == F_R ? s->r[rep] : f == F_M ? s->M[rep] : f == F_BC ? s->Bc : f == F_BH ? s->Bh
                : f == F_G ? s->G : s->scal;
    t.p[i] = q ? q + offset : nullptr;
  }
  return t;
}

static int cov_invert_b(fh_context* ctx, const CovB& cb, int src, int dst, CovField base, int m, double shift, hipStream_t sq) {
  const fh_cov_state* s0 = cb.st[0];
  if (m > 0 && s0->ldg < m) return FH_ESIZE;
  int rc = rep_invert_b(ctx, cb.nimg, cov_tab(cb, F_D, src), cov_tab(cb, F_R, src), cov_tab(cb, base), shift, cov_tab(cb, F_D, dst),
                        cov_tab(cb, F_R, dst), cov_tab(cb, F_G), s0->ldg, s0->d, m, sq);
  if (rc) return rc;
  return inner_update_b(ctx, cb.nimg, cov_tab(cb, F_M, src), s0->ldm, cov_tab(cb, F_G), s0->ldg, cov_tab(cb, F_M, dst), s0->ldm, m,
                        1.0, -1.0, sq);
}

// Forward time shift  X <- (X^-1 + s I)^-1  computed from X's OWN representation (online_update_bfgs.py:166-167, 172-173):
//   D' = D / (1 + s D),  r' = r / (1 + s D),  M' = M (I + s G M)^-1  with  G = B^T diag(r^2 / (1 + s D)) B,
// while the inverse representation only moves its diagonal (Dinv += s; its row scale r / D and inner matrix are unchanged).
// Equal to inverting the shifted inverse representation (what the reference does), but I + s G M has the eigenvalues of
// I + s X restricted to the factor span - in [1, (sigma / sigma')^2] for the covariance, [1, 2) for the Hessian - whereas
// the inverse route weights the Gram by 1 / D (1e-4 .. 1e4 with the DCT prior) and loses up to 1e9 of the 1e16.
static int cov_shift_forward_b(fh_context* ctx, const CovB& cb, int fwd, int inv, CovField base, int m, double s, hipStream_t sq) {
  const fh_cov_state* s0 = cb.st[0];
  const int64_t d = s0->d;
  const unsigned Z = (unsigned)cb.nimg;
  if (m > 0) {
    const int ntiles = (m + kGT - 1) / kGT;
    const int npairs = gram_pairs(m);
    const int64_t gstride = (int64_t)npairs * kGramRowBlocks * kGT * kGT;
    if (gstride * cb.nimg > ctx->gpartial_elems || s0->ldg < m) return FH_ESIZE;
    hipLaunchKernelGGL(k_gram, dim3(kGramRowBlocks, npairs, Z), dim3(256), 0, sq, cov_tab(cb, base), cov_tab(cb, F_R, fwd),
                       cov_tab(cb, F_D, fwd), ctx->gpartial, gstride, d, m, ntiles, 1, s);
    hipLaunchKernelGGL(k_gram_reduce, dim3(npairs, kGT * kGT / 64, Z), dim3(256), 0, sq, (const double*)ctx->gpartial, gstride,
                       kGramRowBlocks, ntiles, m, cov_tab(cb, F_G), s0->ldg);
  }
  hipLaunchKernelGGL(k_forward_diag, dim3((unsigned)((d + 255) / 256), 1, Z), dim3(256), 0, sq, cov_tab(cb, F_D, fwd),
                     cov_tab(cb, F_R, fwd), cov_tab(cb, F_D, inv), s, d);
  FH_LAUNCH_CHECK();
  if (m == 0) return 0;
  return inner_update_b(ctx, cb.nimg, cov_tab(cb, F_M, fwd), s0->ldm, cov_tab(cb, F_G), s0->ldg, cov_tab(cb, F_M, fwd), s0->ldm, m, s,
                        1.0, sq);
}

static int cov_fwd_b(fh_context* ctx, const CovB& cb, const double* in, double* out, int inverse, hipStream_t sq) {
  if (cb.st[0]->use_dct) return dct2d_launch(ctx, in, out, 3 * cb.nimg, inverse, nullptr, sq);
  if (in != out) return fh_axpby(1.0, in, 0.0, nullptr, out, cb.st[0]->d * cb.nimg, sq);
  return 0;
}

static fh_batch rep_batch(const CovB& cb, int rep, CovField base) {
  fh_batch per;
  memset(&per, 0, sizeof(per));
  per.nimg = cb.nimg;
  for (int i = 0; i < cb.nimg; ++i) {
    const fh_cov_state* s = cb.st[i];
    per.D[i] = s->D[rep], per.r[i] = s->r[rep], per.B[i] = base == F_BC ? s->Bc : s->Bh, per.M[i] = s->M[rep];
  }
  return per;
}

// the images of a batch must agree on everything that shapes the launch
static int cov_batch_check(fh_context* ctx, const CovB& cb) {
  if (cb.nimg < 1 || cb.nimg > FH_MAX_BATCH || cb.nimg > ctx->nimg_max || 3 * cb.nimg > ctx->planes_max) return FH_ESIZE;
  const fh_cov_state* a = cb.st[0];
  for (int i = 0; i < cb.nimg; ++i) {
    const fh_cov_state* s = cb.st[i];
    if (!s || s->d != a->d || s->m_c != a->m_c || s->m_h != a->m_h || s->ldm != a->ldm || s->ldg != a->ldg ||
        s->project != a->project || s->use_dct != a->use_dct)
      return FH_EINVAL;
  }
  return 0;
}

// x, score, wx, ws, t0, mean_out, score_out: [nimg][d] contiguous
static int time_update_b(fh_context* ctx, const CovB& cb, const double* x, const double* score, double shift_c, double shift_h,
                         double sigma_next2, int only_covariance, double* wx, double* ws, double* t0, double* mean_out,
                         double* score_out, hipStream_t sq) {
  const fh_cov_state* s0 = cb.st[0];
  if (s0->m_c > kWbMax || s0->m_h > kWbMax) return FH_ESIZE;
  const int64_t d = s0->d, dn = d * cb.nimg;
  int rc = cov_shift_forward_b(ctx, cb, 0, 1, F_BC, s0->m_c, shift_c, sq);  // C <- (C^-1 + shift)^-1, C^-1 += shift
  if (rc || only_covariance) return rc;
  if (!x || !score || !wx || !ws || !t0 || !mean_out || !score_out) return FH_EINVAL;
  if ((rc = cov_fwd_b(ctx, cb, x, wx, 0, sq))) return rc;
  if ((rc = cov_fwd_b(ctx, cb, score, ws, 0, sq))) return rc;
  // t0 = H^-1 score with the OLD Hessian, then H <- (H^-1 + shift)^-1, new score = H t0
  if ((rc = rep_apply_launch(ctx, rep_batch(cb, 3, F_BH), s0->ldm, ws, t0, d, s0->m_h, nullptr, sq))) return rc;
  if ((rc = cov_shift_forward_b(ctx, cb, 2, 3, F_BH, s0->m_h, shift_h, sq))) return rc;
  if ((rc = rep_apply_launch(ctx, rep_batch(cb, 2, F_BH), s0->ldm, t0, ws, d, s0->m_h, nullptr, sq))) return rc;
  if ((rc = fh_axpby(1.0, wx, sigma_next2, ws, wx, dn, sq))) return rc;  // mean' = x + s'^2 score'
  if ((rc = cov_fwd_b(ctx, cb, wx, mean_out, 1, sq))) return rc;
  return cov_fwd_b(ctx, cb, ws, score_out, 1, sq);
}

// mean_x, mean_xn, x, xn, dx, de, cdx: [nimg][d] contiguous (dx, de, cdx: scratch)
static int space_update_b(fh_context* ctx, const CovB& cb, const double* mean_x, const double* mean_xn, double s2, const double* x,
                          const double* xn, double* dx, double* de, double* cdx, hipStream_t sq) {
  const fh_cov_state* s0 = cb.st[0];
  if (s0->m_c + 2 > kWbMax || s0->m_h + 2 > kWbMax) return FH_ESIZE;
  const int mc = s0->m_c, mh = s0->m_h, nimg = cb.nimg;
  const int64_t d = s0->d, dn = d * nimg;
  if (s0->ldm < (s0->project ? mc : mc + 2) || s0->ldm < mh + 2) return FH_EINVAL;
  int rc;
  if ((rc = fh_axpby(1.0, xn, -1.0, x, dx, dn, sq))) return rc;
  if ((rc = cov_fwd_b(ctx, cb, dx, dx, 0, sq))) return rc;
  if ((rc = fh_axpby(1.0, mean_xn, -1.0, mean_x, de, dn, sq))) return rc;
  if ((rc = cov_fwd_b(ctx, cb, de, de, 0, sq))) return rc;
  const PtrTab scal = cov_tab(cb, F_SCAL);
  if ((rc = space_prep_b(ctx, nimg, de, s2, dx, de, scal, d, sq))) return rc;  // de <- s2 dm; scal[0] = dx.de
  // two-pass apply also on an exclusive context: the single-sweep kernel keeps c = M t in registers, and the closed-form
  // C^-1 below reads it from ctx->coef
  const int fused_was = ctx->fused_disabled;
  ctx->fused_disabled = 1;
  rc = rep_apply_launch(ctx, rep_batch(cb, 0, F_BC), s0->ldm, dx, cdx, d, mc, nullptr, sq);
  ctx->fused_disabled = fused_was;
  if (rc) return rc;
  if ((rc = dot_b(ctx, nimg, cdx, dx, scal, 1, d, sq))) return rc;  // scal[1] = dx.(C dx)
  if (!s0->project && mc > 0)  // c = M t of that apply (ctx->coef) is needed after the next dots pass overwrites it
    hipLaunchKernelGGL(k_copy_small, dim3((unsigned)nimg), dim3(256), 0, sq, (const double*)ctx->coef, (int64_t)2 * FH_MAX_COLS,
                       cov_tab(cb, F_G), mc);
  const PtrTab none = tab1(nullptr);
  hipLaunchKernelGGL(k_space_commit, dim3(512, 1, (unsigned)nimg), dim3(256), 0, sq, (const double*)de, (const double*)cdx, 0.0, 0.0,
                     s2, cov_tab(cb, F_D, 0), cov_tab(cb, F_R, 0), s0->project ? none : cov_tab(cb, F_BC, 0, (int64_t)mc * d),
                     s0->project ? none : cov_tab(cb, F_BC, 0, (int64_t)(mc + 1) * d), cov_tab(cb, F_D, 2), cov_tab(cb, F_R, 2),
                     cov_tab(cb, F_BH, 0, (int64_t)mh * d), cov_tab(cb, F_BH, 0, (int64_t)(mh + 1) * d), s0->project, d, 1, scal,
                     cov_tab(cb, F_M, 0), s0->ldm, mc, cov_tab(cb, F_M, 2), s0->ldm, mh);
  FH_LAUNCH_CHECK();
  const int mh2 = mh + 2;
  if (s0->project) {
    // the pair went into the diagonal: C^-1 over the unchanged columns by Woodbury
    if ((rc = cov_invert_b(ctx, cb, 0, 1, F_BC, mc, 0.0, sq))) return rc;
  } else {
    // C^-1 in closed form (k_bfgs_inverse_commit): cC = M t saved above; s = Wi^T de and Mi s from a dots pass over the
    // old columns with the inverse representation's row scale; de^T D^-1 de from one more reduction
    if (mc > 0 && (rc = rep_apply_launch(ctx, rep_batch(cb, 1, F_BC), s0->ldm, de, nullptr, d, mc, nullptr, sq))) return rc;
    hipLaunchKernelGGL(k_wnorm_partial, dim3(kDotBlocks, 1, (unsigned)nimg), dim3(256), 0, sq, (const double*)de, cov_tab(cb, F_D, 1),
                       ctx->w2, (int64_t)kCgScratch, d);
    hipLaunchKernelGGL(k_scalar_reduce, dim3((unsigned)nimg), dim3(256), 0, sq, (const double*)ctx->w2, (int64_t)kCgScratch,
                       kDotBlocks, scal, 2);
    hipLaunchKernelGGL(k_bfgs_inverse_commit, dim3((unsigned)nimg), dim3(256), 0, sq, scal, cov_tab(cb, F_G), (const double*)ctx->coef,
                       (int64_t)2 * FH_MAX_COLS, cov_tab(cb, F_M, 1), s0->ldm, mc);
    FH_LAUNCH_CHECK();
  }
  return cov_invert_b(ctx, cb, 2, 3, F_BH, mh2, 0.0, sq);  // H^-1 (its diagonal was re-derived from C: full Woodbury)
}

int fh_cov_time_update(fh_context* ctx, const fh_cov_state* st, const double* x, const double* score, double shift_c,
                       double shift_h, double sigma_next2, int only_covariance, double* wx, double* ws,
                       double* mean_out, double* score_out, void* stream) {
  if (!ctx || !st) return FH_EINVAL;
  CovB cb;
  cb.nimg = 1, cb.st[0] = st;
  return time_update_b(ctx, cb, x, score, shift_c, shift_h, sigma_next2, only_covariance, wx, ws, st->t0, mean_out, score_out,
                       (hipStream_t)stream);
}

int fh_cov_space_update(fh_context* ctx, const fh_cov_state* st, const double* mean_x, const double* mean_xn, double s2,
                        const double* x, const double* xn, void* stream) {
  if (!ctx || !st || !mean_x || !mean_xn || !x || !xn) return FH_EINVAL;
  CovB cb;
  cb.nimg = 1, cb.st[0] = st;
  return space_update_b(ctx, cb, mean_x, mean_xn, s2, x, xn, st->t0, st->t1, st->t2, (hipStream_t)stream);
}

int fh_cov_time_update_batched(fh_context* ctx, int nimg, const fh_cov_state* sts, const double* x, const double* score,
                               double shift_c, double shift_h, double sigma_next2, double* work, double* mean_out,
                               double* score_out, void* stream) {
  if (!ctx || !sts || !x || !score || !work || !mean_out || !score_out || nimg < 1 || nimg > FH_MAX_BATCH) return FH_EINVAL;
  CovB cb;
  cb.nimg = nimg;
  for (int i = 0; i < nimg; ++i) cb.st[i] = sts + i;
  int rc = cov_batch_check(ctx, cb);
  if (rc) return rc;
  const int64_t dn = sts[0].d * nimg;
  return time_update_b(ctx, cb, x, score, shift_c, shift_h, sigma_next2, 0, work, work + dn, work + 2 * dn, mean_out, score_out,
                       (hipStream_t)stream);
}

int fh_cov_space_update_batched(fh_context* ctx, int nimg, const fh_cov_state* sts, const double* mean_x, const double* mean_xn,
                                double s2, const double* x, const double* xn, double* work, void* stream) {
  if (!ctx || !sts || !mean_x || !mean_xn || !x || !xn || !work || nimg < 1 || nimg > FH_MAX_BATCH) return FH_EINVAL;
  CovB cb;
  cb.nimg = nimg;
  for (int i = 0; i < nimg; ++i) cb.st[i] = sts + i;
  int rc = cov_batch_check(ctx, cb);
  if (rc) return rc;
  const int64_t dn = sts[0].d * nimg;
  return space_update_b(ctx, cb, mean_x, mean_xn, s2, x, xn, work, work + dn, work + 2 * dn, (hipStream_t)stream);
}

int fh_read_scalars(fh_context* ctx, const double* scal, double* out_host, int k, void* stream) {
  if (!ctx || !scal || !out_host || k < 1 || k > 64) return FH_EINVAL;
  FH_CHECK(hipMemcpyAsync(ctx->h_scal, scal, sizeof(double) * k, hipMemcpyDeviceToHost, (hipStream_t)stream));
  FH_CHECK(hipStreamSynchronize((hipStream_t)stream));
  memcpy(out_host, ctx->h_scal, sizeof(double) * k);
  return 0;
}

int fh_conv_circ(fh_context* ctx, const double* in, double* out, const int32_t* dy, const int32_t* dx,
                 const double* w, int ntaps, int halo, int planes, int stride, int adjoint, void* stream) {
  if (!ctx || !in || !out || !dy || !dx || !w || ntaps < 1 || planes < 1) return FH_EINVAL;
  return conv_launch(ctx, in, out, dy, dx, w, ntaps, halo, planes, stride, adjoint, nullptr, 0.0, nullptr,
                     (hipStream_t)stream);
}

int fh_amm(fh_context* ctx, const fh_problem* p, const double* u, double* out, void* stream) {
  if (!ctx || !p || !u || !out) return FH_EINVAL;
  return amm_launch(ctx, p, batch_of(p), u, out, nullptr, (hipStream_t)stream);
}

// One chunk of CG iterations for all images of the batch, enqueued on `st` (eagerly, or while the stream is being
// captured into a graph).
static int cg_enqueue_chunk(fh_context* ctx, const fh_problem* p, const fh_batch& per, int64_t n, int count,
                            hipStream_t st) {
  double *x = ctx->cg_x, *r = ctx->cg_r, *pk = ctx->cg_p, *ap = ctx->cg_ap;
  double* part = ctx->w2;
  double* rzbuf = ctx->w2 + 4 * kDotBlocks;
  fh_cg_state* stt = ctx->cg_state;
  const dim3 grid(kCgBlocks, 1, (unsigned)per.nimg);
  for (int i = 0; i < count; ++i) {
    int dot_n = 0;  // > 0: the operator's last pass already left the p.Ap block partials
    int rc = amm_launch(ctx, p, per, pk, ap, stt, st, part, &dot_n);
    if (rc) return rc;
    if (dot_n == 0) {
      hipLaunchKernelGGL(k_dot_partial, grid, dim3(256), 0, st, (const double*)pk, (const double*)ap,
                         (const double*)nullptr, (const double*)nullptr, part, n, kCgScratch, (const fh_cg_state*)stt);
      dot_n = kCgBlocks;
    }
    hipLaunchKernelGGL(k_cg_step1, grid, dim3(256), 0, st, (const double*)pk, (const double*)ap, x, r,
                       (const double*)part, dot_n, part + 2 * kDotBlocks, (const double*)rzbuf, stt, n);
    hipLaunchKernelGGL(k_cg_step2, grid, dim3(256), 0, st, (const double*)r, pk,
                       (const double*)(part + 2 * kDotBlocks), kCgBlocks, rzbuf, stt, n);
  }
  return 0;
}

// The iteration body depends on device state only (counter, scalars, done flag), so a chunk is captured once per
// (problem, pointers, m) into a hipGraph and replayed: one host call per 8 iterations instead of ~110 launches.
static hipGraphExec_t cg_graph_for(fh_context* ctx, const fh_problem* p, const fh_batch& per, int64_t n, int chunk,
                                   hipStream_t st) {
  if (ctx->graphs_disabled) return nullptr;
  if (getenv("FH_NO_GRAPH") != nullptr) return nullptr;  // A/B switch for profiling
  ++ctx->graph_clock;
  fh_graph_entry* slot = &ctx->graphs[0];
  for (auto& g : ctx->graphs) {
    if (g.exec != nullptr && g.n == n && memcmp(&g.key, p, sizeof(fh_problem)) == 0 &&
        memcmp(&g.bkey, &per, sizeof(fh_batch)) == 0) {
      g.stamp = ctx->graph_clock;
      return g.exec;
    }
    if (g.stamp < slot->stamp) slot = &g;
  }
  if (slot->exec != nullptr) {
    (void)hipGraphExecDestroy(slot->exec);
    (void)hipGraphDestroy(slot->graph);
    slot->exec = nullptr;
  }
  // A capture is short (80 launches) but must not overlap, in another host thread of the process, with the legacy-stream
  // copies and allocations of fh_context_create / fh_context_destroy (a second lock-step group creating its contexts):
  // the capture is invalidated (hipErrorStreamCaptureInvalidated in this thread) or the other call fails
  // (hipErrorStreamCaptureImplicit).  g_capture_mu serialises the two.
  std::lock_guard<std::mutex> capture_lock(g_capture_mu);
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    ctx->graphs_disabled = 1;  // e.g. the legacy null stream: stay on eager launches
    return nullptr;
  }
  const int rc = cg_enqueue_chunk(ctx, p, per, n, chunk, st);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(st, &graph);
  if (rc != 0 || e != hipSuccess || graph == nullptr) {
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    ctx->graphs_disabled = 1;
    return nullptr;
  }
  hipGraphExec_t exec = nullptr;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipGraphDestroy(graph);
    ctx->graphs_disabled = 1;
    return nullptr;
  }
  memcpy(&slot->key, p, sizeof(fh_problem));
  memcpy(&slot->bkey, &per, sizeof(fh_batch));
  slot->n = n;
  slot->exec = exec;
  slot->graph = graph;
  slot->stamp = ctx->graph_clock;
  return exec;
}

int fh_cg_solve_batched(fh_context* ctx, const fh_problem* p, const fh_batch* perp, const double* b, double* x,
                        const double* rtol_host, double atol, int maxiter, fh_cg_info* info, void* stream) {
  if (!ctx || !p || !perp || !b || !x || !rtol_host || !info || maxiter < 1) return FH_EINVAL;
  const fh_batch& per = *perp;
  const int nimg = per.nimg;
  if (nimg < 1 || nimg > ctx->nimg_max || nimg > FH_MAX_BATCH) return FH_ESIZE;
  hipStream_t st = (hipStream_t)stream;
  const int S = ctx->S;
  const int So = S / (p->op == 2 ? p->stride : 1);
  const int64_t n = (int64_t)p->planes * So * So;  // measurement dimension per image
  double *r = ctx->cg_r, *pk = ctx->cg_p, *ap = ctx->cg_ap;
  double* part = ctx->w2;                      // per image: [0,256) pAp / init r.r ; [256,512) init b.b ; [512,768) r.r
  double* rzbuf = ctx->w2 + 4 * kDotBlocks;    // per image: [2]
  fh_cg_state* stt = ctx->cg_state;
  RtolArr rt;
  for (int i = 0; i < FH_MAX_BATCH; ++i) rt.v[i] = i < nimg ? rtol_host[i] : 1.0;
  for (int i = 0; i < nimg; ++i)
    if (!(rtol_host[i] > 0 || atol > 0)) return FH_EINVAL;
  int rc = 0;
  if (!p->cg_scipy) {
    rc = amm_launch(ctx, p, per, b, ap, nullptr, st);  // A x0 with x0 = b
    if (rc) return rc;
  }
  const dim3 grid(kCgBlocks, 1, (unsigned)nimg);
  hipLaunchKernelGGL(k_cg_init, grid, dim3(256), 0, st, b, p->cg_scipy ? (const double*)nullptr : (const double*)ap, ctx->cg_x,
                     r, pk, part, n);
  hipLaunchKernelGGL(k_cg_init_fin, dim3(1, 1, (unsigned)nimg), dim3(256), 0, st, (const double*)part, kCgBlocks, rt,
                     atol, maxiter, stt, rzbuf, (int)p->cg_scipy);
  const int chunk = 8;
  hipGraphExec_t exec = cg_graph_for(ctx, p, per, n, chunk, st);
  fh_cg_state* h = ctx->h_state;  // pinned: the periodic read-back is a true async copy
  memset(h, 0, sizeof(fh_cg_state) * nimg);
  bool all_done = false;
  for (int launched = 0; launched < maxiter + chunk && !all_done; launched += chunk) {
    if (exec != nullptr) {
      FH_CHECK(hipGraphLaunch(exec, st));
    } else {
      rc = cg_enqueue_chunk(ctx, p, per, n, chunk, st);
      if (rc) return rc;
    }
    FH_CHECK(hipMemcpyAsync(h, stt, sizeof(fh_cg_state) * nimg, hipMemcpyDeviceToHost, st));
    FH_CHECK(hipStreamSynchronize(st));
    all_done = true;
    for (int i = 0; i < nimg; ++i) all_done = all_done && h[i].done != 0;
  }
  FH_CHECK(hipMemcpyAsync(x, ctx->cg_x, sizeof(double) * n * nimg, hipMemcpyDeviceToDevice, st));
  FH_LAUNCH_CHECK();
  for (int i = 0; i < nimg; ++i) {
    info[i].niter = (h[i].done == 1 || h[i].done == 2) ? h[i].niter : maxiter;
    info[i].optimal = h[i].optimal;
    info[i].residual_norm = h[i].rnorm;
    info[i].b_norm = h[i].bnorm;
  }
  return 0;
}

int fh_cg_solve(fh_context* ctx, const fh_problem* p, const double* b, double* x, double rtol, double atol,
                int maxiter, fh_cg_info* info, void* stream) {
  if (!p) return FH_EINVAL;
  const fh_batch per = batch_of(p);
  return fh_cg_solve_batched(ctx, p, &per, b, x, &rtol, atol, maxiter, info, stream);
}

}  // extern "C"
