// Shared device helpers for the Free Hunch gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fh_hip.h"

#define FH_CHECK(expr)                       \
  do {                                       \
    hipError_t e_ = (expr);                  \
    if (e_ != hipSuccess) return (int)e_;    \
  } while (0)
#define FH_LAUNCH_CHECK() FH_CHECK(hipGetLastError())

namespace fh {

constexpr int kWave = 64;
constexpr int kPartialRows = 1024;  // max workgroups that write reduction partials

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
  return v;
}

// Sum over a 256-thread block; result valid in every thread.  `red` holds >= 4 doubles.
__device__ __forceinline__ double block_sum_256(double v, double* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

}  // namespace fh

struct fh_cg_state {  // device-resident control block of the CG loop
  double rz, pAp, rnorm, stop, bnorm;
  int done;      // 0 running, 1 converged (optimal), 2 pAp <= 1e-16, 3 maxiter reached
  int niter, optimal;
  int it;        // completed iterations (device-side loop counter: the iteration body is graph-replayable)
  int k_cur;     // iteration in flight, written by step 1, consumed by step 2
  int maxiter;
  int scipy_mode;  // fh_problem.cg_scipy of the running solve
};

struct fh_graph_entry {  // one instantiated chunk-of-iterations graph, keyed by the problem it was captured for
  fh_problem key;
  fh_batch bkey;
  int64_t n;
  hipGraphExec_t exec;
  hipGraph_t graph;
  uint64_t stamp;
};

struct fh_context {
  int S, planes_max, m_cap;
  int nimg_max;       // images a batched call may carry = max(1, planes_max / 3)
  double* basis;      // [S][S]  C[k][n] = s_k cos(pi (2n+1) k / 2S)
  double* basis_t;    // [S][S]  transpose
  double* sym_fwd;    // [2][S/2][S/2] packed half bases of the symmetric forward passes (null unless S % 64 == 0)
  double* sym_inv;    // [2][S/2][S/2] the same for the inverse passes
  double* tmp_img;    // [planes_max*S*S] DCT intermediate
  double* partial;    // [kPartialRows][FH_MAX_COLS] block partials (dots) / scalar partials
  double* gpartial;   // Gram partials
  int64_t gpartial_elems;
  double* coef;       // [2][FH_MAX_COLS] double-buffered t = B^T (r.*z)
  int t_parity;       // which half the next apply accumulates into
  // CG work vectors (n <= planes_max*S*S)
  double *cg_r, *cg_p, *cg_ap, *w0, *w1, *w2;
  fh_cg_state* cg_state;
  fh_cg_state* h_state;  // pinned host mirror of cg_state (read back every few iterations)
  double* h_scal;        // pinned host staging for fh_read_scalars (64 doubles)
  double* cg_x;          // CG iterate (the caller's x is written once at the end)
  fh_graph_entry graphs[4];
  uint64_t graph_clock;
  int graphs_disabled;   // set when stream capture is not possible on the caller's stream (e.g. the null stream)
  unsigned int* sync;    // k_rep_fused: per image {arrive, depart} counters on their own lines + one error word
  int num_cus;
  int exclusive;         // the caller guarantees that no other grid-synchronising kernel shares the GPU with this context
  int fused_disabled;    // set after a k_rep_fused time-out was reported
};
