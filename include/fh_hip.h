/*
 * fh_hip.h - C ABI of libfh_hip.so: the MI355X (gfx950) kernels of the Free Hunch guided-sampling
 * hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless its name ends
 * in `_host`.  Every entry point returns 0 on success, a hipError_t value (>0) on a HIP failure or a
 * negative FH_E* code on bad arguments; nothing throws, nothing allocates in the hot path (work
 * buffers come from the context created once by fh_context_create).  All work is enqueued on the
 * caller's `stream` (a hipStream_t passed as void*); only fh_cg_solve and fh_read_scalars block.
 *
 * The reference (AaltoML/free-hunch) is pure Python and has no FFI; each entry point replaces the
 * PyTorch op sequence cited next to it (paths relative to the reference checkout).  INTEGRATION.md
 * shows the ctypes binding a reference maintainer would add.
 *
 * Layouts (all float64 on the Free Hunch side, matching the reference's float64/complex128 maths):
 *   image vector  v[planes][S][S]   planes = 3*batch, row-major, n = planes*S*S
 *   factor base   B[m_cap][d]       column-major: column j is the contiguous slice B + j*d
 *   small matrix  M[m][ld]          row-major, ld >= m
 * A covariance "representation"  X = diag(D) + diag(r) B M B^T diag(r)  is the real-arithmetic form of
 * the reference's  diag + U U^T - V V^T  (complex128 U,V, plain transposes); see DESIGN.md.
 */
#ifndef FH_HIP_H
#define FH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FH_EINVAL (-1) /* bad argument (null pointer, size out of range) */
#define FH_ESYNC (-3)  /* a single-sweep cov-apply gave up waiting for its peer workgroups (see fh_context_set_exclusive) */
#define FH_ESIZE (-2)  /* size not supported by this build (e.g. m > FH_MAX_COLS) */
#define FH_MAX_COLS 256

typedef struct fh_context fh_context; /* opaque: DCT bases, reduction scratch, CG state */

/* library / context ---------------------------------------------------------------------------- */
int fh_version(void);
/* S = image side (<= 256), planes_max = largest planes count that will be passed, m_cap = column capacity */
int fh_context_create(fh_context** out, int S, int planes_max, int m_cap);
int fh_context_destroy(fh_context* ctx);
/* exclusive >= 1: the caller guarantees that kernels of this context never run concurrently with another
 * grid-synchronising kernel on the same GPU (one FH stream per process, e.g. the single-image sampler or the lock-step
 * batched CG).  exclusive = 2 additionally selects the single-sweep covariance apply (fh_rep_apply[_batched], and inside
 * fh_cg_solve[_batched] / fh_amm): one launch that keeps the factor base in registers between the reduction and the
 * product (B read once instead of twice).  It is bitwise identical to the two-pass kernels and, since those run at the
 * per-CU ingest limit, no longer faster for any launch shape (profiles/r02_cov_apply_single_sweep.md) - kept for the tests
 * and for profiling.  Default 0, and 1: two-pass kernels, safe under any concurrency. */
int fh_context_set_exclusive(fh_context* ctx, int exclusive);
/* 0, or FH_ESYNC if a single-sweep apply of this context timed out waiting for its peer workgroups since the last call
 * (its output was invalid; the context then stays on the two-pass kernels).  Synchronises the stream. */
int fh_context_status(fh_context* ctx, void* stream);
/* profiling aid: with FH_FUSED_DEBUG set in the environment the single-sweep apply records the 100 MHz wall clock at its
 * phase boundaries (8 words per workgroup, in the Gram scratch); this copies `count` words to the host. */
int fh_debug_read_stamps(fh_context* ctx, unsigned long long* out_host, int count, void* stream);

/* 2-D orthonormal DCT-II (inverse = 0) / DCT-III (inverse = 1) over the last two axes of
 * in[planes][S][S].  Replaces torch_dct.dct_2d / idct_2d(norm='ortho'),
 * conditioning_utils/online_update_bfgs.py:351-374.  in may equal out. */
int fh_dct2d(fh_context* ctx, const double* in, double* out, int planes, int inverse, void* stream);

/* out = D.*z + r.*(B (M (B^T (r.*z))))   - one covariance/Hessian apply.
 * Replaces CovarianceHessianBFGS._denoiser_cov_vector_dot (and the three sibling *_vector_dot),
 * online_update_bfgs.py:194-231.  m may be 0 (then B, r, M may be null). */
int fh_rep_apply(fh_context* ctx, const double* D, const double* r, const double* B, const double* M,
                 int ldm, const double* z, double* out, int64_t d, int m, void* stream);

/* Woodbury step of online_update_bfgs.py:87-119 in real form, one pass over B:
 *   Dx      <- Dx + shift                (the time update's diagonal increment, :166 / :172)
 *   Dy      <- 1 / Dx
 *   ry      <- rx / Dx
 *   G[a][b] <- sum_i B[a][i] B[b][i] rx[i]^2 / Dx[i]      (m x m, ld = ldg, device)
 * The caller finishes with the m x m algebra  My = -Mx (I + G Mx)^-1  (fh_woodbury_inner on the device; float64). */
int fh_rep_invert(fh_context* ctx, double* Dx, const double* rx, const double* B, double shift, double* Dy,
                  double* ry, double* G, int ldg, int64_t d, int m, void* stream);

/* BFGS "space" update vector work, online_update_bfgs.py:262-304.  Inputs are DCT-domain vectors.
 *   fh_space_prep:   de <- s2 * dm ;  scal[0] <- dx.de
 *   fh_dot:          scal[slot] <- a.b
 *   fh_space_commit: append columns (de, cdx) to both factor bases (divided by the rep's row scale),
 *                    or fold them into Dc when project != 0; then Dh <- (Dc/s2 - 1)/s2.
 *                    gamma = 1/(dx.de), q = dx.(C dx) come back from the host. */
int fh_space_prep(fh_context* ctx, const double* dm, double s2, const double* dx, double* de, double* scal,
                  int64_t d, void* stream);
int fh_dot(fh_context* ctx, const double* a, const double* b, double* scal, int slot, int64_t d, void* stream);
int fh_space_commit(fh_context* ctx, const double* de, const double* cdx, double gamma, double q, double s2,
                    double* Dc, const double* rc, double* Bc_col0, double* Bc_col1, double* Dh, const double* rh,
                    double* Bh_col0, double* Bh_col1, int project, int64_t d, void* stream);
/* fh_space_commit without the host in the loop: gamma = 1 / scal[0] and q = scal[1] are read from the DEVICE scalars
 * that fh_space_prep / fh_dot wrote, and the pair is appended to the inner matrices on the device as well
 * (Mc: diag(gamma, -1/q) at rows/cols mc, mc+1 unless project; Mh: the same divided by s2^2 at mh, mh+1). */
int fh_space_commit_dev(fh_context* ctx, const double* de, const double* cdx, const double* scal, double s2, double* Dc,
                        const double* rc, double* Bc_col0, double* Bc_col1, double* Dh, const double* rh,
                        double* Bh_col0, double* Bh_col1, double* Mc, int ldc, int mc, double* Mh, int ldh, int mh,
                        int project, int64_t d, void* stream);

/* ---- whole covariance updates in one call ---------------------------------------------------------------------------
 * update_time_step (online_update_bfgs.py:153-192) and update_space_step (:250-312) as single entry points: the ~35
 * kernel launches of an update are enqueued from C (same kernels, same order as the step-by-step entry points above),
 * so that the host thread of an image spends microseconds, not milliseconds, per guidance call.
 * Representations are indexed C = 0, C^-1 = 1, H = 2, H^-1 = 3; all pointers are device memory owned by the caller;
 * the caller grows the bases / inner matrices BEFORE a space update (m_c + 2, m_h + 2 <= capacity) and bumps its own
 * column counts afterwards (m_c += 2 unless project, m_h += 2). */
typedef struct fh_cov_state {
  int64_t d;
  int32_t m_c, m_h;      /* columns of the covariance / Hessian base */
  int32_t ldm, ldg;      /* leading dimensions of the inner matrices and of the Gram scratch */
  int32_t project;       /* project_to_diagonal */
  int32_t use_dct;       /* 1: vectors are transformed with fh_dct2d (3 planes), 0: identity basis */
  double* D[4];
  double* r[4];
  double* M[4];
  double* Bc;            /* [cap][d] */
  double* Bh;
  double* G;             /* [ldg][ldg] */
  double* scal;          /* >= 8 device scalars */
  double* t0;            /* three d-vectors of scratch */
  double* t1;
  double* t2;
} fh_cov_state;

/* x, score: image-space d-vectors; wx, ws: d-vectors of scratch; mean_out, score_out: results (image space).
 * shift_c = float32(s'^-2 - s^-2), shift_h = -float32(s'^2 - s^2) (the reference's float32-rounded increments).
 * Time update (online_update_bfgs.py:157-192): the reference shifts the diagonal of C^-1 (H^-1) and rebuilds C (H) by
 * Woodbury from the inverse; here the same matrices are formed from C's (H's) OWN representation,
 *   D' = D / (1 + s D), r' = r / (1 + s D), M' = M (I + s G M)^-1, G = B^T diag(r^2 / (1 + s D)) B,
 * and the inverse representation only moves its diagonal - algebraically identical, without the 1 / D weighting of the
 * Gram matrix that costs up to 1e9 of float64's 1e16 at d = 196608 with the DCT prior.
 * Space update (:250-312): pair appended to C and H; C^-1 by the closed-form BFGS inverse
 *   (I - g dx de^T) C^-1 (I - g de dx^T) + g dx dx^T  expressed in the shared base (no inversion);
 * H^-1 by Woodbury (its diagonal is re-derived from C, :296).  With project_to_diagonal C^-1 is a Woodbury inverse too. */
int fh_cov_time_update(fh_context* ctx, const fh_cov_state* st, const double* x, const double* score, double shift_c,
                       double shift_h, double sigma_next2, int only_covariance, double* wx, double* ws,
                       double* mean_out, double* score_out, void* stream);
/* mean_x, mean_xn, x, xn: image-space d-vectors; s2 = sigma^2. */
int fh_cov_space_update(fh_context* ctx, const fh_cov_state* st, const double* mean_x, const double* mean_xn, double s2,
                        const double* x, const double* xn, void* stream);

/* The same two updates for the nimg images of a lock-step batch in ONE launch sequence (the reference asserts batch 1,
 * online_update_bfgs.py:161,255, and loops over images; BASELINE configs[1] advances 8 images together): the kernels of the
 * single-image entry points with the image as a grid dimension and per-image pointer tables, so an image's result is
 * bitwise the one fh_cov_time_update / fh_cov_space_update give it.  `sts`: nimg states that agree on d, m_c, m_h, ldm, ldg,
 * project, use_dct (else FH_EINVAL; the caller then updates image by image); their t0 / t1 / t2 are not used.
 * x, score, mean_out, score_out, mean_x, mean_xn, xn: [nimg][d] contiguous; work: 3 x [nimg][d] of scratch.
 * `ctx`: a context created with planes_max >= 3 nimg and m_cap >= 1 (Gram scratch for every image of the batch). */
int fh_cov_time_update_batched(fh_context* ctx, int nimg, const fh_cov_state* sts, const double* x, const double* score,
                               double shift_c, double shift_h, double sigma_next2, double* work, double* mean_out,
                               double* score_out, void* stream);
int fh_cov_space_update_batched(fh_context* ctx, int nimg, const fh_cov_state* sts, const double* mean_x,
                                const double* mean_xn, double s2, const double* x, const double* xn, double* work,
                                void* stream);

/* The m x m algebra of the Woodbury step above on the device (m <= 64; FH_ESIZE beyond, the caller then uses its host
 * path):  Mdst[:m,:m] = sym( -Msrc (I + G Msrc)^-1 ), Gauss-Jordan with partial pivoting in one workgroup.
 * Replaces the D2H copy + numpy.linalg.inv + H2D copy of an update, so that update_time_step / update_space_step
 * (online_update_bfgs.py:153-192, 250-312) run without a host round trip. */
int fh_woodbury_inner(fh_context* ctx, const double* Msrc, int ld_src, const double* G, int ldg, double* Mdst, int ld_dst,
                      int m, void* stream);

/* out = alpha*a + beta*b (b may be null when beta == 0); the few remaining elementwise steps of the
 * time update (mean' = x + sigma'^2 score', online_update_bfgs.py:178-180). */
int fh_axpby(double alpha, const double* a, double beta, const double* b, double* out, int64_t n, void* stream);
/* blocking copy of k <= 64 doubles from device scratch to host through the context's pinned buffer (one stream sync) */
int fh_read_scalars(fh_context* ctx, const double* scal, double* out_host, int k, void* stream);

/* measurement operators, measurement_utils/measurements.py:87-246 + utils_sisr.py:44-96 ----------
 * Circular 2-D convolution with a sparse tap list (the PSF non-zeros, centre at kernel_size/2):
 *   adjoint = 0:  out[p][i][j] = sum_t w[t] * in[p][(i*stride - dy[t]) mod S][(j*stride - dx[t]) mod S]
 *                 (out is [planes][S/stride][S/stride]; stride 4 = blur + decimate of the SR solver)
 *   adjoint = 1:  out[p][i][j] = sum_t w[t] * up(in)[(i + dy[t]) mod S][(j + dx[t]) mod S]
 *                 (in is [planes][S/stride][S/stride], zero-inserted on the fly)
 * halo = max(|dy|, |dx|) over the taps (<= 32); 1-D tap lists may pass -(h+1) (column kernel, all dx = 0) or
 * -(h+101) (row kernel, all dy = 0), 2-D lists 1000 + 64 * max|dy| + max|dx|, so that only the needed halo is staged
 * (the shipped motion PSF spans 58 x 16: a third of the square halo).
 * Equals ifft2(FB * fft2(x)).real / ifft2(conj(FB) * fft2(x)).real of the reference. */
int fh_conv_circ(fh_context* ctx, const double* in, double* out, const int32_t* dy, const int32_t* dx,
                 const double* w, int ntaps, int halo, int planes, int stride, int adjoint, void* stream);

/* the linear solve of conditioning_mechanisms.py:384-419 / 489-527 / 641-675 ------------------------ */
typedef struct fh_problem {
  int32_t op;            /* 0 inpainting, 1 blur (gaussian/motion), 2 super-resolution */
  int32_t use_dct;       /* 1: covariance lives in the DCT basis (CovarianceHessianBFGSDCT) */
  int32_t planes;        /* 3 * batch(=1) */
  int32_t stride;        /* SR scale factor, else 1 */
  int32_t ntaps;
  int32_t m;             /* factor columns in use */
  int32_t ldm;
  int32_t halo;          /* max(|dy|,|dx|) over the taps, or one of the encodings of fh_conv_circ */
  int64_t d;             /* planes*S*S */
  double sigma_y2;       /* measurement-noise variance after the reference's clips */
  const int32_t* tap_dy; /* [ntaps] */
  const int32_t* tap_dx;
  const double* tap_w;
  const double* mask;    /* [d] 0/1 (inpainting) */
  const double* D;       /* covariance rep: diag, row scale, base, inner matrix */
  const double* r;
  const double* B;
  const double* M;
  /* optional second pass of a separable PSF (k = col (x) row): A = conv(tap2) o conv(tap), ntaps2 = 0 when unused */
  int32_t ntaps2;
  int32_t halo2;
  /* 0: the reference's cg() (conditioning_utils/cg.py:118-292: x0 = b, stop checked after every iteration, break at
   * pAp <= 1e-16) - the customcuda solvers.  1: scipy.sparse.linalg.cg as the reference's scipy / customscipy solver
   * variants call it (x0 = 0, the initial residual is tested first - tol >= 1 returns 0 without iterating - and no pAp
   * test; conditioning_mechanisms.py:379, 444, 474, 548, 630, 696). */
  int32_t cg_scipy;
  /* 1: the four fold_* pointers below are PACKED HALF bases [2][S/2][S/2] of a basis with the DCT's mirror symmetry
   * P[k][S-1-n] = (-1)^k P[k][n] (a symmetric PSF): forward [Pe; Po] with Pe[j][n] = P[2j][n], Po[j][n] = P[2j+1][n];
   * inverse [Qe; Qo] with Qe[k][j] = P[2j][k], Qo[k][j] = P[2j+1][k] (k, n, j < S/2) - half the multiply-adds (k_dct_sym). */
  int32_t fold_sym;
  const int32_t* tap2_dy;
  const int32_t* tap2_dx;
  const double* tap2_w;
  /* optional: a separable blur folded into the DCT passes that follow / precede it inside A C A^T (use_dct = 1, op = 1).
   * With A(X) = F_col X F_row^T and P = C_dct F^T:  fold_fwd_w = P_row, fold_fwd_h = P_col  (dct2(A^T u) = P_col u P_row^T),
   * fold_inv_w = P_row^T, fold_inv_h = P_col^T  (A(idct2(v)) = P_col^T v P_row); each [S][S] float64, row-major.
   * All four null: the blur runs as tap-list passes.  free-hunch_amd/measurements.py: folded_dct_blur_bases. */
  const double* fold_fwd_w;
  const double* fold_fwd_h;
  const double* fold_inv_w;
  const double* fold_inv_h;
} fh_problem;

/* Per-image covariance pointers of a batched solve: B images share the operator, the tap lists, m, ldm and
 * sigma_y2 of an fh_problem (whose D, r, B, M, mask fields are then ignored) and differ in these. */
#define FH_MAX_BATCH 16
typedef struct fh_batch {
  int32_t nimg;
  int32_t pad;
  const double* D[FH_MAX_BATCH];
  const double* r[FH_MAX_BATCH];
  const double* B[FH_MAX_BATCH];
  const double* M[FH_MAX_BATCH];
  const double* mask[FH_MAX_BATCH];
} fh_batch;

typedef struct fh_cg_info {
  int32_t niter;
  int32_t optimal;
  double residual_norm;
  double b_norm;
} fh_cg_info;

/* fh_rep_apply for per->nimg images in one launch (grid z = image): z, out are [nimg][d]; image i uses per->D[i], r[i],
 * B[i], M[i] (all with m columns and leading dimension ldm).  The context must have been created with
 * planes_max >= 3 * nimg.  This is what the batched CG runs every iteration. */
int fh_rep_apply_batched(fh_context* ctx, const fh_batch* per, int ldm, const double* z, double* out, int64_t d, int m,
                         void* stream);

/* y -> A_mm(u) = sigma_y2*u + A C A^T u   (one application; exposed for tests) */
int fh_amm(fh_context* ctx, const fh_problem* p, const double* u, double* out, void* stream);
/* conditioning_utils/cg.py:118-292 with M = I, x0 = b: solves A_mm x = b, same stopping rule
 * (||r|| <= max(rtol*||b||, atol) -> optimal; pAp <= 1e-16 -> break; maxiter).  Blocks. */
int fh_cg_solve(fh_context* ctx, const fh_problem* p, const double* b, double* x, double rtol, double atol,
                int maxiter, fh_cg_info* info_host, void* stream);
/* The same for per->nimg independent systems advanced together (one kernel sequence serves all images; an image
 * that has met its stopping rule is skipped by every kernel).  b, x are [nimg][n]; rtol_host, info_host are
 * [nimg] host arrays; the context must have been created with planes_max >= 3 * nimg.  Each image follows exactly
 * the single-image iteration (same arithmetic, same stopping rule). */
int fh_cg_solve_batched(fh_context* ctx, const fh_problem* shared, const fh_batch* per, const double* b, double* x,
                        const double* rtol_host, double atol, int maxiter, fh_cg_info* info_host, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * UNet kernels (float32, activations NHWC [N][H][W][C] in HBM).  They replace the PyTorch modules of
 * training/openai_unet.py / openai_nn.py that the denoiser call and its input-VJP run through
 * (conditioning_mechanisms.py:239 forward, :280 autograd.grad):
 *   conv3x3 / conv1x1 / their input gradients   openai_unet.py:98,131,185,211,222,286,294   -> fh_conv2d_nhwc
 *   GroupNorm32 (+ scale/shift, + SiLU)          openai_nn.py:17-19, openai_unet.py:182-186,246-252 -> fh_groupnorm_*
 *   QKV attention (both channel orders)          openai_unet.py:337-354, 370-384          -> fh_bgemm_f32 + fh_softmax_*
 *   Upsample / AvgPool2d (resblock_updown and the Upsample / Downsample layers)  openai_unet.py:93-139 -> fh_resample2x
 *   torch.cat of the skip connections            openai_unet.py:683                       -> fh_concat_channels
 * ------------------------------------------------------------------------------------------------------------- */

/* Implicit-GEMM convolution on the fp32 matrix cores.  w is [Cout][KH*KW][Cin] (Cin innermost, Cin % 32 == 0);
 * out = conv(in, w) + bias (+ res), out/res are [N][Ho][Wo][Cout].  The input gradient of a stride-1 convolution is
 * the same call with the spatially flipped, in/out-transposed weight copy. */
int fh_conv2d_nhwc(const float* in, const float* w, const float* bias, const float* res, float* out, float* ws,
                   int ksplit, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride,
                   void* stream);
/* recommended K-split for a layer (1 = none).  With ksplit > 1 the K range is divided over blockIdx.z, raw partial
 * sums go to ws [ksplit][N*Ho*Wo][Cout] (caller-owned) and a second kernel adds them in a fixed order (+ bias, res). */
/* Precision of the convolution kernels that run on the bf16 matrix cores (fh_conv2d_x6_nhwc, fh_conv2d_x6_norm_nhwc):
 * 0 (default) = exact 3-way operand split, fp32 accuracy; 1 = plain bf16 compute (operands rounded to bf16, one product,
 * fp32 accumulation) - the reduced-precision UNet mode, counterpart of the reference's fp16 torso
 * (training/openai_fp16_util.py:15-32, flag path openai_preconditioning.py:171); 2 = the two leading bf16 planes of both
 * operands, three products (relative error ~ 2^-16: between TF32, the default convolution arithmetic of the reference's
 * CUDA path, and fp32) - half the matrix work of mode 0; 3 = the reference's fp16 torso arithmetic: operands rounded to IEEE
 * half precision, one product on v_mfma_f32_32x32x16_f16, fp32 accumulation (the weight operand is then ONE plane of
 * half-precision bit patterns in the same [taps][Cin/32][Cout][32] layout); 4 = half-split: both operands as TWO
 * half-precision planes h = rn(x 2^k), m = rn(x 2^k - h) (x carried to <= 2^-23 relative: within one fp32 ulp), three products
 * h h' + h m' + m h' on v_mfma_f32_32x32x16_f16 (the dropped m m' is <= 2^-22, 2^-24 rms, of a term: below the rounding noise
 * of the fp32 accumulation itself) - half the matrix work of mode 0 at the accuracy of an fp32 convolution.  The weight
 * operand is then [2 planes][taps][Cin/32][Cout][32] half bit patterns of w 2^kw followed by ONE float = 2^-kw; the activation
 * scale 2^k comes from fh_gn_epilogue.in_amax (fh_absmax_f32 of the input; required for fh_conv2d_x6_nhwc*) or is the fixed 2^4
 * of the fused GroupNorm input (fh_conv2d_x6_norm_nhwc*: values beyond +-4094 saturate).
 * Per calling host thread. */
int fh_unet_set_precision(int mode);
/* max over out[0 .. FH_AMAX_SLOTS) = max |x[i]| (all 0 for n = 0): the magnitude the half-split convolution scales its
 * activation operand by.  FH_AMAX_SLOTS partial maxima instead of one value because same-address atomics serialise. */
#define FH_AMAX_SLOTS 16
int fh_absmax_f32(const float* x, int64_t n, float* out, void* stream);
int fh_conv2d_splitk(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW);

/* Same convolution as fh_conv2d_nhwc at fp32 accuracy on the bf16 matrix cores: operands are split exactly into three
 * bf16 planes (x = h + m + l) and the six leading cross products are accumulated in fp32 (dropped terms < 2^-24 |x y|).
 * wx: weights split by the caller, bf16 [3 planes][KH*KW][Cin/32][Cout][32] (plane 0 = rn(w), 1 = rn(w - p0),
 * 2 = rn(w - p0 - p1); K chunks of 32 input channels are contiguous per output channel). */
int fh_conv2d_x6_nhwc(const float* in, const void* wx, const float* bias, const float* res, float* out, float* ws,
                      int ksplit, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride,
                      void* stream);

/* GroupNorm(+scale/shift)(+SiLU) fused into the convolution that consumes it (openai_unet.py:236-256: in_layers /
 * out_layers = GroupNorm32 -> SiLU -> conv3x3):  out = conv3x3(act(A[n][c] * in + B[n][c])) + bias (+ res), with
 * fh_groupnorm_table building [N][2][C] = (A, B) from the statistics exactly as fh_groupnorm_apply would (same bits).
 * Only where fh_conv2d_x6_norm_supported() returns 1 (3x3, stride 1, pad 1, row-aligned 128-pixel tiles that fill the
 * chip); elsewhere the caller runs fh_groupnorm_apply + fh_conv2d_*. */
int fh_groupnorm_table(const float* stats, const float* gamma, const float* beta, const float* scale, const float* shift,
                       int ss_stride, float* table, int N, int C, void* stream);
int fh_conv2d_x6_norm_supported(int N, int H, int W, int Cin, int Cout);
int fh_conv2d_x6_norm_nhwc(const float* in, const float* ab_table, int act, const void* wx, const float* bias,
                           const float* res, float* out, int N, int H, int W, int Cin, int Cout, void* stream);

/* Group-sum epilogue of the split-bf16 convolutions (fh_conv2d_x6_nhwc_gn / fh_conv2d_x6_norm_nhwc_gn): while a workgroup still
 * holds its output tile in registers it also forms the per-(image, GroupNorm group) sums that the NEXT GroupNorm pass over that
 * tensor would have to read it again for, and writes them as block partials [N][chunks][32][2] (double; chunks from
 * fh_conv2d_x6_gn_chunks, 0 = this layer's launch has no epilogue: split-K, thin outputs, tiles across two images):
 *   mode 0 - (sum v, sum v^2): the statistics of GroupNorm32 applied to the output (openai_nn.py:17-19; the out_layers /
 *            next block's in_layers norm of a ResBlock, openai_unet.py:236-256) -> fh_groupnorm_finalize(mode 0) = (mean, rstd);
 *   mode 1 - input-gradient convolutions: the output is dL/dy of a GroupNorm(+scale-shift)(+SiLU) whose forward input is `x`;
 *            (sum g, sum g xhat) with g = dy act'(t) gamma (1 + scale), the two reductions of its backward
 *            -> fh_groupnorm_finalize(mode 1), then fh_groupnorm_bwd_apply (the streaming pass alone).
 * Fixed summation order: the results are deterministic. */
typedef struct fh_gn_epilogue {
  double* partial;     /* [N * chunks][64]; null = no epilogue */
  const float* x;      /* mode 1: forward input of the GroupNorm, [N][Ho][Wo][Cout] */
  const float* tab;    /* mode 1: [N][5][Cout] from fh_groupnorm_bwd_table */
  int32_t mode, act;   /* act (mode 1): 1 = SiLU after the affine */
  const float* in_amax; /* precision mode 4: device [FH_AMAX_SLOTS], max = max |in| (fh_absmax_f32); partial may then be null */
} fh_gn_epilogue;
int fh_conv2d_x6_gn_chunks(int ksplit, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride);
int fh_conv2d_x6_nhwc_gn(const float* in, const void* wx, const float* bias, const float* res, float* out, float* ws,
                         int ksplit, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride,
                         const fh_gn_epilogue* epi, void* stream);
int fh_conv2d_x6_norm_nhwc_gn(const float* in, const float* ab_table, int act, const void* wx, const float* bias,
                              const float* res, float* out, int N, int H, int W, int Cin, int Cout, const fh_gn_epilogue* epi,
                              void* stream);
int fh_groupnorm_finalize(const double* partial, float* out, int N, int chunks, double count, int mode, void* stream);
int fh_groupnorm_bwd_table(const float* stats, const float* gamma, const float* beta, const float* scale, const float* shift,
                           int ss_stride, float* table, int N, int C, void* stream);
int fh_groupnorm_bwd_apply(const float* x, const float* dy, const float* stats, const float* sums, const float* gamma,
                           const float* beta, const float* scale, const float* shift, int ss_stride, float* dx, int N, int P,
                           int C, int act, int accumulate, void* stream);
int fh_groupnorm_bwd_sums(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                          const float* scale, const float* shift, int ss_stride, float* sums, double* scratch, int N, int P,
                          int C, int act, void* stream);
/* The same pass with the glue of the UNet's backward folded in: dx = acc_src + add2 + (GroupNorm backward term), where acc_src
 * (the gradient arriving over the block's skip path, openai_unet.py:256) and add2 (the gradient of a U-Net skip tensor, added
 * by autograd where `hs.append(h)` forked it, :663-671) may each be null; with dx2 != null the result is written as two
 * tensors, channels [0, csplit) -> dx [.., csplit] and [csplit, C) -> dx2 [.., C - csplit] - the gradient of
 * th.cat([h, hs.pop()], dim=1) (:682) without a split pass.  amax2 (nullable): [2][FH_AMAX_SLOTS] device floats the CALLER has
 * zeroed; the pass folds max |dx| into [0][*] and max |dx2| into [1][*] - what fh_absmax_f32 would leave for them, for the half-split
 * convolution (fh_unet_set_precision(4)) that consumes the gradient. */
int fh_groupnorm_bwd_apply_ex(const float* x, const float* dy, const float* stats, const float* sums, const float* gamma,
                              const float* beta, const float* scale, const float* shift, int ss_stride, const float* acc_src,
                              const float* add2, float* dx, float* dx2, int csplit, int N, int P, int C, int act, float* amax2,
                              void* stream);

/* 3x3 / stride 1 / pad 1 convolution with a thin output, Cout <= 8 (the 128 -> 6 output convolution and the
 * 128 -> 3 input gradient of the first one): direct form, w [Cout][9][Cin] as for fh_conv2d_nhwc, Cin % 32 == 0. */
int fh_conv3x3_thin_nhwc(const float* in, const float* w, const float* bias, float* out, int N, int H, int W, int Cin,
                         int Cout, void* stream);

/* 3x3 / stride 1 / pad 1 convolution through a fused 1-D Winograd F(2,3) transform along W (W even, Cin % 16 == 0):
 * 1.5x fewer multiplies than fh_conv2d_nhwc at the same exact-fp32 MFMA.  wu is the pre-transformed weight
 * [4][Cout][3][Cin]:  wu[0] = w[.,.,ky,0], wu[1] = (w0+w1+w2)/2, wu[2] = (w0-w1+w2)/2, wu[3] = w[.,.,ky,2] over kx.
 * out = conv(in, w) + bias (+ res). */
int fh_conv3x3_wino_nhwc(const float* in, const float* wu, const float* bias, const float* res, float* out, int N,
                         int H, int W, int Cin, int Cout, void* stream);

/* C[b] = alpha * opA(A[b]) * opB(B[b]), C [M][N] (ldc).  transA = 0: A is [M][K] (lda), 1: [K][M];
 * transB = 0: B is [N][K] (ldb), 1: [K][N].  Batch b in [0, batch): offsets (b / inner) * s?0 + (b % inner) * s?1. */
int fh_bgemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                 int transB, int batch, int inner, int64_t sA0, int64_t sA1, int64_t sB0, int64_t sB1, int64_t sC0,
                 int64_t sC1, float alpha, void* stream);

/* GroupNorm with 32 groups, eps = 1e-5, over x [N][P][C]:  stats [N][32][2] = (mean, rstd);
 * y = act(((x - mean) * rstd * gamma + beta) * (1 + scale[n][c]) + shift[n][c]);  scale/shift may be null;
 * scale[n] starts at scale + n * ss_stride;  act: 0 identity, 1 SiLU.
 * bwd: dx (+)= d/dx of the above applied to dy; sums is [N][32][2] scratch; `scratch` is caller-owned workspace. */
int64_t fh_groupnorm_scratch_doubles(int N, int P); /* size of `scratch` (float64 chunk partials) for the calls below */
int fh_groupnorm_stats(const float* x, float* stats, double* scratch, int N, int P, int C, void* stream);
int fh_groupnorm_apply(const float* x, const float* stats, const float* gamma, const float* beta, const float* scale,
                       const float* shift, int ss_stride, float* y, int N, int P, int C, int act, void* stream);
int fh_groupnorm_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                     const float* scale, const float* shift, int ss_stride, float* sums, double* scratch, float* dx, int N,
                     int P, int C, int act, int accumulate, void* stream);

/* in-place row softmax of s [rows][T];  backward in place on dp: dp <- p .* (dp - rowsum(dp .* p)) */
int fh_softmax_rows(float* s, int64_t rows, int T, void* stream);
int fh_softmax_bwd_rows(const float* p, float* dp, int64_t rows, int T, void* stream);

/* Fused self-attention of an AttentionBlock (training/openai_unet.py:296-305; head split and scaling of QKVAttentionLegacy
 * :337-354 for new_order = 0, of QKVAttention :370-384 for new_order = 1), fp32 on v_mfma_f32_32x32x2_f32, the T x T weight
 * matrix never written:
 *   qkv [N][T][3C] (the qkv convolution's NHWC output)  ->  out [N][T][C] = softmax(q k^T / sqrt(ch)) v per (image, head),
 *   lse [N * heads][T] = log2 sum_s exp2(log2(e) q_t k_s / sqrt(ch))   (kept for the backward instead of the weights).
 * Backward (the weights are recomputed from q, k and lse, as the reference's checkpointed block recomputes its forward):
 *   dqkv [N][T][3C] <- gradient w.r.t. qkv given dout = dL/dout; dsum [N * heads][T] is scratch (rowsum(dout . out)).
 * No atomics: results are deterministic.  Supported: T % 32 == 0, C / heads in {32, 64} (fh_attention_supported; else FH_ESIZE). */
int fh_attention_supported(int T, int C, int heads);
int fh_attention_fwd(const float* qkv, float* out, float* lse, int N, int T, int C, int heads, int new_order, void* stream);
int fh_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dsum, float* dqkv, int N,
                     int T, int C, int heads, int new_order, void* stream);

/* (Hs, Ws) = the SMALL side.  mode 0: 2x2 average pool big -> small; 1: its adjoint small -> big;
 * 2: nearest 2x upsample small -> big; 3: its adjoint big -> small; 4: zero insertion small -> big (value at (2h, 2w)). */
int fh_resample2x(const float* in, float* out, int N, int Hs, int Ws, int C, int mode, void* stream);

/* split = 0: out[p] = [a[p] | b[p]] over P pixels;  split = 1: a, b <- the two channel ranges of out */
int fh_concat_channels(float* a, float* b, float* out, int64_t P, int Ca, int Cb, int split, void* stream);

/* to_nhwc = 1: in [N][C][P] -> out [N][P][Cp] (channels C..Cp-1 zero);  0: in [N][P][Cp] -> out [N][C][P] */
int fh_layout_nchw_nhwc(const float* in, float* out, int N, int C, int64_t P, int Cp, int to_nhwc, void* stream);

/* out = a + b over n floats (n % 4 == 0) */
int fh_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);

/* ---- dense-matrix covariance path (float64, batched [bs][d][d] row-major) ------------------------------------------
 * Device side of the reference's dense update rules update_covariance / update_bfgs
 * (conditioning_utils/online_update_bfgs.py:377-463): the (bs,d,d) @ (bs,d,1) products and the outer-product updates.
 *   fh_dense_matvec: y[b] = alpha * op(A[b]) x[b] + beta * y[b]   (trans = 0: A, 1: A^T; x, y [bs][d]).
 *                    trans = 1 needs `scratch` of fh_dense_matvec_scratch_doubles(bs, d) doubles (fixed-order reduction).
 *   fh_dense_rank2:  out[b] = scale * (A[b] + a1[b] u1[b] v1[b]^T + a2[b] u2[b] v2[b]^T) + shift * I;
 *                    a1, a2 are DEVICE arrays [bs]; u?/v? [bs][d]; a null u? drops that term; out may alias A. */
int64_t fh_dense_matvec_scratch_doubles(int bs, int64_t d);
int fh_dense_matvec(const double* A, const double* x, double* y, double* scratch, int bs, int64_t d, int trans,
                    double alpha, double beta, void* stream);
int fh_dense_rank2(const double* A, double* out, int bs, int64_t d, const double* u1, const double* v1,
                   const double* a1, const double* u2, const double* v2, const double* a2, double scale, double shift,
                   void* stream);

/* Metrics of the sampling harness (generate_conditional.py:539-551): per-image PSNR and mean structural similarity of
 * uint8 [N][C][H][W] image batches, the published SSIM with the defaults of skimage.metrics.structural_similarity(x, y,
 * data_range=255, channel_axis=0) as the reference calls it - 7 x 7 uniform window, sample covariance, K1 = 0.01,
 * K2 = 0.03, windows that fit the image only, mean over positions then channels.  Window sums in exact integer arithmetic,
 * map values and means in float64; deterministic.  scratch: fh_metrics_scratch_doubles(N, C, H, W) doubles. */
int64_t fh_metrics_scratch_doubles(int N, int C, int H, int W);
int fh_metrics_u8(const uint8_t* a, const uint8_t* b, int N, int C, int H, int W, double* scratch, double* ssim_out,
                  double* psnr_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FH_HIP_H */
