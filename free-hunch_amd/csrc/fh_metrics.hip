// Image metrics of the sampling harness on the device (reference: generate_conditional.py:539-551 - per-image PSNR and
// skimage.metrics.structural_similarity(x, y, data_range=255, channel_axis=0) on uint8 CHW images).
//   k_metrics_tiles : one workgroup per 16 x 16 block of window positions of one (image, channel) plane.  The 22 x 22 pixel
//                     patch of both images goes through LDS; every thread forms the five 7 x 7 window sums of its position
//                     in EXACT integer arithmetic (49 x 255^2 < 2^22), evaluates the SSIM map value in float64 and the
//                     workgroup writes (sum of SSIM values, sum of squared pixel differences) as a block partial.
//   k_metrics_final : per image, fixed-order sums of the partials -> mean SSIM (over window positions, then channels) and
//                     PSNR = 10 log10(255^2 / mse).  Deterministic: no atomics.
#include "fh_common.h"

namespace {

constexpr int kMT = 16;       // window positions per tile edge
constexpr int kWin = 7;       // skimage default win_size
constexpr int kPatch = kMT + kWin - 1;

__global__ __launch_bounds__(256) void k_metrics_tiles(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int H, int W,
                                                       int tiles_x, int tiles_y, double* __restrict__ partial) {
  __shared__ int pa[kPatch][kPatch + 1], pb[kPatch][kPatch + 1];
  __shared__ double red[4], red2[4];
  const int plane = blockIdx.z;
  const int ty = blockIdx.y, tx = blockIdx.x;
  const uint8_t* A = a + (int64_t)plane * H * W;
  const uint8_t* B = b + (int64_t)plane * H * W;
  const int y0 = ty * kMT, x0 = tx * kMT;
  for (int i = threadIdx.x; i < kPatch * kPatch; i += 256) {
    const int py = i / kPatch, px = i % kPatch;
    const int gy = y0 + py, gx = x0 + px;
    const bool ok = gy < H && gx < W;
    pa[py][px] = ok ? (int)A[(int64_t)gy * W + gx] : 0;
    pb[py][px] = ok ? (int)B[(int64_t)gy * W + gx] : 0;
  }
  __syncthreads();
  const int ly = threadIdx.x / kMT, lx = threadIdx.x % kMT;
  const int oy = y0 + ly, ox = x0 + lx;  // window position = top-left corner; valid while the window fits
  double ssim = 0.0, sq = 0.0;
  if (oy + kWin <= H && ox + kWin <= W) {
    int sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
    for (int dy = 0; dy < kWin; ++dy)
#pragma unroll
      for (int dx = 0; dx < kWin; ++dx) {
        const int u = pa[ly + dy][lx + dx], v = pb[ly + dy][lx + dx];
        sx += u, sy += v, sxx += u * u, syy += v * v, sxy += u * v;
      }
    constexpr double np_ = kWin * kWin, cn = np_ / (np_ - 1.0);
    const double ux = sx / np_, uy = sy / np_;
    const double vx = cn * (sxx / np_ - ux * ux), vy = cn * (syy / np_ - uy * uy), vxy = cn * (sxy / np_ - ux * uy);
    const double c1 = (0.01 * 255.0) * (0.01 * 255.0), c2 = (0.03 * 255.0) * (0.03 * 255.0);
    ssim = ((2.0 * ux * uy + c1) * (2.0 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
  }
  // squared error: every pixel belongs to exactly one tile's 16 x 16 core
  if (oy < H && ox < W) {
    const int dd = pa[ly][lx] - pb[ly][lx];
    sq = (double)(dd * dd);
  }
  ssim = fh::block_sum_256(ssim, red);
  sq = fh::block_sum_256(sq, red2);
  if (threadIdx.x == 0) {
    double* dst = partial + (((int64_t)plane * tiles_y + ty) * tiles_x + tx) * 2;
    dst[0] = ssim, dst[1] = sq;
  }
}

__global__ __launch_bounds__(256) void k_metrics_final(const double* __restrict__ partial, int C, int tiles, int H, int W,
                                                       double* __restrict__ ssim_out, double* __restrict__ psnr_out) {
  __shared__ double red[4], red2[4];
  const int n = blockIdx.x;
  double ssim_mean = 0.0, sq_total = 0.0;
  for (int c = 0; c < C; ++c) {
    const double* p = partial + ((int64_t)(n * C + c) * tiles) * 2;
    double s = 0.0, q = 0.0;
    for (int t = threadIdx.x; t < tiles; t += 256) s += p[2 * t], q += p[2 * t + 1];
    s = fh::block_sum_256(s, red);
    q = fh::block_sum_256(q, red2);
    ssim_mean += s / ((double)(H - kWin + 1) * (W - kWin + 1));
    sq_total += q;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ssim_out[n] = ssim_mean / C;
    const double mse = sq_total / ((double)C * H * W);
    psnr_out[n] = 10.0 * log10(255.0 * 255.0 / (mse > 1e-12 ? mse : 1e-12));
  }
}

}  // namespace

extern "C" {

int64_t fh_metrics_scratch_doubles(int N, int C, int H, int W) {
  const int64_t tiles = (int64_t)((H + kMT - 1) / kMT) * ((W + kMT - 1) / kMT);
  return (int64_t)N * C * tiles * 2;
}

int fh_metrics_u8(const uint8_t* a, const uint8_t* b, int N, int C, int H, int W, double* scratch, double* ssim_out,
                  double* psnr_out, void* stream) {
  if (!a || !b || !scratch || !ssim_out || !psnr_out || N < 1 || C < 1 || H < kWin || W < kWin) return FH_EINVAL;
  const int tx = (W + kMT - 1) / kMT, ty = (H + kMT - 1) / kMT;
  if ((int64_t)N * C > 65535) return FH_ESIZE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_metrics_tiles, dim3(tx, ty, N * C), dim3(256), 0, st, a, b, H, W, tx, ty, scratch);
  hipLaunchKernelGGL(k_metrics_final, dim3(N), dim3(256), 0, st, (const double*)scratch, C, tx * ty, H, W, ssim_out, psnr_out);
  FH_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
