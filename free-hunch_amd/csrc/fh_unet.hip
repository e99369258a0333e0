// UNet kernels for gfx950 (CDNA4): float32, activations NHWC in HBM.
//
//  * k_conv_igemm   implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32):
//                   M = N*Ho*Wo pixels, N = Cout, K = taps*Cin with Cin innermost.  Used for 3x3 / 1x1 forward
//                   and - with the pre-flipped, transposed weight copy - for the input gradient (dgrad).
//                   Fused epilogue: + bias[co] (+ residual[pixel][co]).
//  * k_bgemm        strided batched GEMM (same MFMA core) for attention: QK^T, PV and the four backward products.
//  * GroupNorm(32) statistics / apply(+scale-shift, +SiLU) / backward, softmax forward / backward,
//    2x2 average pool and nearest 2x upsample (forward and adjoint), channel concat / split, NCHW<->NHWC.
//
// Tiling (wave64): a workgroup of 4 waves owns a (2*MI*32) x (2*NI*32) output tile, each wave MI x NI
// accumulator tiles of 32x32 (16 VGPRs each).  K advances in chunks of 32 floats staged through LDS with both
// operands k-contiguous and a row stride of 36 floats, which makes the ds_read_b128 fragment reads
// conflict-free; one b128 read feeds four MFMAs (k order inside a group of 8 is permuted identically for A
// and B: lane half h supplies k = 4h + s at step s).  Global loads for chunk c+1 are issued before the MFMAs
// of chunk c and written to the other LDS buffer afterwards: one barrier per chunk.
#include "fh_common.h"
#include <stdlib.h>

typedef float float16_t __attribute__((ext_vector_type(16)));

namespace {

constexpr int kBK = 32;       // K chunk (floats)
constexpr int kLd = kBK + 4;  // LDS row stride in floats (144 B)

struct ConvArgs {
  const float* in;    // [N][H][W][Cin]
  const float* w;     // [Cout][taps][Cin]
  const float* bias;  // [Cout] or null
  const float* res;   // [N][Ho][Wo][Cout] or null
  float* out;         // [N][Ho][Wo][Cout]
  int N, H, W, Cin, Cout, KH, KW, pad, stride, Ho, Wo;
  int ksplit;   // > 1: blockIdx.z owns a contiguous range of K chunks and writes raw partial sums to ws[z][M][Cout]
  float* ws;
};

// XCD-aware tile order.  Workgroups are dispatched round-robin over the 8 XCDs (linear id % 8), each with its own L2.
// Remap so that one XCD walks a contiguous range of pixel tiles, visiting all Cout tiles of a pixel tile back to back:
// neighbouring pixel tiles (which share the 3x3 halo rows) and the weight panel then hit in that XCD's L2.
__device__ __forceinline__ void xcd_tile(int& tile_m, int& tile_n) {
  const unsigned gx = gridDim.x, gy = gridDim.y;
  const unsigned total = gx * gy;
  unsigned lin = blockIdx.y * gx + blockIdx.x;
  if ((total & 7u) == 0) lin = (lin & 7u) * (total >> 3) + (lin >> 3);
  tile_n = (int)(lin % gy);
  tile_m = (int)(lin / gy);
}

template <int MI, int NI>
__global__ __launch_bounds__(256) void k_conv_igemm(ConvArgs a) {
  constexpr int BM = 2 * MI * 32, BN = 2 * NI * 32;
  constexpr int TPR_A = 256 / BM, TPR_B = 256 / BN;    // threads per staged row
  constexpr int FA = kBK / TPR_A, FB = kBK / TPR_B;    // floats per thread per chunk
  __shared__ __align__(16) float As[2][BM][kLd];
  __shared__ __align__(16) float Bs[2][BN][kLd];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * (MI * 32), wn = (wave & 1) * (NI * 32);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t M = (int64_t)a.N * a.Ho * a.Wo;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int taps = a.KH * a.KW;
  const int cpt = a.Cin / kBK;  // chunks per tap
  const int nchunks_all = taps * cpt;
  const int c_begin = (int)((int64_t)nchunks_all * blockIdx.z / a.ksplit);
  const int c_end = (int)((int64_t)nchunks_all * (blockIdx.z + 1) / a.ksplit);

  // staging coordinates (fixed over the K loop)
  const int ra = tid / TPR_A, ca = (tid % TPR_A) * FA;
  const int rb = tid / TPR_B, cb = (tid % TPR_B) * FB;
  const int64_t pm = m0 + ra;
  const bool pvalid = pm < M;
  int pn = 0, pho = 0, pwo = 0;
  if (pvalid) {
    pn = (int)(pm / ((int64_t)a.Ho * a.Wo));
    const int rem = (int)(pm % ((int64_t)a.Ho * a.Wo));
    pho = rem / a.Wo;
    pwo = rem % a.Wo;
  }
  const int co_b = n0 + rb;
  const bool bvalid = co_b < a.Cout;
  const float* wrow = a.w + (int64_t)(bvalid ? co_b : 0) * taps * a.Cin + cb;

  float4 ra_reg[FA / 4], rb_reg[FB / 4];

  auto load_chunk = [&](int c) {
    const int tap = c / cpt, c0 = (c % cpt) * kBK;
    const int ky = tap / a.KW, kx = tap % a.KW;
    const int hi = pho * a.stride + ky - a.pad, wi = pwo * a.stride + kx - a.pad;
    const bool ok = pvalid && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
    const float* src = a.in + (((int64_t)pn * a.H + (ok ? hi : 0)) * a.W + (ok ? wi : 0)) * a.Cin + c0 + ca;
#pragma unroll
    for (int e = 0; e < FA / 4; ++e)
      ra_reg[e] = ok ? *reinterpret_cast<const float4*>(src + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wsrc = wrow + (int64_t)tap * a.Cin + c0;
#pragma unroll
    for (int e = 0; e < FB / 4; ++e)
      rb_reg[e] = bvalid ? *reinterpret_cast<const float4*>(wsrc + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int e = 0; e < FA / 4; ++e) *reinterpret_cast<float4*>(&As[buf][ra][ca + 4 * e]) = ra_reg[e];
#pragma unroll
    for (int e = 0; e < FB / 4; ++e) *reinterpret_cast<float4*>(&Bs[buf][rb][cb + 4 * e]) = rb_reg[e];
  };

  float16_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_chunk(c_begin);
  store_chunk(0);
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    if (c + 1 < c_end) load_chunk(c + 1);
#pragma unroll
    for (int g = 0; g < kBK / 8; ++g) {
      float4 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(&As[buf][wm + i * 32 + lr][g * 8 + 4 * lh]);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[buf][wn + j * 32 + lr][g * 8 + 4 * lh]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const float av = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const float bv = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    if (c + 1 < c_end) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + wn + j * 32 + lr;
    if (co >= a.Cout) continue;
    const float bv = (a.bias != nullptr && a.ksplit == 1) ? a.bias[co] : 0.f;
    float* dst = a.ksplit == 1 ? a.out : a.ws + (int64_t)blockIdx.z * M * a.Cout;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M) {
          float v = acc[i][j][r] + bv;
          if (a.res != nullptr && a.ksplit == 1) v += a.res[row * a.Cout + co];
          dst[row * a.Cout + co] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fp32 convolution on the bf16 matrix cores by exact operand splitting ("bf16x6").
//   Every fp32 value is the exact sum of three bf16 numbers, x = h + m + l (h = rn(x), m = rn(x - h), l = rn(x - h - m);
//   three signed 8-bit significands cover the 24-bit fp32 significand).  The product of two split operands has nine
//   terms, each exact in fp32; the six largest (hh', hm', mh', hl', lh', mm') are accumulated in fp32 by
//   v_mfma_f32_32x32x16_bf16 and the three dropped ones are below 2^-24 |x y| - the size of one fp32 rounding - so the
//   result carries the error of an fp32 dot product (tests: error against float64 on par with the fp32-MFMA kernel).
//   Six bf16 MFMAs (32 cycles each, K = 16) replace eight fp32 MFMAs (64 cycles each, K = 2): 2.67x the fp32 matrix
//   peak.  gfx950 has no xf32 path, so this is the only way past 157 TFLOP/s at fp32 accuracy.
// Same implicit-GEMM tiling as k_conv_igemm (4 waves, (2*MI*32) x (2*NI*32) tile, K chunks of 32 channels).  The
// activation operand is split while it is staged (global fp32 -> registers -> 3 bf16 planes in LDS; ~6 VALU per
// element, hidden under the MFMAs); the weights are split once on the host, wx [3 planes][taps][Cin/32][Cout][32]
// bf16, so that the weight tile of a K chunk is one contiguous run of 64-byte rows.
// LDS rows are 32 bf16 + 8 pad = 80 B, so the 16-byte fragment reads (8 consecutive k per lane) are conflict-free;
// one buffer (61 KB for 128 x 128) keeps two workgroups per CU, which hide each other's staging barriers.
// Measured on MI355X (128 x 128 tile, 8 x 128 x 128 x 256 -> 256, fp32-equivalent TFLOP/s): LDS reads + MFMAs alone 265
// (the chip clocks down under dense bf16 MFMA), + split / LDS writes 217, + global loads 172 (this kernel).  Variants
// that trade occupancy for overlap lost: two LDS buffers with the wave halves running the chunk's phases in opposite
// order (one workgroup per CU) 155; producer / consumer wave specialisation 140; two-deep register prefetch spills.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int kXLd = kBK + 8;  // bf16 row stride (80 B)

struct ConvArgsX {
  const float* ab;    // [N][2][Cin] GroupNorm affine table (A plane, B plane) of the fused-input form, else null
  int act;            // fused input: 1 = SiLU after the affine
  const float* in;
  const __bf16* wx;   // [3 planes][taps][Cin/32][Cout][32]
  const float* bias;
  const float* res;
  float* out;
  int N, H, W, Cin, Cout, KH, KW, pad, stride, Ho, Wo;
  int ksplit;
  float* ws;
  // optional group-sum epilogue (fh_gn_epilogue): per-(image, GroupNorm group) sums of the output tile as block partials
  double* gn_partial;   // [N][gn_chunks][32][2] or null
  const float* gn_x;    // mode 1: forward input of the GroupNorm whose backward consumes this output
  const float* gn_tab;  // mode 1: [N][5][Cout]  A, Bc, mean, rstd, gamma (1 + scale)
  int gn_mode, gn_act, gn_chunks;
  // half-split mode (NP = 2 on the f16 MFMA): the activation operand is staged as x * 2^k; k from the tensor's largest
  // magnitude (in_amax, device) or, for the fused-GroupNorm input, the fixed in_scale
  const float* in_amax;
  float in_scale;
};

// SiLU with the hardware reciprocal (1 ulp) instead of an IEEE division (~10 instructions): the fused-input convolution
// evaluates it while staging its activation tile, where every VALU cycle is exposed; the stand-alone apply kernel uses the
// same function, so the two paths stay bit-identical
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

// Group-sum epilogue of the split-bf16 convolutions.  A lane of the 32 x 32 MFMA layout holds 16 rows x MI tiles of ONE output
// channel per column tile j, so the sums of a channel are formed in registers (double), the two lane halves are folded with a
// shuffle, every wave leaves its 32 x NI channel sums in LDS and 64 threads add them per (group, which) in a FIXED order
// (deterministic) and write the block's partial: the statistics pass of the consuming GroupNorm (mode 0: sum v, sum v^2) or the
// two sums of its backward (mode 1: sum g, sum g xhat with g = dy act'(t) gamma (1 + scale)) never read the tensor again.
struct GnAcc {  // per lane: <= 32 values of one channel, summed in fp32 (then widened for the cross-lane / cross-tile sums)
  float s0, s1;
};
__device__ __forceinline__ void gn_accum(const ConvArgsX& a, GnAcc& g, float v, int64_t row, int co, const float (&tb)[5]) {
  if (a.gn_mode == 0) {
    g.s0 += v;
    g.s1 = fmaf(v, v, g.s1);
  } else {
    const float x = a.gn_x[row * a.Cout + co];
    const float t = fmaf(x, tb[0], tb[1]);
    const float xh = (x - tb[2]) * tb[3];
    float gg = v;
    if (a.gn_act) {
      const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-t));  // (1 ulp reciprocal: these are sums of ~1e5 terms)
      gg *= sg * fmaf(t, 1.f - sg, 1.f);
    }
    gg *= tb[4];
    g.s0 += gg;
    g.s1 = fmaf(gg, xh, g.s1);
  }
}
template <int NI, int WM, int WN, int BN>
__device__ __forceinline__ void gn_reduce(const ConvArgsX& a, const GnAcc (&g)[NI], double* gred, int n, int chunk, int n0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const double s0 = (double)g[j].s0 + (double)__shfl_xor(g[j].s0, 32, 64);
    const double s1 = (double)g[j].s1 + (double)__shfl_xor(g[j].s1, 32, 64);
    if (lane < 32) {
      gred[((wave * 32 + lr) * NI + j) * 2 + 0] = s0;
      gred[((wave * 32 + lr) * NI + j) * 2 + 1] = s1;
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int G = tid >> 1, which = tid & 1, cg = a.Cout / 32;
    int c0 = G * cg, c1 = c0 + cg;
    c0 = c0 < n0 ? n0 : c0;
    c1 = c1 > n0 + BN ? n0 + BN : c1;
    c1 = c1 > a.Cout ? a.Cout : c1;
    double sum = 0.0;
    for (int c = c0; c < c1; ++c) {
      const int cl = c - n0, wn_i = cl / (NI * 32), j = (cl % (NI * 32)) / 32, l = cl % 32;
      for (int wm_i = 0; wm_i < WM; ++wm_i) sum += gred[(((wm_i * WN + wn_i) * 32 + l) * NI + j) * 2 + which];
    }
    a.gn_partial[((int64_t)n * a.gn_chunks + chunk) * 64 + tid] = sum;
  }
}

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  float r = x - (float)h;
  m = (__bf16)r;
  r -= (float)m;
  l = (__bf16)r;
}

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// the 16-bit pattern of x rounded to half precision, carried in a __bf16 slot (LDS tiles and fragments are typed __bf16)
__device__ __forceinline__ __bf16 half_bits(float x) {
  const _Float16 h = (_Float16)x;
  return __builtin_bit_cast(__bf16, h);
}
template <bool F16>
__device__ __forceinline__ float16_t mfma16(bf16x8_t a, bf16x8_t b, float16_t c) {
  if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Half-split mode (NP = 2 with F16): x 2^k = h + m with h = rn_half(x 2^k), m = rn_half(x 2^k - h): h + m carries x to
// <= 2^-23 relative (the residual has at most 13 significant bits, the half keeps 11 of them: at most one unit of the
// fp32 last place is lost, and nothing for three operands out of four) as long as m stays a normal half, i.e. for
// |x| 2^k >= 2^-3; smaller elements keep an absolute error of 2^-25 / 2^k.  The three products h h' + h m' + m h' drop
// m m' (<= 2^-22 |x x'|, 2^-24 rms): in total below the rounding noise of the fp32 accumulation that every fp32
// convolution carries (sqrt(K) 2^-25 of a term at K = 9 Cin), measured in tests/test_hip_unet.py against float64.
struct SplitScale {
  float in, inv_in, winv;
};
template <int NP, bool F16>
__device__ __forceinline__ SplitScale split_scale(const ConvArgsX& a, int64_t wplane) {
  SplitScale s{1.f, 1.f, 1.f};
  if (F16 && NP == 2) {
    s.winv = *reinterpret_cast<const float*>(a.wx + 2 * wplane);  // 1 / (power-of-two scale of the weight planes), behind them
    if (a.in_amax != nullptr) {  // largest magnitude (the max over the FH_AMAX_SLOTS partial maxima) -> [2^11, 2^12)
      unsigned mb = 0u;
#pragma unroll
      for (int q = 0; q < FH_AMAX_SLOTS; ++q) mb = max(mb, __float_as_uint(a.in_amax[q]));
      int e = (int)((mb >> 23) & 255u) - 127;
      e = e < -110 ? -110 : (e > 126 ? 126 : e);
      s.in = __uint_as_float((unsigned)(127 + 11 - e) << 23);
      s.inv_in = __uint_as_float((unsigned)(127 - 11 + e) << 23);
    } else {
      s.in = a.in_scale;
      s.inv_in = 1.f / a.in_scale;  // (a power of two)
    }
  }
  return s;
}
__device__ __forceinline__ void split_half2(float xs, __bf16& h, __bf16& m) {
  xs = fminf(fmaxf(xs, -65504.f), 65504.f);  // saturate instead of overflowing (fixed-scale inputs only)
  const _Float16 hh = (_Float16)xs;
  h = __builtin_bit_cast(__bf16, hh);
  m = half_bits(xs - (float)hh);
}

// NP = 3: the exact split above (six products).  NP = 2: the two leading planes, three products (h h' + h m' + m h'):
// operands carried to 2^-17, product terms below 2^-16 dropped - between TF32 (2^-11, the default convolution arithmetic of
// the reference's CUDA path) and fp32.  NP = 1: plain bf16 compute - operands rounded to bf16 (plane 0 = rn(x)),
// one product, fp32 accumulation: the reduced-precision mode of the UNet (fh_unet_set_precision; NOT an fp32-parity mode).
// F16 (with NP = 1): the single plane holds the operands rounded to IEEE half precision (11-bit significand) and the product
// runs on v_mfma_f32_32x32x16_f16 - the arithmetic of the reference's use_fp16 torso (openai_fp16_util.py:15-32: convolution
// weights and inputs in float16), with fp32 accumulation and fp32 storage between the layers.
template <int MI, int NI, int WM, int WN, int MINW, int NP = 3, bool F16 = false>  // WM x WN waves, each MI x NI accumulator tiles of 32 x 32
__global__ __launch_bounds__(64 * WM * WN, MINW) void k_conv_x6(ConvArgsX a) {
  constexpr int T = 64 * WM * WN;
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  constexpr int IA = BM * 8 / T;      // float4 items per thread per chunk (A tile = BM rows x 8 float4)
  constexpr int IB = (NP * BN * 4 + T - 1) / T;  // 16-byte items per thread per chunk (B tile = NP planes x BN rows x 4 parts)
  static_assert(BM * 8 % T == 0, "tile / thread-count mismatch");
  __shared__ __align__(16) __bf16 As[NP][BM][kXLd];
  __shared__ __align__(16) __bf16 Bs[NP][BN][kXLd];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WN) * (MI * 32), wn = (wave % WN) * (NI * 32);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t M = (int64_t)a.N * a.Ho * a.Wo;
  int tile_m, tile_n;
  xcd_tile(tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;
  const int taps = a.KH * a.KW;
  const int cpt = a.Cin / kBK;
  const int nchunks_all = taps * cpt;
  const int c_begin = (int)((int64_t)nchunks_all * blockIdx.z / a.ksplit);
  const int c_end = (int)((int64_t)nchunks_all * (blockIdx.z + 1) / a.ksplit);

  // Staging maps: consecutive lanes take consecutive 16-byte pieces, so a wave-load covers whole 128-byte lines
  // (8 lanes per pixel row of 32 floats) instead of 32 partial lines - the vector-memory path, not the MFMA, bounds
  // this kernel.  Item e of a thread is element e * 256 + tid of the tile.
  int a_row[IA], a_n[IA], a_ho[IA], a_wo[IA];
  bool a_ok[IA];
  const int a_c4 = (tid & 7) * 4;
#pragma unroll
  for (int e = 0; e < IA; ++e) {
    a_row[e] = (e * T + tid) >> 3;
    const int64_t pm = m0 + a_row[e];
    a_ok[e] = pm < M;
    a_n[e] = a_ho[e] = a_wo[e] = 0;
    if (a_ok[e]) {
      a_n[e] = (int)(pm / ((int64_t)a.Ho * a.Wo));
      const int rem = (int)(pm % ((int64_t)a.Ho * a.Wo));
      a_ho[e] = rem / a.Wo;
      a_wo[e] = rem % a.Wo;
    }
  }
  // weights: wx [3][taps][Cin/32][Cout][32] - the B tile of one plane is BN contiguous 64-byte rows
  const int64_t wplane = (int64_t)taps * cpt * a.Cout * kBK;
  const SplitScale ssc = split_scale<NP, F16>(a, wplane);
  int b_pl[IB], b_row[IB];
  bool b_ok[IB];
  const int b_part = (tid & 3) * 8;
#pragma unroll
  for (int e = 0; e < IB; ++e) {
    const int idx = e * T + tid;
    b_pl[e] = idx < NP * BN * 4 ? idx / (BN * 4) : -1;
    b_row[e] = (idx % (BN * 4)) >> 2;
    b_ok[e] = b_pl[e] >= 0 && n0 + b_row[e] < a.Cout;
  }

  float4 ra_reg[IA];
  bf16x8_t rb_reg[IB];

  auto load_chunk = [&](int c) {
    const int cc = c / taps, tap = c % taps;  // channel chunk outer, taps inner: the 9 taps re-read the same lines while hot in L2
    const int ky = tap / a.KW, kx = tap % a.KW;
#pragma unroll
    for (int e = 0; e < IA; ++e) {
      const int hi = a_ho[e] * a.stride + ky - a.pad, wi = a_wo[e] * a.stride + kx - a.pad;
      const bool ok = a_ok[e] && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
      const float* src = a.in + (((int64_t)a_n[e] * a.H + (ok ? hi : 0)) * a.W + (ok ? wi : 0)) * a.Cin + cc * kBK + a_c4;
      ra_reg[e] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const __bf16* wbase = a.wx + ((int64_t)tap * cpt + cc) * a.Cout * kBK;
#pragma unroll
    for (int e = 0; e < IB; ++e) {
      bf16x8_t z;
#pragma unroll
      for (int q = 0; q < 8; ++q) z[q] = (__bf16)0.f;
      const __bf16* src = wbase + (b_ok[e] ? b_pl[e] : 0) * wplane + (int64_t)(b_ok[e] ? n0 + b_row[e] : 0) * kBK + b_part;
      rb_reg[e] = b_ok[e] ? *reinterpret_cast<const bf16x8_t*>(src) : z;
    }
  };
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  auto store_chunk = [&]() {
#pragma unroll
    for (int e = 0; e < IA; ++e) {
      bf16x4_t h4, m4, l4;
      const float4 v = ra_reg[e];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x = q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
        __bf16 h, m, l;
        if (F16 && NP == 2) {
          split_half2(x * ssc.in, h, m);
          l = m;
        } else {
          split3(x, h, m, l);
          if (F16) h = half_bits(x);
        }
        h4[q] = h, m4[q] = m, l4[q] = l;
      }
      *reinterpret_cast<bf16x4_t*>(&As[0][a_row[e]][a_c4]) = h4;
      if (NP >= 2) *reinterpret_cast<bf16x4_t*>(&As[NP >= 2 ? 1 : 0][a_row[e]][a_c4]) = m4;
      if (NP == 3) *reinterpret_cast<bf16x4_t*>(&As[NP - 1][a_row[e]][a_c4]) = l4;
    }
#pragma unroll
    for (int e = 0; e < IB; ++e)
      if (b_pl[e] >= 0) *reinterpret_cast<bf16x8_t*>(&Bs[b_pl[e]][b_row[e]][b_part]) = rb_reg[e];
  };

  float16_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&]() {
#pragma unroll
  for (int ks = 0; ks < kBK / 16; ++ks) {
    const int ko = ks * 16 + 8 * lh;
    bf16x8_t af[NP][MI], bfr[NP][NI];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[p][i] = *reinterpret_cast<const bf16x8_t*>(&As[p][wm + i * 32 + lr][ko]);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[p][j] = *reinterpret_cast<const bf16x8_t*>(&Bs[p][wn + j * 32 + lr][ko]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        float16_t t = acc[i][j];
        if (NP == 3) {  // smallest terms first: l h', h l', m m'
          t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NP - 1][i], bfr[0][j], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[NP - 1][j], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NP >= 2 ? 1 : 0][i], bfr[NP >= 2 ? 1 : 0][j], t, 0, 0, 0);
        }
        if (NP >= 2) {  // m h', h m'
          t = mfma16<F16>(af[NP >= 2 ? 1 : 0][i], bfr[0][j], t);
          t = mfma16<F16>(af[0][i], bfr[NP >= 2 ? 1 : 0][j], t);
        }
        t = mfma16<F16>(af[0][i], bfr[0][j], t);  // h h'
        acc[i][j] = t;
      }
  }
  };

  // registers prefetch chunk c + 1 while chunk c is multiplied; one LDS buffer (two workgroups per CU overlap instead)
  load_chunk(c_begin);
  store_chunk();
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    if (c + 1 < c_end) load_chunk(c + 1);
    compute();
    __syncthreads();
    if (c + 1 < c_end) {
      store_chunk();
      __syncthreads();
    }
  }

  __shared__ double gred[WM * WN * 32 * NI * 2];
  const bool gn = a.gn_partial != nullptr;  // (host: only with ksplit == 1 and tiles that lie inside one image)
  const int gn_n = gn ? (int)(m0 / ((int64_t)a.Ho * a.Wo)) : 0;
  GnAcc gsum[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    gsum[j].s0 = gsum[j].s1 = 0.f;
    const int co = n0 + wn + j * 32 + lr;
    if (co >= a.Cout) continue;
    const float bv = (a.bias != nullptr && a.ksplit == 1) ? a.bias[co] : 0.f;
    float* dst = a.ksplit == 1 ? a.out : a.ws + (int64_t)blockIdx.z * M * a.Cout;
    float tb[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (gn && a.gn_mode == 1) {
#pragma unroll
      for (int q = 0; q < 5; ++q) tb[q] = a.gn_tab[((int64_t)gn_n * 5 + q) * a.Cout + co];
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M) {
          float v = acc[i][j][r];
          if (F16 && NP == 2) v = v * ssc.inv_in * ssc.winv;
          v += bv;
          if (a.res != nullptr && a.ksplit == 1) v += a.res[row * a.Cout + co];
          dst[row * a.Cout + co] = v;
          if (gn) gn_accum(a, gsum[j], v, row, co, tb);
        }
      }
    }
  }
  if (gn) {
    const int tiles_per_img = (int)(((int64_t)a.Ho * a.Wo) / BM);
    gn_reduce<NI, WM, WN, BN>(a, gsum, gred, gn_n, (tile_m % tiles_per_img) * (int)gridDim.y + tile_n, n0);
  }
}

// ------------------------------------------------------------------------------------------------
// Row-reuse form of k_conv_x6 for 3x3 / stride 1 / pad 1 layers whose 128-pixel tiles are whole row segments (W % 128 == 0,
// or W = 64 / 32 with the tile spanning 2 / 4 full rows): the 256^2 ... 32^2 levels of the UNet.  The three taps of a kernel row read the same
// input pixels shifted by one, so the activation tile is staged ONCE per (channel chunk, kernel row) with a one-pixel halo
// on each side of every row segment (130 ... 136 pixels) and the kx = 0, 1, 2 products read it at row offsets 0, 1, 2: global loads, bf16 splits and
// LDS writes of the activation operand drop 3x; only the weight tile changes per tap.  Same tile shape as the generic
// kernel (128 pixels x 128 channels, 8 waves of 64 x 32), same fused epilogue, K order (channel chunk, ky, kx).
// ------------------------------------------------------------------------------------------------
// NORM: the input is GroupNorm(+scale/shift)(+SiLU) of `in`, applied while the tile is staged: y = act(A[n][c] x + B[n][c])
// with the same fmaf / SiLU as k_gn_stream (bit-identical to the two-kernel path); padding pixels stay exactly zero.  The
// separate apply pass (one read + one write of the activation) and the normalised tensor itself disappear; the affine is
// re-evaluated 3 x (Cout / 128) times per element, which the MFMAs hide.
// BM = 256 (16 waves, one workgroup per CU): the kernel is bound by the ~12 B/clk a CU can pull from L2, and per K step it
// pulls 1.33 B per pixel (fp32, every third tap) but 6 B per output channel (three bf16 planes, every tap): arithmetic
// intensity 2 BM BN / (1.33 BM + 6 BN) = 35 flop/B at 128 x 128 (-> ~190 TFLOP/s fp32-equivalent, as measured) and 59 at
// 256 x 128, past the 50 flop/B where the MFMAs at the held clock become the limit.
// GL: the weight tile arrives by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write pass) into one of TWO
// buffers, issued a tap ahead and retired by the single barrier of the tap (one barrier per tap instead of two).  The DMA
// image is lane-linear (64-byte rows, no padding), so the bank spread of the 16-byte fragment reads comes from a swizzle of
// the 16-byte chunk index with bits 2..3 of the row, applied on the per-lane SOURCE address and on the read.
// GL = 2 (deep pipeline, half-split mode): with three products per tap instead of six a tap's MFMAs (~0.8 us) no longer
// cover the L2 latency of the next tap's weight DMA, nor does the exposed activation staging hide behind anything.  So the
// weight tiles go through a ring of THREE buffers issued two taps ahead, the activation tile is double-buffered (global
// loads at the first tap of a kernel row, split + LDS write after the third tap's products, into the buffer the row before
// last used), and the one barrier per tap waits with an explicit `s_waitcnt vmcnt(n)` that leaves the younger DMA / loads
// in flight (a __syncthreads() would drain them).  Every wave issues exactly one DMA per tap and IA (+2 table) loads per
// row, unconditionally, so the counts are exact.
template <int SEGW, bool NORM = false, int BM = 128, int NP = 3, int GL = 0, bool F16 = false>  // SEGW: pixels per row segment of the tile (W, capped at BM)
__global__ __launch_bounds__(4 * BM, 4) void k_conv_x6r(ConvArgsX a) {
  constexpr int MI = 2, NI = 1, WN = 4, T = 4 * BM;
  constexpr int NSEG = BM / SEGW, SP = SEGW + 2;  // row segments per tile, staged pixels per segment (one halo each side)
  constexpr int BN = 128, AR = NSEG * SP;
  constexpr int IA = (AR * 8 + T - 1) / T;  // float4 items per thread for the activation tile (1040 items)
  constexpr int IB = (NP * BN * 4 + T - 1) / T;  // 16-byte items per thread for the weight tile (1536 items at NP = 3)
  constexpr int NB = GL == 2 ? 3 : (GL ? 2 : 1);  // (a second REGISTER-staged weight buffer was measured: no gain)
  constexpr int NA = GL == 2 ? 2 : 1;
  static_assert(GL != 2 || (NP * 8 == T / 64), "deep pipeline: one weight DMA per wave and tap");
  constexpr int BLD = GL ? kBK : kXLd;    // weight-row stride in LDS
  // ONE __shared__ object for everything: beside an LDS-DMA staging array a second __shared__ object makes hipcc wait
  // vmcnt(0) before the first ds_read of every step, i.e. serialises the DMA with the MFMAs
  constexpr int kAsBytes = NA * NP * AR * kXLd * 2, kBsBytes = NB * NP * BN * BLD * 2;
  constexpr int kGredBytes = (BM / 64) * WN * 32 * NI * 2 * 8;
  __shared__ __align__(16) unsigned char smem[kAsBytes + kBsBytes + kGredBytes];
  __bf16 (*As)[NP][AR][kXLd] = reinterpret_cast<__bf16 (*)[NP][AR][kXLd]>(smem);
  __bf16 (*Bs)[NP][BN][BLD] = reinterpret_cast<__bf16 (*)[NP][BN][BLD]>(smem + kAsBytes);
  double* gred = reinterpret_cast<double*>(smem + kAsBytes + kBsBytes);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WN) * (MI * 32), wn = (wave % WN) * (NI * 32);
  const int lr = lane & 31, lh = lane >> 5;
  int tile_m, tile_n;
  xcd_tile(tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;
  const int cpt = a.Cin / kBK;
  const int nchunks = cpt * 9;
  // the tile is NSEG consecutive row segments of one image: rows py .. py + NSEG - 1, columns px0 .. px0 + SEGW - 1
  const int pn = (int)(m0 / ((int64_t)a.H * a.W));
  const int rem = (int)(m0 % ((int64_t)a.H * a.W));
  const int py = rem / a.W, px0 = rem % a.W;

  int a_row[IA], a_seg[IA], a_gx[IA];
  bool a_in[IA];
  const int a_c4 = (tid & 7) * 4;
#pragma unroll
  for (int e = 0; e < IA; ++e) {
    const int idx = e * T + tid;
    a_row[e] = idx >> 3;
    a_seg[e] = a_row[e] / SP;
    a_gx[e] = px0 - 1 + a_row[e] % SP;
    a_in[e] = idx < AR * 8 && a_gx[e] >= 0 && a_gx[e] < a.W;
    if (idx >= AR * 8) a_row[e] = -1;
  }
  const int64_t wplane = (int64_t)9 * cpt * a.Cout * kBK;
  const SplitScale ssc = split_scale<NP, F16>(a, wplane);
  int b_pl[IB], b_row[IB];
  bool b_ok[IB];
  const int b_part = (tid & 3) * 8;
#pragma unroll
  for (int e = 0; e < IB; ++e) {
    const int idx = e * T + tid;
    b_pl[e] = idx < NP * BN * 4 ? idx / (BN * 4) : -1;
    b_row[e] = (idx % (BN * 4)) >> 2;
    b_ok[e] = b_pl[e] >= 0 && n0 + b_row[e] < a.Cout;
  }
  float4 ra_reg[IA];
  bool ra_ok[IA];
  float4 tA = make_float4(1.f, 1.f, 1.f, 1.f), tB = make_float4(0.f, 0.f, 0.f, 0.f);
  bf16x8_t rb_reg[IB];
  auto load_a = [&](int cc, int ky) {
    if (NORM) {
      const float* t = a.ab + (int64_t)pn * 2 * a.Cin + cc * kBK + a_c4;
      tA = *reinterpret_cast<const float4*>(t);
      tB = *reinterpret_cast<const float4*>(t + a.Cin);
    }
    const float* base = a.in + (int64_t)pn * a.H * a.W * a.Cin + cc * kBK + a_c4;
#pragma unroll
    for (int e = 0; e < IA; ++e) {
      const int gy = py + a_seg[e] + ky - 1;
      const bool ok = a_in[e] && gy >= 0 && gy < a.H;
      ra_ok[e] = ok;
      if (GL == 2) {  // always issued (clamped address): the explicit vmcnt waits count on IA loads per wave
        const float4 v = *reinterpret_cast<const float4*>(base + ((int64_t)(ok ? gy : 0) * a.W + (ok ? a_gx[e] : 0)) * a.Cin);
        ra_reg[e] = v;  // (masked in store_a, a barrier later: a select here would let the compiler branch around the load)
      } else {
        ra_reg[e] = ok ? *reinterpret_cast<const float4*>(base + ((int64_t)gy * a.W + a_gx[e]) * a.Cin)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto load_b = [&](int cc, int tap) {
    const __bf16* wbase = a.wx + ((int64_t)tap * cpt + cc) * a.Cout * kBK;
#pragma unroll
    for (int e = 0; e < IB; ++e) {
      bf16x8_t z;
#pragma unroll
      for (int q = 0; q < 8; ++q) z[q] = (__bf16)0.f;
      const __bf16* src = wbase + (b_ok[e] ? b_pl[e] : 0) * wplane + (int64_t)(b_ok[e] ? n0 + b_row[e] : 0) * kBK + b_part;
      rb_reg[e] = b_ok[e] ? *reinterpret_cast<const bf16x8_t*>(src) : z;
    }
  };
  // LDS-DMA form: piece p = (plane p / 8, rows 16 (p % 8) .. + 15) is one wave-instruction of 64 x 16 bytes
  auto issue_b = [&](int cc, int tap, int buf) {
    const __bf16* wbase = a.wx + ((int64_t)tap * cpt + cc) * a.Cout * kBK;
    for (int p = wave; p < NP * 8; p += T / 64) {
      const int plane = p >> 3, row = (p & 7) * 16 + (lane >> 2);
      const int ch = (lane & 3) ^ ((row >> 2) & 3);
      int grow = n0 + row;
      grow = grow < a.Cout ? grow : a.Cout - 1;  // columns beyond Cout only feed accumulators that are never stored
      const __bf16* src = wbase + plane * wplane + (int64_t)grow * kBK + ch * 8;
#if defined(__HIP_DEVICE_COMPILE__)  // (the builtin exists in the device pass only; the host pass needs just the stub)
      __builtin_amdgcn_global_load_lds(src, &Bs[buf][plane][(p & 7) * 16][0], 16, 0, 0);
#else
      (void)src;
#endif
    }
  };
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  auto store_a = [&](int abuf = 0) {
#pragma unroll
    for (int e = 0; e < IA; ++e) {
      if (a_row[e] < 0) continue;
      bf16x4_t h4, m4, l4;
      const float4 v = ra_reg[e];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float x = q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
        if (GL == 2 && !NORM) x = ra_ok[e] ? x : 0.f;
        if (NORM) {
          const float ta = q == 0 ? tA.x : q == 1 ? tA.y : q == 2 ? tA.z : tA.w;
          const float tb = q == 0 ? tB.x : q == 1 ? tB.y : q == 2 ? tB.z : tB.w;
          const float t = fmaf(x, ta, tb);
          x = ra_ok[e] ? (a.act ? silu_f(t) : t) : 0.f;
        }
        __bf16 h, m, l;
        if (F16 && NP == 2) {
          split_half2(x * ssc.in, h, m);
          l = m;
        } else {
          split3(x, h, m, l);
          if (F16) h = half_bits(x);
        }
        h4[q] = h, m4[q] = m, l4[q] = l;
      }
      *reinterpret_cast<bf16x4_t*>(&As[abuf][0][a_row[e]][a_c4]) = h4;
      if (NP >= 2) *reinterpret_cast<bf16x4_t*>(&As[abuf][NP >= 2 ? 1 : 0][a_row[e]][a_c4]) = m4;
      if (NP == 3) *reinterpret_cast<bf16x4_t*>(&As[abuf][NP - 1][a_row[e]][a_c4]) = l4;
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int e = 0; e < IB; ++e)
      if (b_pl[e] >= 0) *reinterpret_cast<bf16x8_t*>(&Bs[buf][b_pl[e]][b_row[e]][b_part]) = rb_reg[e];
  };
  float16_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int arow[MI];  // staged row of this lane's pixel: pixel j of the tile sits at j + 2 * (j / SEGW) + 1, tap kx at - 1 + kx
#pragma unroll
  for (int i = 0; i < MI; ++i) arow[i] = wm + i * 32 + lr + 2 * ((wm + i * 32 + lr) / SEGW);
  auto compute = [&](int kx, int bbuf, int abuf = 0) {
#pragma unroll
    for (int ks = 0; ks < kBK / 16; ++ks) {
      const int ko = ks * 16 + 8 * lh;
      bf16x8_t af[NP][MI], bfr[NP][NI];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[p][i] = *reinterpret_cast<const bf16x8_t*>(&As[abuf][p][arow[i] + kx][ko]);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int brow = wn + j * 32 + lr;
          const int bko = GL ? (((ko >> 3) ^ ((brow >> 2) & 3)) << 3) : ko;
          bfr[p][j] = *reinterpret_cast<const bf16x8_t*>(&Bs[bbuf][p][brow][bko]);
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          float16_t t = acc[i][j];
          if (NP == 3) {  // smallest terms first: l h', h l', m m'
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NP - 1][i], bfr[0][j], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[NP - 1][j], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NP >= 2 ? 1 : 0][i], bfr[NP >= 2 ? 1 : 0][j], t, 0, 0, 0);
          }
          if (NP >= 2) {  // m h', h m'
            t = mfma16<F16>(af[NP >= 2 ? 1 : 0][i], bfr[0][j], t);
            t = mfma16<F16>(af[0][i], bfr[NP >= 2 ? 1 : 0][j], t);
          }
          t = mfma16<F16>(af[0][i], bfr[0][j], t);
          acc[i][j] = t;
        }
    }
  };
  // chunk c = (cc * 3 + ky) * 3 + kx
  if (GL == 2) {
    // s_waitcnt as the BUILTIN: the compiler's wait-count pass reads it and knows what is still in flight afterwards (after
    // an inline-asm wait it assumed everything was, and drained the fresh DMA with a vmcnt(0) at the first use of the row's
    // loads).  Immediate (gfx9): vmcnt = bits 3:0, expcnt = 6:4 (7 = no wait), lgkmcnt = 11:8 (0: LDS reads / writes done).
    // sched_barrier: the machine scheduler would otherwise lift store_a's register arithmetic across the barrier into the
    // earlier taps; the empty asm keeps the compiler's LDS / global accesses on their side of it.
#define FH_WAIT_BARRIER(n)                          \
  do {                                              \
    __builtin_amdgcn_sched_barrier(0);              \
    asm volatile("" ::: "memory");                  \
    __builtin_amdgcn_s_waitcnt(0x0070 | (n));       \
    __builtin_amdgcn_s_barrier();                   \
    asm volatile("" ::: "memory");                  \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)
    constexpr int kLoads = IA + (NORM ? 2 : 0);  // global loads of one activation row group per wave
    static_assert(kLoads + 1 == 4 || kLoads + 1 == 6, "FH_WAIT_BARRIER immediates below");
    load_a(0, 0);
    issue_b(0, 0, 0);
    issue_b(0, 1, 1);
    store_a(0);
    FH_WAIT_BARRIER(0);
    // one iteration = one (channel chunk, kernel row) = three taps in ring slots 0, 1, 2; straight-line so that the
    // compiler's own wait-count model sees the row's loads retired at store_a (with the taps as a runtime switch it
    // drained everything at the head of the next row)
    auto tap_args = [&](int c, int& cc, int& tap) { cc = c / 9, tap = ((c / 3) % 3) * 3 + c % 3; };
    const int nrg = nchunks / 3;
    int cc, tap;
    for (int rg = 0; rg + 1 < nrg; ++rg) {  // (the last row is peeled: nothing conditional inside, the model stays exact)
      const int c0 = 3 * rg, ab = rg & 1;
      // tap 0: [DMA(c0 + 1) from the tap before] loads of the next row, DMA(c0 + 2) -> only the oldest must have landed
      load_a((rg + 1) / 3, (rg + 1) % 3);
      tap_args(c0 + 2, cc, tap);
      issue_b(cc, tap, 2);
      compute(0, 0, ab);
      if (kLoads + 1 == 4)
        FH_WAIT_BARRIER(4);
      else
        FH_WAIT_BARRIER(6);
      // tap 1: DMA(c0 + 2) must have landed, this tap's DMA(c0 + 3) may fly (the loads, which the scheduler may have put
      // before the older DMA, are waited for too: they have had a whole tap)
      tap_args(c0 + 3, cc, tap);
      issue_b(cc, tap, 0);
      compute(1, 1, ab);
      FH_WAIT_BARRIER(1);
      // tap 2: products, then the next row's tile into the other activation buffer (last read in row rg - 1)
      tap_args(c0 + 4, cc, tap);
      issue_b(cc, tap, 1);
      compute(2, 2, ab);
      store_a(ab ^ 1);
      FH_WAIT_BARRIER(1);
    }
    {
      const int c0 = 3 * (nrg - 1), ab = (nrg - 1) & 1;
      tap_args(c0 + 2, cc, tap);
      issue_b(cc, tap, 2);
      compute(0, 0, ab);
      FH_WAIT_BARRIER(1);
      compute(1, 1, ab);
      FH_WAIT_BARRIER(0);
      compute(2, 2, ab);
    }
#undef FH_WAIT_BARRIER
  } else if (GL) {
    load_a(0, 0);
    issue_b(0, 0, 0);
    store_a();
    __syncthreads();  // (the barrier's fence also retires the DMA)
    for (int c = 0; c < nchunks; ++c) {
      const int kx = c % 3, nx = c + 1;
      const bool more = nx < nchunks, new_row = more && (nx % 3 == 0);
      if (more) {
        const int ncc = nx / 9, nky = (nx / 3) % 3, nkx = nx % 3;
        issue_b(ncc, nky * 3 + nkx, nx & 1);  // buffer nx & 1 was last read in iteration c - 1, before its barrier
        if (new_row) load_a(ncc, nky);
      }
      compute(kx, c & 1);
      __syncthreads();
      if (new_row) {
        store_a();
        __syncthreads();
      }
    }
  } else {
  load_a(0, 0);
  load_b(0, 0);
  store_a();
  store_b(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int kx = c % 3;
    const int nx = c + 1;
    const bool more = nx < nchunks;
    const bool new_row = more && (nx % 3 == 0);
    if (more) {
      const int ncc = nx / 9, nky = (nx / 3) % 3, nkx = nx % 3;
      load_b(ncc, nky * 3 + nkx);
      if (new_row) load_a(ncc, nky);
    }
    {
      compute(kx, 0);
      __syncthreads();
      if (more) {
        store_b(0);
        if (new_row) store_a();
        __syncthreads();
      }
    }
  }
  }
  const int64_t M = (int64_t)a.N * a.Ho * a.Wo;
  const bool gn = a.gn_partial != nullptr;
  GnAcc gsum[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    gsum[j].s0 = gsum[j].s1 = 0.f;
    const int co = n0 + wn + j * 32 + lr;
    if (co >= a.Cout) continue;
    const float bv = a.bias != nullptr ? a.bias[co] : 0.f;
    float tb[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (gn && a.gn_mode == 1) {
#pragma unroll
      for (int q = 0; q < 5; ++q) tb[q] = a.gn_tab[((int64_t)pn * 5 + q) * a.Cout + co];
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M) {
          float v = acc[i][j][r];
          if (F16 && NP == 2) v = v * ssc.inv_in * ssc.winv;
          v += bv;
          if (a.res != nullptr) v += a.res[row * a.Cout + co];
          a.out[row * a.Cout + co] = v;
          if (gn) gn_accum(a, gsum[j], v, row, co, tb);
        }
      }
    }
  }
  if (gn) gn_reduce<NI, BM / 64, WN, BN>(a, gsum, gred, pn, (rem / BM) * (int)gridDim.y + tile_n, n0);
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with a 1-D Winograd F(2,3) transform along W, fused into the implicit GEMM.
//   tile = 2 horizontally adjacent outputs (W even).  For kernel row ky the 4 input pixels d0..d3 at columns
//   2xt-1 .. 2xt+2 of input row y+ky-1 give  v = (d0-d2, d1+d2, d2-d1, d1-d3);  the weights arrive pre-transformed,
//   u = (w0, (w0+w1+w2)/2, (w0-w1+w2)/2, w2)  per (co, ky, ci);  four accumulator sets m_xi += V_xi U_xi^T over
//   K = 3*Cin;  y0 = m0+m1+m2,  y1 = m1-m2-m3.   4 multiplies per 2 outputs and kernel row instead of 6: 1.5x fewer
//   MFMAs than the direct kernel at the same exact-fp32 MFMA (error constants ~2x those of the direct sum).
// Data movement is what decides whether the 1.5x materialises, so the input is staged RAW: every pixel of the
// workgroup's tile range is loaded once per chunk (d0 / d3 of a tile are its neighbours' pixels) and the transform
// happens in registers when the MFMA fragments are read (4 b128 reads + 16 VALU per 4 fragments - the same read count
// as four pre-transformed matrices would need).  Workgroup = 8 waves (4 x 2): 256 tiles x 64 output channels; wave tile
// 64 tiles x 32 channels x 4 positions (128 accumulator VGPRs); K chunks of 16 channels, double-buffered LDS (row
// stride 20 floats: conflict-free b128 reads), register prefetch.  wu layout: [4 positions][Cout][3 ky][Cin].
// ------------------------------------------------------------------------------------------------
constexpr int kWK = 16;        // channels per chunk
constexpr int kWLd = kWK + 4;  // LDS row stride (floats)

template <int TB>  // tiles per workgroup: 256 (wave tile 64 x 32 x 4 positions) or 128 (32 x 32 x 4) for smaller grids
__global__ __launch_bounds__(512) void k_conv_wino(ConvArgs a) {
  constexpr int BN = 64;
  __shared__ __align__(16) float Ds[2][2][TB + 2][kWLd];  // raw pixels [buf][x parity][slot = tile + 1]; lane stride 20 floats
  __shared__ __align__(16) float Us[2][4][BN][kWLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int MI = TB / 128;
  const int wm = (wave >> 1) * (32 * MI), wn = (wave & 1) * 32;
  const int lr = lane & 31, lh = lane >> 5;
  const int Wt = a.W / 2;                                   // tiles per row
  const int64_t Mt = (int64_t)a.N * a.H * Wt;               // tiles in total
  const int64_t t0 = (int64_t)blockIdx.x * TB;
  const int n0 = blockIdx.y * BN;
  const int cpt = a.Cin / kWK;
  const int nchunks = 3 * cpt;

  // A staging: (TB + 2) * 2 pixel slots x 4 channel quads = 2064 float4 items; thread -> items tid + 512 e (e < 4), and
  // the first 16 threads one more.  Pixel coordinates of every item are fixed over the K loop.
  constexpr int NI = (TB + 2) * 8 / 512 + 1;
  int it_y[NI], it_slot[NI], it_q[NI];
  int64_t it_base[NI];  // ((n * H) * W + x) * Cin + 4 q ; row term added per chunk
  bool it_ok[NI];
#pragma unroll
  for (int e = 0; e < NI; ++e) {
    const int item = tid + 512 * e;
    const bool has = e < NI - 1 || tid < 16;
    const int ps = item >> 2;  // pixel slot 0 .. 2*(TB+2)-1
    it_q[e] = (item & 3) * 4;
    it_slot[e] = ps;
    const int64_t tg = t0 + (ps >> 1) - 1;
    it_ok[e] = has && tg >= 0 && tg < Mt;
    it_y[e] = 0;
    it_base[e] = 0;
    if (it_ok[e]) {
      const int n = (int)(tg / ((int64_t)a.H * Wt));
      const int rem = (int)(tg % ((int64_t)a.H * Wt));
      it_y[e] = rem / Wt;
      const int x = (rem % Wt) * 2 + (ps & 1);
      it_base[e] = (((int64_t)n * a.H) * a.W + x) * a.Cin + it_q[e];
    }
    if (!has) it_slot[e] = -1;
  }
  // B staging: 4 positions x 64 couts x 16 k = 1024 float4; thread -> two float4
  const int ub_pos[2] = {tid >> 8, 2 + (tid >> 8)};
  const int ub_co = (tid >> 2) & 63, ub_q = (tid & 3) * 4;
  const bool uvalid = n0 + ub_co < a.Cout;

  float4 rd[NI], ru[2];
  auto load_chunk = [&](int c) {
    const int ky = c / cpt, c0 = (c % cpt) * kWK;
#pragma unroll
    for (int e = 0; e < NI; ++e) {
      const int hy = it_y[e] + ky - 1;
      const bool ok = it_ok[e] && hy >= 0 && hy < a.H;
      rd[e] = ok ? *reinterpret_cast<const float4*>(a.in + it_base[e] + (int64_t)hy * a.W * a.Cin + c0)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float* up = a.w + (((int64_t)ub_pos[e] * a.Cout + (uvalid ? n0 + ub_co : 0)) * 3 + ky) * a.Cin + c0 + ub_q;
      ru[e] = uvalid ? *reinterpret_cast<const float4*>(up) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int e = 0; e < NI; ++e)
      if (it_slot[e] >= 0) *reinterpret_cast<float4*>(&Ds[buf][it_slot[e] & 1][it_slot[e] >> 1][it_q[e]]) = rd[e];
    *reinterpret_cast<float4*>(&Us[buf][ub_pos[0]][ub_co][ub_q]) = ru[0];
    *reinterpret_cast<float4*>(&Us[buf][ub_pos[1]][ub_co][ub_q]) = ru[1];
  };

  // per-lane tile rows of the two 32-tile halves of the wave tile: row-boundary flags (zero padding in x)
  float mfirst[MI], mlast[MI];  // 0 where the tile touches the left / right image border (zero padding in x)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int64_t tg = t0 + wm + i * 32 + lr;
    const int xt = (int)(tg % Wt);
    mfirst[i] = xt == 0 ? 0.f : 1.f;
    mlast[i] = xt == Wt - 1 ? 0.f : 1.f;
  }

  float16_t acc[4][MI];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][i][r] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
    for (int g = 0; g < kWK / 8; ++g) {
      const int ko = g * 8 + 4 * lh;
      float4 v[4][MI], bf[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) bf[p] = *reinterpret_cast<const float4*>(&Us[buf][p][wn + lr][ko]);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int sl = wm + i * 32 + lr + 1;
        const float4 d1 = *reinterpret_cast<const float4*>(&Ds[buf][0][sl][ko]);
        const float4 d2 = *reinterpret_cast<const float4*>(&Ds[buf][1][sl][ko]);
        const float4 d0 = *reinterpret_cast<const float4*>(&Ds[buf][1][sl - 1][ko]);
        const float4 d3 = *reinterpret_cast<const float4*>(&Ds[buf][0][sl + 1][ko]);
        const float f0 = mfirst[i], f3 = mlast[i];
        v[0][i] = make_float4(fmaf(d0.x, f0, -d2.x), fmaf(d0.y, f0, -d2.y), fmaf(d0.z, f0, -d2.z), fmaf(d0.w, f0, -d2.w));
        v[1][i] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
        v[2][i] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
        v[3][i] = make_float4(fmaf(-d3.x, f3, d1.x), fmaf(-d3.y, f3, d1.y), fmaf(-d3.z, f3, d1.z), fmaf(-d3.w, f3, d1.w));
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float bv = s4 == 0 ? bf[p].x : s4 == 1 ? bf[p].y : s4 == 2 ? bf[p].z : bf[p].w;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const float av = s4 == 0 ? v[p][i].x : s4 == 1 ? v[p][i].y : s4 == 2 ? v[p][i].z : v[p][i].w;
            acc[p][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[p][i], 0, 0, 0);
          }
        }
      }
    }
    if (c + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // output transform + epilogue.  C/D map: col (cout) = lane & 31, row (tile) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int co = n0 + wn + lr;
  if (co >= a.Cout) return;
  const float bv = a.bias != nullptr ? a.bias[co] : 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t t = t0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (t >= Mt) continue;
      const int n = (int)(t / ((int64_t)a.H * Wt));
      const int rem = (int)(t % ((int64_t)a.H * Wt));
      const int64_t pix = ((int64_t)n * a.H + rem / Wt) * a.W + (rem % Wt) * 2;
      float y0 = acc[0][i][r] + acc[1][i][r] + acc[2][i][r] + bv;
      float y1 = acc[1][i][r] - acc[2][i][r] - acc[3][i][r] + bv;
      if (a.res != nullptr) {
        y0 += a.res[pix * a.Cout + co];
        y1 += a.res[(pix + 1) * a.Cout + co];
      }
      a.out[pix * a.Cout + co] = y0;
      a.out[(pix + 1) * a.Cout + co] = y1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with a THIN output (Cout <= 8): the UNet's last convolution (128 -> 6 channels)
// and the input gradient of its first one (128 -> 3).  A GEMM tile would be > 90 % padding there (measured 5-10 TFLOP/s);
// this is a direct form instead: a workgroup owns a 16 x 16 pixel tile of one image, stages the 18 x 18 halo tile in
// chunks of 32 channels through LDS (row stride 36 floats: conflict-free b128 reads), a thread owns one pixel and all
// NO outputs, and the weights - uniform across the workgroup - arrive through the scalar path.  HBM-bound: the input is
// read ~1.27x (halo), the output is NO floats per pixel.
// ------------------------------------------------------------------------------------------------
template <int NO>
__global__ __launch_bounds__(256) void k_conv_thin(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, int H,
                                                   int W, int Cin, int Cout) {
  constexpr int TS = 16, HS = TS + 2, LD = 36;
  __shared__ __align__(16) float xs[HS * HS][LD];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const int64_t img = blockIdx.z;
  const float* src = in + img * H * W * Cin;
  float acc[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) acc[o] = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += 32) {
    for (int idx = tid; idx < HS * HS * 8; idx += 256) {
      const int pix = idx >> 3, c4 = (idx & 7) * 4;
      const int gy = y0 + pix / HS - 1, gx = x0 + pix % HS - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy >= 0 && gy < H && gx >= 0 && gx < W)
        v = *reinterpret_cast<const float4*>(src + ((int64_t)gy * W + gx) * Cin + c0 + c4);
      *reinterpret_cast<float4*>(&xs[pix][c4]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int p = (ty + tap / 3) * HS + tx + tap % 3;
#pragma unroll
      for (int c4 = 0; c4 < 32; c4 += 4) {
        const float4 v = *reinterpret_cast<const float4*>(&xs[p][c4]);
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          const float* wp = w + ((int64_t)(o < Cout ? o : 0) * 9 + tap) * Cin + c0 + c4;  // uniform: scalar loads
          acc[o] = fmaf(v.x, wp[0], fmaf(v.y, wp[1], fmaf(v.z, wp[2], fmaf(v.w, wp[3], acc[o]))));
        }
      }
    }
    __syncthreads();
  }
  const int oy = y0 + ty, ox = x0 + tx;
  if (oy >= H || ox >= W) return;
  float* dst = out + ((img * H + oy) * W + ox) * Cout;
#pragma unroll
  for (int o = 0; o < NO; ++o)
    if (o < Cout) dst[o] = acc[o] + (bias != nullptr ? bias[o] : 0.f);
}

// split-K epilogue: out = sum_z ws[z] + bias (+ res), fixed summation order
// max |x| of a tensor: non-negative floats order like their bit patterns, so the block maxima meet in one atomicMax on the
// uint image (order-independent, hence deterministic).  A NaN anywhere wins (its pattern is above +inf's).
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
  unsigned m = 0u;
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = max(m, __float_as_uint(fabsf(x[i])));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  __shared__ unsigned sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  // (same-address atomics serialise in L2: the blocks spread over FH_AMAX_SLOTS slots, the reader takes their max)
  if (threadIdx.x == 0) atomicMax(out + (blockIdx.x % FH_AMAX_SLOTS), max(max(sm[0], sm[1]), max(sm[2], sm[3])));
}

__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ ws, const float* __restrict__ bias,
                                                       const float* __restrict__ res, float* __restrict__ out,
                                                       int64_t total, int Cout, int ksplit) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float v = bias != nullptr ? bias[i % Cout] : 0.f;
    for (int z = 0; z < ksplit; ++z) v += ws[(int64_t)z * total + i];
    if (res != nullptr) v += res[i];
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Batched GEMM for attention.  C[b] = alpha * opA(A[b]) * opB(B[b]);  C is [M][N] row-major (ldc).
//   TA = 0: A stored [M][K] (lda)   TA = 1: A stored [K][M] (lda)
//   TB = 0: B stored [N][K] (ldb)   TB = 1: B stored [K][N] (ldb)      ("[N][K]" = K-contiguous, as the conv weights)
// batch index b = blockIdx.z -> (b / inner, b % inner) with two strides per operand so that (image, head) pairs
// address interleaved qkv channels without copies.  64x64 tile, one 32x32 MFMA tile per wave.
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K, lda, ldb, ldc, inner;
  int64_t sA0, sA1, sB0, sB1, sC0, sC1;
  float alpha;
};

template <int TA, int TB>
__global__ __launch_bounds__(256) void k_bgemm(GemmArgs g) {
  constexpr int BM = 64, BN = 64;
  __shared__ __align__(16) float As[BM][kLd];
  __shared__ __align__(16) float Bs[BN][kLd];
  const int b0 = blockIdx.z / g.inner, b1 = blockIdx.z % g.inner;
  const float* A = g.A + b0 * g.sA0 + b1 * g.sA1;
  const float* B = g.B + b0 * g.sB0 + b1 * g.sB1;
  float* C = g.C + b0 * g.sC0 + b1 * g.sC1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32, lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  float16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < g.K; k0 += kBK) {
    // stage 64 x 32 of each operand (2048 floats, 8 per thread)
    if (TA == 0) {
      const int row = tid >> 2, col = (tid & 3) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int gm = m0 + row, gk = k0 + col + e;
        As[row][col + e] = (gm < g.M && gk < g.K) ? A[(int64_t)gm * g.lda + gk] : 0.f;
      }
    } else {
      const int kk = tid >> 3, col = (tid & 7) * 8;  // 32 k-rows x 64 m
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int gm = m0 + col + e, gk = k0 + kk;
        As[col + e][kk] = (gm < g.M && gk < g.K) ? A[(int64_t)gk * g.lda + gm] : 0.f;
      }
    }
    if (TB == 0) {
      const int row = tid >> 2, col = (tid & 3) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int gn = n0 + row, gk = k0 + col + e;
        Bs[row][col + e] = (gn < g.N && gk < g.K) ? B[(int64_t)gn * g.ldb + gk] : 0.f;
      }
    } else {
      const int kk = tid >> 3, col = (tid & 7) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int gn = n0 + col + e, gk = k0 + kk;
        Bs[col + e][kk] = (gn < g.N && gk < g.K) ? B[(int64_t)gk * g.ldb + gn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int gq = 0; gq < kBK / 8; ++gq) {
      const float4 af = *reinterpret_cast<const float4*>(&As[wm + lr][gq * 8 + 4 * lh]);
      const float4 bf = *reinterpret_cast<const float4*>(&Bs[wn + lr][gq * 8 + 4 * lh]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = n0 + wn + lr;
  if (col < g.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < g.M) C[(int64_t)row * g.ldc + col] = g.alpha * acc[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm (32 groups, eps 1e-5) on NHWC.  x [N][P][C], P = H*W, cg = C / 32 channels per group.
//   k_gn_stats : one workgroup per (n, group); sums in double; writes mean / rstd (float) [N][32][2]
//   k_gn_apply : y = act( ((x - mean) * rstd * gamma + beta) * (1 + scale[n][c]) + shift[n][c] ),  act = SiLU or id
//   k_gn_bwd_* : dx given dy (through act, scale/shift, affine and the normalisation)
// ------------------------------------------------------------------------------------------------
// pixels per workgroup in the GroupNorm passes: 512, halved until the launch has >= 256 workgroups (min 32)
static inline int gn_chunk(int N, int P) {
  int c = 512;
  while (c > 32 && (int64_t)N * ((P + c - 1) / c) < 256) c >>= 1;
  return c;
}

// Shared skeleton of the two statistics passes.  Workgroup = (image n, chunk of kGnChunk pixels); a thread owns one
// float4 of channels (fixed over its pixel loop), accumulates two double sums per channel, folds them into 32 group
// slots in LDS, and the workgroup writes its 64 partial sums.  MODE 0: (sum x, sum x^2).  MODE 1: (sum g, sum g*xhat)
// with g = dL/d(xhat) rebuilt from dy through act / scale-shift / affine.
template <int MODE>
__global__ __launch_bounds__(256) void k_gn_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int ss_stride,
                                                    double* __restrict__ partial, int P, int C, int act, int nchunks,
                                                    int kGnChunk) {
  __shared__ double acc[64];
  const int n = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
  const int cg = C / 32, c4n = C / 4;
  if (threadIdx.x < 64) acc[threadIdx.x] = 0.0;
  __syncthreads();
  const int p0 = chunk * kGnChunk;
  const int p1 = p0 + kGnChunk < P ? p0 + kGnChunk : P;
  const int lanes_p = c4n >= 256 ? 1 : 256 / c4n;  // pixel lanes
  const int active = c4n >= 256 ? 256 : lanes_p * c4n;
  if ((int)threadIdx.x < active) {
    for (int c4 = threadIdx.x % (c4n < 256 ? c4n : 256); c4 < c4n; c4 += 256) {
      const int psub = c4n >= 256 ? 0 : threadIdx.x / c4n;
      double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
      float ga[4], be[4], sc[4], sh[4], mean[4], rstd[4];
      if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * c4 + e;
          ga[e] = gamma[c], be[e] = beta[c];
          sc[e] = scale != nullptr ? 1.f + scale[(int64_t)n * ss_stride + c] : 1.f;
          sh[e] = shift != nullptr ? shift[(int64_t)n * ss_stride + c] : 0.f;
          mean[e] = stats[((int64_t)n * 32 + c / cg) * 2], rstd[e] = stats[((int64_t)n * 32 + c / cg) * 2 + 1];
        }
      }
      // pixels in batches of 4 so that four (MODE 1: eight) independent 16-byte loads are in flight per thread
      for (int pb = p0 + psub; pb < p1; pb += 4 * lanes_p) {
        float4 xv[4], gv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = pb + u * lanes_p;
          ok[u] = p < p1;
          const int64_t idx = ((int64_t)n * P + (ok[u] ? p : p0)) * C + 4 * c4;
          xv[u] = *reinterpret_cast<const float4*>(x + idx);
          if (MODE == 1) gv[u] = *reinterpret_cast<const float4*>(dy + idx);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok[u]) continue;
          const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
          if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s0[e] += xs[e];
              s1[e] += (double)xs[e] * xs[e];
            }
          } else {
            const float gs[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xh = (xs[e] - mean[e]) * rstd[e];
              const float t = (xh * ga[e] + be[e]) * sc[e] + sh[e];
              float g = gs[e];
              if (act) {
                const float sg = 1.f / (1.f + __expf(-t));
                g *= sg * (1.f + t * (1.f - sg));
              }
              g *= sc[e] * ga[e];
              s0[e] += g;
              s1[e] += (double)g * xh;
            }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int grp = (4 * c4 + e) / cg;
        atomicAdd(&acc[2 * grp], s0[e]);
        atomicAdd(&acc[2 * grp + 1], s1[e]);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) partial[(int64_t)blockIdx.x * 64 + threadIdx.x] = acc[threadIdx.x];
}

// MODE 0 -> (mean, rstd);  MODE 1 -> (mean g, mean g*xhat).  One workgroup of 1024 threads per image: 16 row groups x 64
// slots, so a thread adds nchunks / 16 partials (the convolution epilogues leave up to 512 chunks per image); fixed order.
template <int MODE>
__global__ __launch_bounds__(1024) void k_gn_finalize(const double* __restrict__ partial, float* __restrict__ out,
                                                      int nchunks, double count) {
  __shared__ double red[16][64];
  const int n = blockIdx.x, slot = threadIdx.x & 63, rg = threadIdx.x >> 6;
  double s0 = 0.0, s1 = 0.0;
  int c = rg;
  for (; c + 16 < nchunks; c += 32) {
    s0 += partial[((int64_t)n * nchunks + c) * 64 + slot];
    s1 += partial[((int64_t)n * nchunks + c + 16) * 64 + slot];
  }
  if (c < nchunks) s0 += partial[((int64_t)n * nchunks + c) * 64 + slot];
  red[rg][slot] = s0 + s1;
  __syncthreads();
  if (threadIdx.x >= 64) return;
  double s = 0.0;
#pragma unroll
  for (int g = 0; g < 16; ++g) s += red[g][slot];
  const int grp = slot >> 1, which = slot & 1;
  const double other = __shfl_xor(s, 1, 64);
  if (MODE == 1) {
    out[((int64_t)n * 32 + grp) * 2 + which] = (float)(s / count);
  } else {
    const double sum = which ? other : s, sq = which ? s : other;
    const double mean = sum / count;
    double var = sq / count - mean * mean;
    var = var < 0 ? 0 : var;
    out[((int64_t)n * 32 + grp) * 2 + which] = which ? (float)(1.0 / sqrt(var + 1e-5)) : (float)mean;
  }
}


// Streaming passes.  Workgroup = (image n, chunk of kGnChunk pixels); a thread owns one float4 of channels for its
// whole pixel loop, so everything that depends on (n, channel) only - mean, rstd, gamma, beta, scale, shift, the
// backward sums - is folded into per-thread constants once and the loop is load / 4 FMAs (+ SiLU) / store.
//   forward :  y  = act(x * A + Bc),   A = rstd*gamma*(1+scale),  Bc = (beta - mean*rstd*gamma)*(1+scale) + shift
//   backward:  dx (+)= rstd * (g - a - xhat * b),  g = dy * act'(t) * (1+scale) * gamma,  (a, b) = sums
// Backward extras (null / 0 = off): `add2` - a second addend with x's layout (the gradient of a skip tensor that the
// reference adds at the skip's push point, folded into the pass that produces the other addend); `out2` / `csplit` - the
// result is written as TWO tensors, channels [0, csplit) to out ([.., csplit]) and [csplit, C) to out2 ([.., C - csplit]):
// the torch.cat of the decoder input splits its gradient without a copy pass (csplit % 4 == 0).
typedef float f4v_t __attribute__((ext_vector_type(4)));
// read-once operands of the large backward passes (gradients of tensors beyond the Infinity Cache): streaming loads
__device__ __forceinline__ float4 ld_stream(const float* p, int nt) {
  if (nt) {
    const f4v_t v = __builtin_nontemporal_load(reinterpret_cast<const f4v_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const float4*>(p);
}

template <int BWD, bool AMAX = false>  // AMAX: also fold max |out| (, |out2|) into `amax` (half-split mode); its own instance so that the default pass keeps its registers
__global__ __launch_bounds__(256) void k_gn_stream(const float* __restrict__ x, const float* __restrict__ dy,
                                                   const float* __restrict__ stats, const float* __restrict__ sums,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   int ss_stride, float* __restrict__ out, int P, int C, int act,
                                                   int accumulate, int nchunks, int kGnChunk,
                                                   const float* __restrict__ acc_src = nullptr,
                                                   const float* __restrict__ add2 = nullptr, float* __restrict__ out2 = nullptr,
                                                   int csplit = 0, unsigned* __restrict__ amax = nullptr, int nt = 0) {
  // amax (backward, optional): [2][FH_AMAX_SLOTS] uint images of max |out|, max |out2|, folded in with atomicMax (the caller zeroes them):
  // the magnitude the half-split input-gradient convolution that consumes the tensor scales it by (fh_absmax_f32's result)
  unsigned mx0 = 0u, mx1 = 0u;
  const int n = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
  const int cg = C / 32, c4n = C / 4;
  const int p0 = chunk * kGnChunk;
  const int p1 = p0 + kGnChunk < P ? p0 + kGnChunk : P;
  const int lanes_p = c4n >= 256 ? 1 : 256 / c4n;
  const int active = c4n >= 256 ? 256 : lanes_p * c4n;
  const bool live = (int)threadIdx.x < active;
  if (!live && !(BWD && AMAX)) return;
  for (int c4 = threadIdx.x % (c4n < 256 ? c4n : 256); live && c4 < c4n; c4 += 256) {
    const int psub = c4n >= 256 ? 0 : threadIdx.x / c4n;
    float A[4], Bc[4], mean[4], rstd[4], gsc[4], sa[4], sb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * c4 + e;
      const int64_t si = ((int64_t)n * 32 + c / cg) * 2;
      mean[e] = stats[si], rstd[e] = stats[si + 1];
      const float sc = scale != nullptr ? 1.f + scale[(int64_t)n * ss_stride + c] : 1.f;
      const float sh = shift != nullptr ? shift[(int64_t)n * ss_stride + c] : 0.f;
      const float rg = rstd[e] * gamma[c];
      A[e] = rg * sc;
      Bc[e] = (beta[c] - mean[e] * rg) * sc + sh;
      gsc[e] = sc * gamma[c];
      if (BWD) sa[e] = sums[si], sb[e] = sums[si + 1];
    }
    for (int pb = p0 + psub; pb < p1; pb += 4 * lanes_p) {
      float4 xv[4], gv[4], ov[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = pb + u * lanes_p;
        ok[u] = p < p1;
        const int64_t idx = ((int64_t)n * P + (ok[u] ? p : p0)) * C + 4 * c4;
        xv[u] = *reinterpret_cast<const float4*>(x + idx);
        if (BWD) {
          gv[u] = ld_stream(dy + idx, nt);
          if (accumulate) ov[u] = ld_stream((acc_src != nullptr ? acc_src : out) + idx, nt);
          if (add2 != nullptr) {
            const float4 t2 = ld_stream(add2 + idx, nt);
            if (accumulate)
              ov[u].x += t2.x, ov[u].y += t2.y, ov[u].z += t2.z, ov[u].w += t2.w;
            else
              ov[u] = t2;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        const int64_t idx = ((int64_t)n * P + pb + u * lanes_p) * C + 4 * c4;
        const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
        float o[4];
        if (!BWD) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = fmaf(xs[e], A[e], Bc[e]);
            o[e] = act ? silu_f(t) : t;
          }
        } else {
          const float gs[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
          const float os[4] = {ov[u].x, ov[u].y, ov[u].z, ov[u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = fmaf(xs[e], A[e], Bc[e]);
            const float xh = (xs[e] - mean[e]) * rstd[e];
            float g = gs[e];
            if (act) {
              const float sg = 1.f / (1.f + __expf(-t));
              g *= sg * (1.f + t * (1.f - sg));
            }
            g *= gsc[e];
            o[e] = ((accumulate || add2 != nullptr) ? os[e] : 0.f) + rstd[e] * (g - sa[e] - xh * sb[e]);
          }
        }
        if (BWD && AMAX) {
          const unsigned m4 = max(max(__float_as_uint(fabsf(o[0])), __float_as_uint(fabsf(o[1]))),
                                  max(__float_as_uint(fabsf(o[2])), __float_as_uint(fabsf(o[3]))));
          if (out2 != nullptr && 4 * c4 >= csplit)
            mx1 = max(mx1, m4);
          else
            mx0 = max(mx0, m4);
        }
        if (BWD && out2 != nullptr) {
          const int64_t pix = (int64_t)n * P + pb + u * lanes_p;
          float* dst = 4 * c4 < csplit ? out + pix * csplit + 4 * c4 : out2 + pix * (C - csplit) + (4 * c4 - csplit);
          *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        } else
        *reinterpret_cast<float4*>(out + idx) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  if (BWD && AMAX) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mx0 = max(mx0, (unsigned)__shfl_xor((int)mx0, o, 64));
      mx1 = max(mx1, (unsigned)__shfl_xor((int)mx1, o, 64));
    }
    __shared__ unsigned sm[8];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx0, sm[4 + (threadIdx.x >> 6)] = mx1;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx0 = max(max(sm[0], sm[1]), max(sm[2], sm[3])), mx1 = max(max(sm[4], sm[5]), max(sm[6], sm[7]));
      const int slot = blockIdx.x % FH_AMAX_SLOTS;
      if (mx0) atomicMax(amax + slot, mx0);
      if (mx1) atomicMax(amax + FH_AMAX_SLOTS + slot, mx1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Row softmax (rows of length T <= 4096), forward in place and backward  dS = P .* (dP - rowsum(dP .* P))
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_softmax_rows(float* __restrict__ s, int T) {
  __shared__ float redf[4];
  __shared__ double redd[4];
  float* row = s + (int64_t)blockIdx.x * T;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < T; i += 256) mx = fmaxf(mx, row[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  double sum = 0.0;
  for (int i = threadIdx.x; i < T; i += 256) {
    const float e = __expf(row[i] - mx);
    row[i] = e;
    sum += e;
  }
  sum = fh::block_sum_256(sum, redd);
  const float inv = (float)(1.0 / sum);
  for (int i = threadIdx.x; i < T; i += 256) row[i] *= inv;
}

__global__ __launch_bounds__(256) void k_softmax_bwd_rows(const float* __restrict__ p, float* __restrict__ dp, int T) {
  __shared__ double redd[4];
  const float* pr = p + (int64_t)blockIdx.x * T;
  float* dr = dp + (int64_t)blockIdx.x * T;
  double dot = 0.0;
  for (int i = threadIdx.x; i < T; i += 256) dot += (double)pr[i] * dr[i];
  dot = fh::block_sum_256(dot, redd);
  const float d = (float)dot;
  for (int i = threadIdx.x; i < T; i += 256) dr[i] = pr[i] * (dr[i] - d);
}

// ------------------------------------------------------------------------------------------------
// Resampling, concat and layout changes (NHWC, float4 along C)
// ------------------------------------------------------------------------------------------------
// mode 0: 2x2 average pool  [N][H][W][C] -> [N][H/2][W/2][C]        mode 1: its adjoint (grad / 4 to 4 pixels)
// mode 2: nearest 2x upsample [N][H][W][C] -> [N][2H][2W][C]        mode 3: its adjoint (sum of 4 grads)
// mode 4: zero insertion small -> big (out[2h][2w] = in[h][w], 0 elsewhere): the cotangent of a stride-2 convolution before
//         its stride-1 input-gradient pass (Downsample with conv_resample, openai_unet.py:131)
__global__ __launch_bounds__(256) void k_resample(const float* __restrict__ in, float* __restrict__ out, int N, int H,
                                                  int W, int C, int mode) {
  // H, W are always the SMALL side's dimensions
  const bool to_small = (mode == 0 || mode == 3);
  const int C4 = C / 4;
  const int64_t total = to_small ? (int64_t)N * H * W * C4 : (int64_t)N * 4 * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    if (to_small) {
      const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
      const float4* src = reinterpret_cast<const float4*>(in) + (((int64_t)n * 2 * H + 2 * h) * 2 * W + 2 * w) * C4 + c4;
      const float4 a = src[0], b = src[C4], c = src[(int64_t)2 * W * C4], d = src[(int64_t)2 * W * C4 + C4];
      const float f = mode == 0 ? 0.25f : 1.f;
      reinterpret_cast<float4*>(out)[i] = make_float4(f * (a.x + b.x + c.x + d.x), f * (a.y + b.y + c.y + d.y),
                                                      f * (a.z + b.z + c.z + d.z), f * (a.w + b.w + c.w + d.w));
    } else {
      const int w = (int)(p % (2 * W)), h = (int)((p / (2 * W)) % (2 * H)), n = (int)(p / ((int64_t)4 * W * H));
      float4 v = reinterpret_cast<const float4*>(in)[(((int64_t)n * H + h / 2) * W + w / 2) * C4 + c4];
      if (mode == 1) v = make_float4(0.25f * v.x, 0.25f * v.y, 0.25f * v.z, 0.25f * v.w);
      if (mode == 4 && ((h | w) & 1)) v = make_float4(0.f, 0.f, 0.f, 0.f);  // zero insertion: only (even, even) carries a value
      reinterpret_cast<float4*>(out)[i] = v;
    }
  }
}

// out[p][0:Ca] = a[p], out[p][Ca:Ca+Cb] = b[p]   (split = 1: the reverse, out -> a, b)
__global__ __launch_bounds__(256) void k_concat(float* __restrict__ a, float* __restrict__ b, float* __restrict__ out,
                                                int64_t P, int Ca, int Cb, int split) {
  const int C4 = (Ca + Cb) / 4, Ca4 = Ca / 4, Cb4 = Cb / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P * C4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int64_t p = i / C4;
    float4* o = reinterpret_cast<float4*>(out) + i;
    float4* s = c4 < Ca4 ? reinterpret_cast<float4*>(a) + p * Ca4 + c4 : reinterpret_cast<float4*>(b) + p * Cb4 + (c4 - Ca4);
    if (split) *s = *o; else *o = *s;
  }
}

// NCHW [N][C][P] <-> NHWC [N][P][Cp] (Cp >= C, extra channels zero / ignored)
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out, int N,
                                                      int C, int64_t P, int Cp) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)N * P * Cp; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Cp);
    const int64_t p = (i / Cp) % P, n = i / (Cp * P);
    out[i] = c < C ? in[((int64_t)n * C + c) * P + p] : 0.f;
  }
}
__global__ __launch_bounds__(256) void k_nhwc_to_nchw(const float* __restrict__ in, float* __restrict__ out, int N,
                                                      int C, int64_t P, int Cp) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)N * C * P; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i % P;
    const int c = (int)((i / P) % C);
    const int64_t n = i / (P * C);
    out[i] = in[((int64_t)n * P + p) * Cp + c];
  }
}

__global__ __launch_bounds__(256) void k_add_f32(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

inline unsigned grid_for(int64_t work_items, int per_block = 256, int cap = 4096) {
  int64_t b = (work_items + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

// Precision of the bf16-MFMA convolution kernels: 3 = exact 3-way split (fp32 accuracy, default), 1 = plain bf16 compute
// (operands rounded to bf16, one product, fp32 accumulation) - the reduced-precision UNet mode, the counterpart of the
// reference's use_fp16 torso (training/openai_fp16_util.py:15-32).  Process-wide: set before a forward / VJP.
// Mode 4 (half-split, 32): two half-precision planes per operand, three products on the f16 MFMA (see split_scale above):
// the host picks it per launch, so the switch is per host thread (the lock-step groups of a process launch concurrently).
static thread_local int g_conv_np = 3;
#define X6_DISPATCH(K1, K2, K3, K16, K32, ...)        \
  do {                                               \
    if (g_conv_np == 1)                              \
      hipLaunchKernelGGL(K1, __VA_ARGS__);           \
    else if (g_conv_np == 2)                         \
      hipLaunchKernelGGL(K2, __VA_ARGS__);           \
    else if (g_conv_np == 16)                        \
      hipLaunchKernelGGL(K16, __VA_ARGS__);          \
    else if (g_conv_np == 32)                        \
      hipLaunchKernelGGL(K32, __VA_ARGS__);          \
    else                                             \
      hipLaunchKernelGGL(K3, __VA_ARGS__);           \
  } while (0)

int fh_unet_set_precision(int mode) {
  if (mode < 0 || mode > 4) return FH_EINVAL;
  // 16: one half-precision plane on the f16 MFMA; 32: two half-precision planes, three products
  g_conv_np = mode == 0 ? 3 : (mode == 1 ? 1 : (mode == 2 ? 2 : (mode == 3 ? 16 : 32)));
  return 0;
}

int fh_absmax_f32(const float* x, int64_t n, float* out, void* stream) {
  if (!out || n < 0 || (n > 0 && !x) || ((uintptr_t)x & 15)) return FH_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  FH_CHECK(hipMemsetAsync(out, 0, FH_AMAX_SLOTS * sizeof(float), st));
  if (n > 0) {
    const int64_t want = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(k_absmax, dim3((unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want))), dim3(256), 0, st, x, n,
                       reinterpret_cast<unsigned*>(out));
  }
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_conv2d_splitk(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW) {
  // K-split factor for layers whose output grid cannot fill the chip (8x8 ... 32x32 at small batch)
  const int64_t M = (int64_t)N * Ho * Wo;
  const int64_t blocks = ((M + 63) / 64) * ((Cout + 63) / 64);
  const int nchunks = KH * KW * (Cin / kBK);
  int z = 1;
  static const int target = getenv("FH_SPLITK_TARGET") ? atoi(getenv("FH_SPLITK_TARGET")) : 2048;  // 64 x 64-tile blocks x z wanted (measured: 1024 -> 2048 +9 % on the 16^2 / 32^2 grids, 4096 no better)
  static const int zmax = getenv("FH_SPLITK_MAX") ? atoi(getenv("FH_SPLITK_MAX")) : 8;
  while (blocks * z < target && z < zmax && nchunks / (2 * z) >= 6) z *= 2;
  return z;
}

int fh_conv2d_nhwc(const float* in, const float* w, const float* bias, const float* res, float* out, float* ws,
                   int ksplit, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride,
                   void* stream) {
  if (!in || !w || !out || N < 1 || H < 1 || W < 1 || Cin < kBK || Cin % kBK != 0 || Cout < 1 || stride < 1)
    return FH_EINVAL;
  if (ksplit < 1 || ksplit > 32 || (ksplit > 1 && !ws)) return FH_EINVAL;
  ConvArgs a;
  a.in = in, a.w = w, a.bias = bias, a.res = res, a.out = out;
  a.N = N, a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout, a.KH = KH, a.KW = KW, a.pad = pad, a.stride = stride;
  a.Ho = (H + 2 * pad - KH) / stride + 1;
  a.Wo = (W + 2 * pad - KW) / stride + 1;
  a.ksplit = ksplit, a.ws = ws;
  const int64_t M = (int64_t)N * a.Ho * a.Wo;
  hipStream_t st = (hipStream_t)stream;
  const unsigned Z = (unsigned)ksplit;
  // tile choice: 128x128 when it still yields >= 1.5 workgroups per CU, else shrink M, then N
  const int64_t b128 = ((M + 127) / 128) * ((Cout + 127) / 128);
  if (ksplit == 1 && Cout > 64 && b128 >= 384) {
    hipLaunchKernelGGL((k_conv_igemm<2, 2>), dim3((unsigned)((M + 127) / 128), (Cout + 127) / 128, 1), dim3(256), 0, st, a);
  } else if (ksplit == 1 && Cout > 64 && ((M + 63) / 64) * ((Cout + 127) / 128) >= 256) {
    hipLaunchKernelGGL((k_conv_igemm<1, 2>), dim3((unsigned)((M + 63) / 64), (Cout + 127) / 128, 1), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((k_conv_igemm<1, 1>), dim3((unsigned)((M + 63) / 64), (Cout + 63) / 64, Z), dim3(256), 0, st, a);
  }
  if (ksplit > 1) {
    const int64_t total = M * Cout;
    hipLaunchKernelGGL(k_splitk_reduce, dim3(grid_for(total)), dim3(256), 0, st, (const float*)ws, bias, res, out, total,
                       Cout, ksplit);
  }
  FH_LAUNCH_CHECK();
  return 0;
}

// tile height (pixels) of the launch fh_conv2d_x6_nhwc takes for this layer, 0 when the group-sum epilogue does not apply
// (split-K, the 64-column tile, output tiles that straddle two images, channels not a multiple of 32)
static int x6_gn_tile(int ksplit, int64_t M, int HoWo, int Ho, int Wo, int Cout, int KH, int KW, int pad, int stride) {
  if (ksplit != 1 || Cout <= 64 || Cout % 32 != 0) return 0;
  const int64_t b128 = ((M + 127) / 128) * ((Cout + 127) / 128);
  int bm = 0;
  if (b128 >= 384) {
    const bool r3 = !getenv("FH_X6_NOREUSE") && KH == 3 && KW == 3 && pad == 1 && stride == 1 && M % 128 == 0;
    const bool big = r3 && !getenv("FH_X6_NOBIG") && M % 256 == 0 && (M / 256) * ((Cout + 127) / 128) >= 256;
    if (big && (Wo % 256 == 0 || (Wo == 128 && Ho % 2 == 0) || (Wo == 64 && Ho % 4 == 0)))
      bm = 256;
    else
      bm = 128;
  } else if (((M + 63) / 64) * ((Cout + 127) / 128) >= 256) {
    bm = 64;
  }
  return (bm > 0 && HoWo % bm == 0) ? bm : 0;
}

static int conv2d_x6_impl(const float* in, const void* wx, const float* bias, const float* res, float* out, float* ws,
                          int ksplit, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride,
                          const fh_gn_epilogue* epi, void* stream);

int fh_conv2d_x6_nhwc(const float* in, const void* wx, const float* bias, const float* res, float* out, float* ws,
                      int ksplit, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride,
                      void* stream) {
  return conv2d_x6_impl(in, wx, bias, res, out, ws, ksplit, N, H, W, Cin, Cout, KH, KW, pad, stride, nullptr, stream);
}

int fh_conv2d_x6_nhwc_gn(const float* in, const void* wx, const float* bias, const float* res, float* out, float* ws,
                         int ksplit, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride,
                         const fh_gn_epilogue* epi, void* stream) {
  return conv2d_x6_impl(in, wx, bias, res, out, ws, ksplit, N, H, W, Cin, Cout, KH, KW, pad, stride, epi, stream);
}

int fh_conv2d_x6_gn_chunks(int ksplit, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride) {
  if (N < 1 || H < 1 || W < 1 || Cout < 1 || stride < 1) return 0;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const int bm = x6_gn_tile(ksplit, (int64_t)N * Ho * Wo, Ho * Wo, Ho, Wo, Cout, KH, KW, pad, stride);
  return bm ? (Ho * Wo / bm) * ((Cout + 127) / 128) : 0;
}

static int set_gn(ConvArgsX& a, const fh_gn_epilogue* epi, int chunks) {
  a.gn_partial = nullptr, a.gn_x = nullptr, a.gn_tab = nullptr, a.gn_mode = a.gn_act = a.gn_chunks = 0;
  a.in_amax = epi != nullptr ? epi->in_amax : nullptr, a.in_scale = 1.f;
  if (g_conv_np == 32 && a.ab == nullptr && a.in_amax == nullptr) return FH_EINVAL;  // half-split of a raw tensor needs its magnitude
  if (a.ab != nullptr) a.in_amax = nullptr, a.in_scale = 16.f;  // GroupNorm(+SiLU) output: O(1) by construction
  if (epi == nullptr || epi->partial == nullptr) return 0;
  if (chunks <= 0 || epi->mode < 0 || epi->mode > 1 || (epi->mode == 1 && (!epi->x || !epi->tab))) return FH_EINVAL;
  a.gn_partial = epi->partial, a.gn_x = epi->x, a.gn_tab = epi->tab;
  a.gn_mode = epi->mode, a.gn_act = epi->act, a.gn_chunks = chunks;
  return 0;
}

static int conv2d_x6_impl(const float* in, const void* wx, const float* bias, const float* res, float* out, float* ws,
                          int ksplit, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride,
                          const fh_gn_epilogue* epi, void* stream) {
  if (!in || !wx || !out || N < 1 || H < 1 || W < 1 || Cin < kBK || Cin % kBK != 0 || Cout < 1 || stride < 1)
    return FH_EINVAL;
  if (ksplit < 1 || ksplit > 32 || (ksplit > 1 && !ws)) return FH_EINVAL;
  ConvArgsX a;
  a.ab = nullptr, a.act = 0;
  {
    const int rc = set_gn(a, epi, fh_conv2d_x6_gn_chunks(ksplit, N, H, W, Cin, Cout, KH, KW, pad, stride));
    if (rc) return rc;
  }
  a.in = in, a.wx = (const __bf16*)wx, a.bias = bias, a.res = res, a.out = out;
  a.N = N, a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout, a.KH = KH, a.KW = KW, a.pad = pad, a.stride = stride;
  a.Ho = (H + 2 * pad - KH) / stride + 1;
  a.Wo = (W + 2 * pad - KW) / stride + 1;
  a.ksplit = ksplit, a.ws = ws;
  const int64_t M = (int64_t)N * a.Ho * a.Wo;
  hipStream_t st = (hipStream_t)stream;
  const unsigned Z = (unsigned)ksplit;
  // 128 x 128 tiles run as 8 waves (2 x 4, wave tile 64 x 32: 102 VGPRs, 4 waves per SIMD over two workgroups per CU)
  const int64_t b128 = ((M + 127) / 128) * ((Cout + 127) / 128);
  if (ksplit == 1 && Cout > 64 && b128 >= 384) {
    static const int no_reuse = getenv("FH_X6_NOREUSE") != nullptr;
    const bool r3 = !no_reuse && KH == 3 && KW == 3 && pad == 1 && stride == 1 && M % 128 == 0;
    static const int no_big = getenv("FH_X6_NOBIG") != nullptr;
    const bool big = r3 && !no_big && M % 256 == 0 && (M / 256) * ((Cout + 127) / 128) >= 256;
    const dim3 gbig((unsigned)(M / 256), (Cout + 127) / 128, 1);
    static const int glds = getenv("FH_X6_GLDS") ? atoi(getenv("FH_X6_GLDS")) : 1;  // weight tile by LDS-DMA (default; 0 = register staging, the A/B switch)
    static const int deep = getenv("FH_X6_DEEP") ? atoi(getenv("FH_X6_DEEP")) : 1;  // half-split mode: three-slot weight ring + double-buffered activations (0 = the two-slot pipeline, the A/B switch)
    if (big && glds && W % 256 == 0)
      X6_DISPATCH((k_conv_x6r<256, false, 256, 1, true>), (k_conv_x6r<256, false, 256, 2, true>), (k_conv_x6r<256, false, 256, 3, true>), (k_conv_x6r<256, false, 256, 1, true, true>), (deep ? k_conv_x6r<256, false, 256, 2, 2, true> : k_conv_x6r<256, false, 256, 2, 1, true>), gbig, dim3(1024), 0, st, a);
    else if (big && glds && W == 128 && H % 2 == 0)
      X6_DISPATCH((k_conv_x6r<128, false, 256, 1, true>), (k_conv_x6r<128, false, 256, 2, true>), (k_conv_x6r<128, false, 256, 3, true>), (k_conv_x6r<128, false, 256, 1, true, true>), (deep ? k_conv_x6r<128, false, 256, 2, 2, true> : k_conv_x6r<128, false, 256, 2, 1, true>), gbig, dim3(1024), 0, st, a);
    else if (big && glds && W == 64 && H % 4 == 0)
      X6_DISPATCH((k_conv_x6r<64, false, 256, 1, true>), (k_conv_x6r<64, false, 256, 2, true>), (k_conv_x6r<64, false, 256, 3, true>), (k_conv_x6r<64, false, 256, 1, true, true>), (deep ? k_conv_x6r<64, false, 256, 2, 2, true> : k_conv_x6r<64, false, 256, 2, 1, true>), gbig, dim3(1024), 0, st, a);
    else if (big && W % 256 == 0)
      X6_DISPATCH((k_conv_x6r<256, false, 256, 1>), (k_conv_x6r<256, false, 256, 2>), (k_conv_x6r<256, false, 256, 3>), (k_conv_x6r<256, false, 256, 1, false, true>), (k_conv_x6r<256, false, 256, 2, false, true>), gbig, dim3(1024), 0, st, a);
    else if (big && W == 128 && H % 2 == 0)
      X6_DISPATCH((k_conv_x6r<128, false, 256, 1>), (k_conv_x6r<128, false, 256, 2>), (k_conv_x6r<128, false, 256, 3>), (k_conv_x6r<128, false, 256, 1, false, true>), (k_conv_x6r<128, false, 256, 2, false, true>), gbig, dim3(1024), 0, st, a);
    else if (big && W == 64 && H % 4 == 0)
      X6_DISPATCH((k_conv_x6r<64, false, 256, 1>), (k_conv_x6r<64, false, 256, 2>), (k_conv_x6r<64, false, 256, 3>), (k_conv_x6r<64, false, 256, 1, false, true>), (k_conv_x6r<64, false, 256, 2, false, true>), gbig, dim3(1024), 0, st, a);
    else if (r3 && W % 128 == 0)
      X6_DISPATCH((k_conv_x6r<128, false, 128, 1>), (k_conv_x6r<128, false, 128, 2>), (k_conv_x6r<128, false, 128, 3>), (k_conv_x6r<128, false, 128, 1, false, true>), (k_conv_x6r<128, false, 128, 2, false, true>), dim3((unsigned)(M / 128), (Cout + 127) / 128, 1), dim3(512), 0, st, a);
    else if (r3 && W == 64 && H % 2 == 0)
      X6_DISPATCH((k_conv_x6r<64, false, 128, 1>), (k_conv_x6r<64, false, 128, 2>), (k_conv_x6r<64, false, 128, 3>), (k_conv_x6r<64, false, 128, 1, false, true>), (k_conv_x6r<64, false, 128, 2, false, true>), dim3((unsigned)(M / 128), (Cout + 127) / 128, 1), dim3(512), 0, st, a);
    else if (r3 && W == 32 && H % 4 == 0)
      X6_DISPATCH((k_conv_x6r<32, false, 128, 1>), (k_conv_x6r<32, false, 128, 2>), (k_conv_x6r<32, false, 128, 3>), (k_conv_x6r<32, false, 128, 1, false, true>), (k_conv_x6r<32, false, 128, 2, false, true>), dim3((unsigned)(M / 128), (Cout + 127) / 128, 1), dim3(512), 0, st, a);
    else
      X6_DISPATCH((k_conv_x6<2, 1, 2, 4, 4, 1>), (k_conv_x6<2, 1, 2, 4, 4, 2>), (k_conv_x6<2, 1, 2, 4, 4, 3>), (k_conv_x6<2, 1, 2, 4, 4, 1, true>), (k_conv_x6<2, 1, 2, 4, 4, 2, true>), dim3((unsigned)((M + 127) / 128), (Cout + 127) / 128, 1), dim3(512), 0, st, a);
  } else if (ksplit > 1 && Cout > 64 && b128 * ksplit >= 256 && !getenv("FH_X6_NOBIGSPLIT")) {
    // small grids: 128 x 128 tiles (21 flop per byte pulled from L2 instead of 12.8) once split-K still fills the chip
    X6_DISPATCH((k_conv_x6<2, 1, 2, 4, 4, 1>), (k_conv_x6<2, 1, 2, 4, 4, 2>), (k_conv_x6<2, 1, 2, 4, 4, 3>), (k_conv_x6<2, 1, 2, 4, 4, 1, true>), (k_conv_x6<2, 1, 2, 4, 4, 2, true>), dim3((unsigned)((M + 127) / 128), (Cout + 127) / 128, Z), dim3(512), 0, st, a);
  } else if (ksplit == 1 && Cout > 64 && ((M + 63) / 64) * ((Cout + 127) / 128) >= 256) {
    X6_DISPATCH((k_conv_x6<1, 2, 2, 2, 2, 1>), (k_conv_x6<1, 2, 2, 2, 2, 2>), (k_conv_x6<1, 2, 2, 2, 2, 3>), (k_conv_x6<1, 2, 2, 2, 2, 1, true>), (k_conv_x6<1, 2, 2, 2, 2, 2, true>), dim3((unsigned)((M + 63) / 64), (Cout + 127) / 128, 1), dim3(256), 0, st, a);
  } else {
    X6_DISPATCH((k_conv_x6<1, 1, 2, 2, 2, 1>), (k_conv_x6<1, 1, 2, 2, 2, 2>), (k_conv_x6<1, 1, 2, 2, 2, 3>), (k_conv_x6<1, 1, 2, 2, 2, 1, true>), (k_conv_x6<1, 1, 2, 2, 2, 2, true>), dim3((unsigned)((M + 63) / 64), (Cout + 63) / 64, Z), dim3(256), 0, st, a);
  }
  if (ksplit > 1) {
    const int64_t total = M * Cout;
    hipLaunchKernelGGL(k_splitk_reduce, dim3(grid_for(total)), dim3(256), 0, st, (const float*)ws, bias, res, out, total,
                       Cout, ksplit);
  }
  FH_LAUNCH_CHECK();
  return 0;
}

// A[n][c] = rstd * gamma * (1 + scale), B[n][c] = (beta - mean * rstd * gamma) * (1 + scale) + shift: exactly k_gn_stream's
// per-channel constants, as a table [N][2][C] for the fused-input convolution
__global__ __launch_bounds__(256) void k_gn_table(const float* __restrict__ stats, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, int ss_stride,
                                                  float* __restrict__ table, int C) {
  const int n = blockIdx.x, cg = C / 32;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int64_t si = ((int64_t)n * 32 + c / cg) * 2;
    const float mean = stats[si], rstd = stats[si + 1];
    const float sc = scale != nullptr ? 1.f + scale[(int64_t)n * ss_stride + c] : 1.f;
    const float sh = shift != nullptr ? shift[(int64_t)n * ss_stride + c] : 0.f;
    const float rg = rstd * gamma[c];
    table[((int64_t)n * 2 + 0) * C + c] = rg * sc;
    table[((int64_t)n * 2 + 1) * C + c] = (beta[c] - mean * rg) * sc + sh;
  }
}

int fh_groupnorm_table(const float* stats, const float* gamma, const float* beta, const float* scale, const float* shift,
                       int ss_stride, float* table, int N, int C, void* stream) {
  if (!stats || !gamma || !beta || !table || N < 1 || C % 32 != 0) return FH_EINVAL;
  hipLaunchKernelGGL(k_gn_table, dim3(N), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta, scale, shift, ss_stride,
                     table, C);
  FH_LAUNCH_CHECK();
  return 0;
}

// 1 when fh_conv2d_x6_norm_nhwc supports the layer (the row-reuse tiling must apply and fill the chip)
int fh_conv2d_x6_norm_supported(int N, int H, int W, int Cin, int Cout) {
  const int64_t M = (int64_t)N * H * W;
  if (Cin < kBK || Cin % kBK != 0 || Cout <= 64 || M % 128 != 0) return 0;
  if (((M + 127) / 128) * ((Cout + 127) / 128) < 384) return 0;
  return (W % 128 == 0) || (W == 64 && H % 2 == 0) || (W == 32 && H % 4 == 0);
}

static int conv2d_x6_norm_impl(const float* in, const float* ab_table, int act, const void* wx, const float* bias,
                               const float* res, float* out, int N, int H, int W, int Cin, int Cout, const fh_gn_epilogue* epi,
                               void* stream);

int fh_conv2d_x6_norm_nhwc(const float* in, const float* ab_table, int act, const void* wx, const float* bias,
                           const float* res, float* out, int N, int H, int W, int Cin, int Cout, void* stream) {
  return conv2d_x6_norm_impl(in, ab_table, act, wx, bias, res, out, N, H, W, Cin, Cout, nullptr, stream);
}

int fh_conv2d_x6_norm_nhwc_gn(const float* in, const float* ab_table, int act, const void* wx, const float* bias,
                              const float* res, float* out, int N, int H, int W, int Cin, int Cout, const fh_gn_epilogue* epi,
                              void* stream) {
  return conv2d_x6_norm_impl(in, ab_table, act, wx, bias, res, out, N, H, W, Cin, Cout, epi, stream);
}

static int conv2d_x6_norm_impl(const float* in, const float* ab_table, int act, const void* wx, const float* bias,
                               const float* res, float* out, int N, int H, int W, int Cin, int Cout, const fh_gn_epilogue* epi,
                               void* stream) {
  if (!in || !ab_table || !wx || !out || !fh_conv2d_x6_norm_supported(N, H, W, Cin, Cout)) return FH_EINVAL;
  ConvArgsX a;
  a.ab = ab_table, a.act = act;
  {
    // the fused-input kernel takes the same tile as the plain 3 x 3 / stride 1 / pad 1 launch of the layer
    const int rc = set_gn(a, epi, fh_conv2d_x6_gn_chunks(1, N, H, W, Cin, Cout, 3, 3, 1, 1));
    if (rc) return rc;
  }
  a.in = in, a.wx = (const __bf16*)wx, a.bias = bias, a.res = res, a.out = out;
  a.N = N, a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout, a.KH = 3, a.KW = 3, a.pad = 1, a.stride = 1;
  a.Ho = H, a.Wo = W, a.ksplit = 1, a.ws = nullptr;
  const int64_t M = (int64_t)N * H * W;
  const dim3 grid((unsigned)(M / 128), (Cout + 127) / 128, 1);
  hipStream_t st = (hipStream_t)stream;
  static const int no_big = getenv("FH_X6_NOBIG") != nullptr;
  const bool big = !no_big && M % 256 == 0 && (M / 256) * ((Cout + 127) / 128) >= 256;
  const dim3 gbig((unsigned)(M / 256), (Cout + 127) / 128, 1);
  static const int glds = getenv("FH_X6_GLDS") ? atoi(getenv("FH_X6_GLDS")) : 1;
  static const int deep = getenv("FH_X6_DEEP") ? atoi(getenv("FH_X6_DEEP")) : 1;
  if (big && glds && W % 256 == 0)
    X6_DISPATCH((k_conv_x6r<256, true, 256, 1, true>), (k_conv_x6r<256, true, 256, 2, true>), (k_conv_x6r<256, true, 256, 3, true>), (k_conv_x6r<256, true, 256, 1, true, true>), (deep ? k_conv_x6r<256, true, 256, 2, 2, true> : k_conv_x6r<256, true, 256, 2, 1, true>), gbig, dim3(1024), 0, st, a);
  else if (big && glds && W == 128 && H % 2 == 0)
    X6_DISPATCH((k_conv_x6r<128, true, 256, 1, true>), (k_conv_x6r<128, true, 256, 2, true>), (k_conv_x6r<128, true, 256, 3, true>), (k_conv_x6r<128, true, 256, 1, true, true>), (deep ? k_conv_x6r<128, true, 256, 2, 2, true> : k_conv_x6r<128, true, 256, 2, 1, true>), gbig, dim3(1024), 0, st, a);
  else if (big && glds && W == 64 && H % 4 == 0)
    X6_DISPATCH((k_conv_x6r<64, true, 256, 1, true>), (k_conv_x6r<64, true, 256, 2, true>), (k_conv_x6r<64, true, 256, 3, true>), (k_conv_x6r<64, true, 256, 1, true, true>), (deep ? k_conv_x6r<64, true, 256, 2, 2, true> : k_conv_x6r<64, true, 256, 2, 1, true>), gbig, dim3(1024), 0, st, a);
  else if (big && W % 256 == 0)
    X6_DISPATCH((k_conv_x6r<256, true, 256, 1>), (k_conv_x6r<256, true, 256, 2>), (k_conv_x6r<256, true, 256, 3>), (k_conv_x6r<256, true, 256, 1, false, true>), (k_conv_x6r<256, true, 256, 2, false, true>), gbig, dim3(1024), 0, st, a);
  else if (big && W == 128 && H % 2 == 0)
    X6_DISPATCH((k_conv_x6r<128, true, 256, 1>), (k_conv_x6r<128, true, 256, 2>), (k_conv_x6r<128, true, 256, 3>), (k_conv_x6r<128, true, 256, 1, false, true>), (k_conv_x6r<128, true, 256, 2, false, true>), gbig, dim3(1024), 0, st, a);
  else if (big && W == 64 && H % 4 == 0)
    X6_DISPATCH((k_conv_x6r<64, true, 256, 1>), (k_conv_x6r<64, true, 256, 2>), (k_conv_x6r<64, true, 256, 3>), (k_conv_x6r<64, true, 256, 1, false, true>), (k_conv_x6r<64, true, 256, 2, false, true>), gbig, dim3(1024), 0, st, a);
  else if (W % 128 == 0)
    X6_DISPATCH((k_conv_x6r<128, true, 128, 1>), (k_conv_x6r<128, true, 128, 2>), (k_conv_x6r<128, true, 128, 3>), (k_conv_x6r<128, true, 128, 1, false, true>), (k_conv_x6r<128, true, 128, 2, false, true>), grid, dim3(512), 0, st, a);
  else if (W == 64)
    X6_DISPATCH((k_conv_x6r<64, true, 128, 1>), (k_conv_x6r<64, true, 128, 2>), (k_conv_x6r<64, true, 128, 3>), (k_conv_x6r<64, true, 128, 1, false, true>), (k_conv_x6r<64, true, 128, 2, false, true>), grid, dim3(512), 0, st, a);
  else
    X6_DISPATCH((k_conv_x6r<32, true, 128, 1>), (k_conv_x6r<32, true, 128, 2>), (k_conv_x6r<32, true, 128, 3>), (k_conv_x6r<32, true, 128, 1, false, true>), (k_conv_x6r<32, true, 128, 2, false, true>), grid, dim3(512), 0, st, a);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_conv3x3_thin_nhwc(const float* in, const float* w, const float* bias, float* out, int N, int H, int W, int Cin,
                         int Cout, void* stream) {
  if (!in || !w || !out || N < 1 || N > 65535 || H < 1 || W < 1 || Cin < 32 || Cin % 32 != 0 || Cout < 1 || Cout > 8)
    return FH_EINVAL;
  const dim3 grid((W + 15) / 16, (H + 15) / 16, N);
  if (Cout <= 4)
    hipLaunchKernelGGL(k_conv_thin<4>, grid, dim3(256), 0, (hipStream_t)stream, in, w, bias, out, H, W, Cin, Cout);
  else
    hipLaunchKernelGGL(k_conv_thin<8>, grid, dim3(256), 0, (hipStream_t)stream, in, w, bias, out, H, W, Cin, Cout);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_conv3x3_wino_nhwc(const float* in, const float* wu, const float* bias, const float* res, float* out, int N,
                         int H, int W, int Cin, int Cout, void* stream) {
  if (!in || !wu || !out || N < 1 || H < 1 || W < 2 || (W & 1) || Cin < kWK || Cin % kWK != 0 || Cout < 1)
    return FH_EINVAL;
  ConvArgs a;
  a.in = in, a.w = wu, a.bias = bias, a.res = res, a.out = out;
  a.N = N, a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout, a.KH = 3, a.KW = 3, a.pad = 1, a.stride = 1;
  a.Ho = H, a.Wo = W, a.ksplit = 1, a.ws = nullptr;
  const int64_t Mt = (int64_t)N * H * (W / 2);
  const int nb = (Cout + 63) / 64;
  if (((Mt + 255) / 256) * nb >= 224)  // a full wave of 256-tile workgroups over the 256 CUs, else the finer tiling
    hipLaunchKernelGGL(k_conv_wino<256>, dim3((unsigned)((Mt + 255) / 256), nb, 1), dim3(512), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_conv_wino<128>, dim3((unsigned)((Mt + 127) / 128), nb, 1), dim3(512), 0, (hipStream_t)stream, a);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_bgemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                 int transB, int batch, int inner, int64_t sA0, int64_t sA1, int64_t sB0, int64_t sB1, int64_t sC0,
                 int64_t sC1, float alpha, void* stream) {
  if (!A || !B || !C || M < 1 || N < 1 || K < 1 || batch < 1 || inner < 1) return FH_EINVAL;
  GemmArgs g;
  g.A = A, g.B = B, g.C = C, g.M = M, g.N = N, g.K = K, g.lda = lda, g.ldb = ldb, g.ldc = ldc, g.inner = inner;
  g.sA0 = sA0, g.sA1 = sA1, g.sB0 = sB0, g.sB1 = sB1, g.sC0 = sC0, g.sC1 = sC1, g.alpha = alpha;
  dim3 grid((M + 63) / 64, (N + 63) / 64, batch);
  hipStream_t st = (hipStream_t)stream;
  if (!transA && !transB) hipLaunchKernelGGL((k_bgemm<0, 0>), grid, dim3(256), 0, st, g);
  else if (!transA && transB) hipLaunchKernelGGL((k_bgemm<0, 1>), grid, dim3(256), 0, st, g);
  else if (transA && !transB) hipLaunchKernelGGL((k_bgemm<1, 0>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((k_bgemm<1, 1>), grid, dim3(256), 0, st, g);
  FH_LAUNCH_CHECK();
  return 0;
}

int64_t fh_groupnorm_scratch_doubles(int N, int P) {
  const int chunk = gn_chunk(N, P);
  return (int64_t)N * ((P + chunk - 1) / chunk) * 64;
}

int fh_groupnorm_stats(const float* x, float* stats, double* scratch, int N, int P, int C, void* stream) {
  if (!x || !stats || !scratch || C % 32 != 0) return FH_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int kGnChunk = gn_chunk(N, P);
  const int nchunks = (P + kGnChunk - 1) / kGnChunk;
  double* part = scratch;
  hipLaunchKernelGGL(k_gn_partial<0>, dim3(N * nchunks), dim3(256), 0, st, x, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, 0, part, P, C, 0, nchunks, kGnChunk);
  hipLaunchKernelGGL(k_gn_finalize<0>, dim3(N), dim3(1024), 0, st, (const double*)part, stats, nchunks,
                     (double)P * (C / 32));
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_groupnorm_apply(const float* x, const float* stats, const float* gamma, const float* beta, const float* scale,
                       const float* shift, int ss_stride, float* y, int N, int P, int C, int act, void* stream) {
  if (!x || !stats || !gamma || !beta || !y || C % 32 != 0) return FH_EINVAL;
  const int kGnChunk = gn_chunk(N, P);
  const int nchunks = (P + kGnChunk - 1) / kGnChunk;
  hipLaunchKernelGGL(k_gn_stream<0>, dim3(N * nchunks), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr,
                     stats, (const float*)nullptr, gamma, beta, scale, shift, ss_stride, y, P, C, act, 0, nchunks,
                     kGnChunk);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_groupnorm_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                     const float* scale, const float* shift, int ss_stride, float* sums, double* scratch, float* dx, int N,
                     int P, int C, int act, int accumulate, void* stream) {
  if (!x || !dy || !stats || !gamma || !beta || !sums || !scratch || !dx || C % 32 != 0) return FH_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int kGnChunk = gn_chunk(N, P);
  const int nchunks = (P + kGnChunk - 1) / kGnChunk;
  double* part = scratch;
  hipLaunchKernelGGL(k_gn_partial<1>, dim3(N * nchunks), dim3(256), 0, st, x, dy, stats, gamma, beta, scale, shift,
                     ss_stride, part, P, C, act, nchunks, kGnChunk);
  hipLaunchKernelGGL(k_gn_finalize<1>, dim3(N), dim3(1024), 0, st, (const double*)part, sums, nchunks,
                     (double)P * (C / 32));
  hipLaunchKernelGGL(k_gn_stream<1>, dim3(N * nchunks), dim3(256), 0, st, x, dy, stats, (const float*)sums, gamma, beta,
                     scale, shift, ss_stride, dx, P, C, act, accumulate, nchunks, kGnChunk);
  FH_LAUNCH_CHECK();
  return 0;
}

// ---- pieces of the GroupNorm passes for the convolutions' group-sum epilogue (fh_gn_epilogue) -------------------------
// the block partials of an epilogue -> (mean, rstd) [mode 0] or the two backward sums [mode 1], [N][32][2] float
int fh_groupnorm_finalize(const double* partial, float* out, int N, int chunks, double count, int mode, void* stream) {
  if (!partial || !out || N < 1 || chunks < 1 || count <= 0 || mode < 0 || mode > 1) return FH_EINVAL;
  if (mode == 0)
    hipLaunchKernelGGL(k_gn_finalize<0>, dim3(N), dim3(1024), 0, (hipStream_t)stream, partial, out, chunks, count);
  else
    hipLaunchKernelGGL(k_gn_finalize<1>, dim3(N), dim3(1024), 0, (hipStream_t)stream, partial, out, chunks, count);
  FH_LAUNCH_CHECK();
  return 0;
}

// per-(image, channel) constants of a GroupNorm backward for the mode-1 epilogue: [N][5][C] = A, Bc (t = A x + Bc, as
// k_gn_stream), mean, rstd, gamma (1 + scale)
__global__ __launch_bounds__(256) void k_gn_bwd_table(const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int ss_stride,
                                                      float* __restrict__ table, int C) {
  const int n = blockIdx.x, cg = C / 32;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int64_t si = ((int64_t)n * 32 + c / cg) * 2;
    const float mean = stats[si], rstd = stats[si + 1];
    const float sc = scale != nullptr ? 1.f + scale[(int64_t)n * ss_stride + c] : 1.f;
    const float sh = shift != nullptr ? shift[(int64_t)n * ss_stride + c] : 0.f;
    const float rg = rstd * gamma[c];
    float* t = table + (int64_t)n * 5 * C + c;
    t[0] = rg * sc;
    t[C] = (beta[c] - mean * rg) * sc + sh;
    t[2 * C] = mean;
    t[3 * C] = rstd;
    t[4 * C] = sc * gamma[c];
  }
}

int fh_groupnorm_bwd_table(const float* stats, const float* gamma, const float* beta, const float* scale, const float* shift,
                           int ss_stride, float* table, int N, int C, void* stream) {
  if (!stats || !gamma || !beta || !table || N < 1 || C % 32 != 0) return FH_EINVAL;
  hipLaunchKernelGGL(k_gn_bwd_table, dim3(N), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta, scale, shift, ss_stride,
                     table, C);
  FH_LAUNCH_CHECK();
  return 0;
}

// the streaming pass of fh_groupnorm_bwd alone, with the two sums given (from fh_groupnorm_finalize mode 1)
int fh_groupnorm_bwd_apply(const float* x, const float* dy, const float* stats, const float* sums, const float* gamma,
                           const float* beta, const float* scale, const float* shift, int ss_stride, float* dx, int N, int P,
                           int C, int act, int accumulate, void* stream) {
  if (!x || !dy || !stats || !sums || !gamma || !beta || !dx || C % 32 != 0) return FH_EINVAL;
  const int kGnChunk = gn_chunk(N, P);
  const int nchunks = (P + kGnChunk - 1) / kGnChunk;
  hipLaunchKernelGGL(k_gn_stream<1>, dim3(N * nchunks), dim3(256), 0, (hipStream_t)stream, x, dy, stats, sums, gamma, beta,
                     scale, shift, ss_stride, dx, P, C, act, accumulate, nchunks, kGnChunk);
  FH_LAUNCH_CHECK();
  return 0;
}

// the two reductions of fh_groupnorm_bwd alone: sums [N][32][2] = (mean g, mean g xhat)
int fh_groupnorm_bwd_sums(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                          const float* scale, const float* shift, int ss_stride, float* sums, double* scratch, int N, int P,
                          int C, int act, void* stream) {
  if (!x || !dy || !stats || !gamma || !beta || !sums || !scratch || C % 32 != 0) return FH_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int kGnChunk = gn_chunk(N, P);
  const int nchunks = (P + kGnChunk - 1) / kGnChunk;
  hipLaunchKernelGGL(k_gn_partial<1>, dim3(N * nchunks), dim3(256), 0, st, x, dy, stats, gamma, beta, scale, shift,
                     ss_stride, scratch, P, C, act, nchunks, kGnChunk);
  hipLaunchKernelGGL(k_gn_finalize<1>, dim3(N), dim3(1024), 0, st, (const double*)scratch, sums, nchunks,
                     (double)P * (C / 32));
  FH_LAUNCH_CHECK();
  return 0;
}

// fh_groupnorm_bwd_apply with the two extras of k_gn_stream<1>: acc_src (addend read from another tensor than dx), add2 (a
// second addend) and the split destination (dx: [.., csplit], dx2: [.., C - csplit]); any of them may be null / 0
int fh_groupnorm_bwd_apply_ex(const float* x, const float* dy, const float* stats, const float* sums, const float* gamma,
                              const float* beta, const float* scale, const float* shift, int ss_stride, const float* acc_src,
                              const float* add2, float* dx, float* dx2, int csplit, int N, int P, int C, int act, float* amax2,
                              void* stream) {
  if (!x || !dy || !stats || !sums || !gamma || !beta || !dx || C % 32 != 0) return FH_EINVAL;
  if (dx2 != nullptr && (csplit <= 0 || csplit >= C || csplit % 4 != 0)) return FH_EINVAL;
  if (dx2 == nullptr && acc_src != nullptr && acc_src != dx) {
    // (without a split the addend may still come from elsewhere: dx is written, acc_src only read)
  }
  const int kGnChunk = gn_chunk(N, P);
  const int nchunks = (P + kGnChunk - 1) / kGnChunk;
  static const int nt_env = getenv("FH_GN_NT") ? atoi(getenv("FH_GN_NT")) : 1;  // streaming loads of the read-once gradients beyond the Infinity Cache (measured 233 -> 218 us at 8 x 256^2 x 128; 0 = off)
  const int nt = nt_env && (int64_t)N * P * C * 4 > ((int64_t)200 << 20);
  if (amax2 != nullptr)
    hipLaunchKernelGGL((k_gn_stream<1, true>), dim3(N * nchunks), dim3(256), 0, (hipStream_t)stream, x, dy, stats, sums, gamma,
                       beta, scale, shift, ss_stride, dx, P, C, act, acc_src != nullptr ? 1 : 0, nchunks, kGnChunk, acc_src, add2,
                       dx2, csplit, reinterpret_cast<unsigned*>(amax2), nt);
  else
    hipLaunchKernelGGL((k_gn_stream<1, false>), dim3(N * nchunks), dim3(256), 0, (hipStream_t)stream, x, dy, stats, sums, gamma,
                       beta, scale, shift, ss_stride, dx, P, C, act, acc_src != nullptr ? 1 : 0, nchunks, kGnChunk, acc_src, add2,
                       dx2, csplit, (unsigned*)nullptr, nt);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_softmax_rows(float* s, int64_t rows, int T, void* stream) {
  if (!s || rows < 1 || T < 1) return FH_EINVAL;
  hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, T);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_softmax_bwd_rows(const float* p, float* dp, int64_t rows, int T, void* stream) {
  if (!p || !dp || rows < 1 || T < 1) return FH_EINVAL;
  hipLaunchKernelGGL(k_softmax_bwd_rows, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, p, dp, T);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_resample2x(const float* in, float* out, int N, int Hs, int Ws, int C, int mode, void* stream) {
  if (!in || !out || C % 4 != 0 || mode < 0 || mode > 4) return FH_EINVAL;
  const int64_t items = (int64_t)N * Hs * Ws * (C / 4) * ((mode == 0 || mode == 3) ? 1 : 4);
  hipLaunchKernelGGL(k_resample, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, in, out, N, Hs, Ws, C, mode);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_concat_channels(float* a, float* b, float* out, int64_t P, int Ca, int Cb, int split, void* stream) {
  if (!a || !b || !out || Ca % 4 != 0 || Cb % 4 != 0) return FH_EINVAL;
  hipLaunchKernelGGL(k_concat, dim3(grid_for(P * ((Ca + Cb) / 4))), dim3(256), 0, (hipStream_t)stream, a, b, out, P, Ca,
                     Cb, split);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_layout_nchw_nhwc(const float* in, float* out, int N, int C, int64_t P, int Cp, int to_nhwc, void* stream) {
  if (!in || !out || Cp < C) return FH_EINVAL;
  if (to_nhwc)
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(grid_for((int64_t)N * P * Cp)), dim3(256), 0, (hipStream_t)stream, in, out, N,
                       C, P, Cp);
  else
    hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(grid_for((int64_t)N * P * C)), dim3(256), 0, (hipStream_t)stream, in, out, N,
                       C, P, Cp);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
  if (!a || !b || !out || n % 4 != 0) return FH_EINVAL;
  hipLaunchKernelGGL(k_add_f32, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4);
  FH_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
