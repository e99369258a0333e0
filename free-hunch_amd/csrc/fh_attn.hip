// Fused self-attention of the UNet's AttentionBlock (reference: training/openai_unet.py:296-305 `_forward`, :337-354
// QKVAttentionLegacy, :370-384 QKVAttention) for gfx950, fp32 throughout.
//
//   forward :  one kernel per (image, head, 128 queries):  S = Q K^T / sqrt(ch) -> online softmax -> . V
//   backward:  P is RECOMPUTED from (Q, K, log-sum-exp) - as the reference's checkpointed block does - so the T x T matrix
//              never exists in memory:  k_attn_bwd_dq (per query block: D = rowsum(dO . O), dQ) and k_attn_bwd_dkv (per key
//              block: dK, dV); no atomics, fixed summation order.
//
// Matrix products run on v_mfma_f32_32x32x2_f32 (fp32 operands: the softmax weights and their gradients are produced in
// fp32 registers and feed the next product without a conversion).  A wave owns 32 queries (keys in the dK / dV kernel):
// lane (c = lane % 32, h = lane / 32) keeps row c of its block for the channel half h in registers, and the contraction
// index of MFMA step s is channel  h * ch/2 + s.  Products are formed TRANSPOSED (S^T = K Q^T: keys as rows, queries as
// columns), so that
//   * the softmax reduction over keys runs inside a lane (16 accumulator registers) plus one exchange with lane ^ 32, and
//   * the accumulator layout of S^T (lane: column c, rows 8 (r / 4) + 4 h + r % 4) IS the B-operand layout of the next
//     product (O^T += V^T P^T with contraction step r <-> that key): P never moves between registers or through LDS.
// K / V (or Q / dO) tiles of 32 rows are staged through LDS, double-buffered, one barrier per tile.
#include "fh_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));  // (native vector: arrays of it stay in registers)

struct AttnArgs {
  const float* qkv;   // [N][T][3C]
  const float* o;     // [N][T][C]   forward output (backward)
  const float* dout;  // [N][T][C]   its gradient (backward)
  float* out;         // [N][T][C]   (forward)
  float* lse;         // [N * heads][T]  log2 sum exp2 of the scaled logits (log2 units)
  float* dsum;        // [N * heads][T]  rowsum(dO . O)
  float* dqkv;        // [N][T][3C]
  int T, C, heads, hs, qo, ko, vo;  // head h: q at h * hs + qo, k at h * hs + ko, v at h * hs + vo (floats within a token row)
  float scale;        // 1 / sqrt(ch)
};

constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ floatx16 mfma(float a, float b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
// row (within a 32 x 32 accumulator tile) that register r of a lane in half lh holds
__device__ __forceinline__ int acc_row(int r, int lh) { return 8 * (r >> 2) + 4 * lh + (r & 3); }

// X^T-form tile product: acc[row = tile row][col = lane's register row] += sum_d tile[row][d] * reg[d]; the lane reads its tile
// row `lr`, channels DH * lh + 0 .. DH - 1, as float4s
template <int DH, int LD>
__device__ __forceinline__ floatx16 rows_times_regs(const float (*tile)[LD], int lr, int lh, const float* reg) {
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s4 = 0; s4 < DH / 4; ++s4) {
    const float4 v = *reinterpret_cast<const float4*>(&tile[lr][DH * lh + 4 * s4]);
    acc = mfma(v.x, reg[4 * s4 + 0], acc);
    acc = mfma(v.y, reg[4 * s4 + 1], acc);
    acc = mfma(v.z, reg[4 * s4 + 2], acc);
    acc = mfma(v.w, reg[4 * s4 + 3], acc);
  }
  return acc;
}

// out^T[d][c] += sum_rows tile[row][d] * w[row][c], w in accumulator layout (register st <-> tile row acc_row(st, lh))
template <int DT, int LD>
__device__ __forceinline__ void cols_times_acc(const float (*tile)[LD], int lr, int lh, const floatx16& w, floatx16* out) {
#pragma unroll
  for (int st = 0; st < 16; ++st) {
    const int row = acc_row(st, lh);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) out[dt] = mfma(tile[row][dt * 32 + lr], w[st], out[dt]);
  }
}

// the lane's 32 x D block row `p` (channels DH * lh ..) into registers, times `f`
template <int DH>
__device__ __forceinline__ void load_row_half(const float* p, float f, float* reg) {
#pragma unroll
  for (int s4 = 0; s4 < DH / 4; ++s4) {
    const float4 v = *reinterpret_cast<const float4*>(p + 4 * s4);
    reg[4 * s4 + 0] = v.x * f, reg[4 * s4 + 1] = v.y * f, reg[4 * s4 + 2] = v.z * f, reg[4 * s4 + 3] = v.w * f;
  }
}

// acc^T tile (rows d, column = the lane's query / key) -> global row-major [.][d], scaled
template <int DT>
__device__ __forceinline__ void store_cols(const floatx16* acc, float f, float* rowp, int lh) {
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(rowp + dt * 32 + 8 * j + 4 * lh) =
          make_float4(acc[dt][4 * j] * f, acc[dt][4 * j + 1] * f, acc[dt][4 * j + 2] * f, acc[dt][4 * j + 3] * f);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_attn_fwd(AttnArgs a) {
  constexpr int DH = D / 2, DT = D / 32, KLD = D + 4, VLD = D + 8, C4 = D / 4;
  __shared__ __align__(16) float Ks[2][32][KLD];  // read as rows (b128, 16 distinct bank quads per 16 lanes)
  __shared__ __align__(16) float Vs[2][32][VLD];  // read as columns: rows r and r + 4 sit 32 banks apart
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int nh = blockIdx.y, n = nh / a.heads, h = nh % a.heads;
  const int row = 3 * a.C;
  const float* base = a.qkv + (int64_t)n * a.T * row + h * a.hs;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bool active = q0 < a.T;  // (T % 32 == 0: a wave is all in or all out)
  float qreg[DH];
  if (active) load_row_half<DH>(base + (int64_t)(q0 + lr) * row + a.qo + DH * lh, a.scale * kLog2e, qreg);
  f4 rk[DT], rv[DT];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int e = 0; e < DT; ++e) {
      const int idx = e * 256 + tid, key = idx / C4, c4 = idx % C4;
      const float* p = base + (int64_t)(kt * 32 + key) * row + 4 * c4;
      rk[e] = *reinterpret_cast<const f4*>(p + a.ko);
      rv[e] = *reinterpret_cast<const f4*>(p + a.vo);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int e = 0; e < DT; ++e) {
      const int idx = e * 256 + tid, key = idx / C4, c4 = idx % C4;
      *reinterpret_cast<f4*>(&Ks[buf][key][4 * c4]) = rk[e];
      *reinterpret_cast<f4*>(&Vs[buf][key][4 * c4]) = rv[e];
    }
  };
  floatx16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int nkt = a.T / 32;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    if (active) {
      floatx16 s = rows_times_regs<DH, KLD>(Ks[buf], lr, lh, qreg);  // S^T tile: 32 keys x this lane's query (log2 units)
      float mt = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float mn = fmaxf(m, mt), alpha = ex2(m - mn);
      m = mn;
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = ex2(s[r] - mn);
        rs += s[r];
      }
      rs += __shfl_xor(rs, 32, 64);
      l = l * alpha + rs;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      cols_times_acc<DT, VLD>(Vs[buf], lr, lh, s, o);  // O^T += V^T P^T
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);  // (last read in iteration kt - 1, before its barrier)
    __syncthreads();
  }
  if (active) {
    store_cols<DT>(o, 1.f / l, a.out + ((int64_t)n * a.T + q0 + lr) * a.C + h * D, lh);
    if (lh == 0) a.lse[(int64_t)nh * a.T + q0 + lr] = m + __builtin_amdgcn_logf(l);  // (v_log_f32 = log2)
  }
}

// ------------------------------------------------------------------------------------------------
// backward, query side: D = rowsum(dO . O) and dQ = scale * dZ K with dZ = P . (dP - D), dP = dO V^T
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_attn_bwd_dq(AttnArgs a) {
  constexpr int DH = D / 2, DT = D / 32, KLD = D + 4, C4 = D / 4;
  __shared__ __align__(16) float Ks[2][32][KLD];
  __shared__ __align__(16) float Vs[2][32][KLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int nh = blockIdx.y, n = nh / a.heads, h = nh % a.heads;
  const int row = 3 * a.C;
  const float* base = a.qkv + (int64_t)n * a.T * row + h * a.hs;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bool active = q0 < a.T;
  float qreg[DH], doreg[DH];
  float lse = 0.f, dsum = 0.f;
  if (active) {
    load_row_half<DH>(base + (int64_t)(q0 + lr) * row + a.qo + DH * lh, a.scale * kLog2e, qreg);
    const int64_t orow = ((int64_t)n * a.T + q0 + lr) * a.C + h * D + DH * lh;
    load_row_half<DH>(a.dout + orow, 1.f, doreg);
#pragma unroll
    for (int s4 = 0; s4 < DH / 4; ++s4) {
      const float4 v = *reinterpret_cast<const float4*>(a.o + orow + 4 * s4);
      dsum += v.x * doreg[4 * s4] + v.y * doreg[4 * s4 + 1] + v.z * doreg[4 * s4 + 2] + v.w * doreg[4 * s4 + 3];
    }
    dsum += __shfl_xor(dsum, 32, 64);
    lse = a.lse[(int64_t)nh * a.T + q0 + lr];
    if (lh == 0) a.dsum[(int64_t)nh * a.T + q0 + lr] = dsum;
  }
  f4 rk[DT], rv[DT];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int e = 0; e < DT; ++e) {
      const int idx = e * 256 + tid, key = idx / C4, c4 = idx % C4;
      const float* p = base + (int64_t)(kt * 32 + key) * row + 4 * c4;
      rk[e] = *reinterpret_cast<const f4*>(p + a.ko);
      rv[e] = *reinterpret_cast<const f4*>(p + a.vo);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int e = 0; e < DT; ++e) {
      const int idx = e * 256 + tid, key = idx / C4, c4 = idx % C4;
      *reinterpret_cast<f4*>(&Ks[buf][key][4 * c4]) = rk[e];
      *reinterpret_cast<f4*>(&Vs[buf][key][4 * c4]) = rv[e];
    }
  };
  floatx16 dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  const int nkt = a.T / 32;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    if (active) {
      floatx16 s = rows_times_regs<DH, KLD>(Ks[buf], lr, lh, qreg);          // S^T (log2 units)
      const floatx16 dp = rows_times_regs<DH, KLD>(Vs[buf], lr, lh, doreg);  // dP^T = V dO^T
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = ex2(s[r] - lse) * (dp[r] - dsum);  // dZ^T
      cols_times_acc<DT, KLD>(Ks[buf], lr, lh, s, dq);                       // dQ^T += K^T dZ^T
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (active) store_cols<DT>(dq, a.scale, a.dqkv + ((int64_t)n * a.T + q0 + lr) * row + h * a.hs + a.qo, lh);
}

// ------------------------------------------------------------------------------------------------
// backward, key side: dV = P^T dO, dK = scale * dZ^T Q (P, dZ recomputed per 32-query tile)
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_attn_bwd_dkv(AttnArgs a) {
  constexpr int DH = D / 2, DT = D / 32, KLD = D + 4, C4 = D / 4;
  __shared__ __align__(16) float Qs[2][32][KLD];
  __shared__ __align__(16) float Os[2][32][KLD];  // dO tile
  __shared__ __align__(16) float Ls[2][2][32];    // [buf][0: lse, 1: rowsum(dO . O)][query]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int nh = blockIdx.y, n = nh / a.heads, h = nh % a.heads;
  const int row = 3 * a.C;
  const float* base = a.qkv + (int64_t)n * a.T * row + h * a.hs;
  const float* dobase = a.dout + (int64_t)n * a.T * a.C + h * D;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const bool active = k0 < a.T;
  float kreg[DH], vreg[DH];
  if (active) {
    load_row_half<DH>(base + (int64_t)(k0 + lr) * row + a.ko + DH * lh, 1.f, kreg);
    load_row_half<DH>(base + (int64_t)(k0 + lr) * row + a.vo + DH * lh, 1.f, vreg);
  }
  f4 rq[DT], ro[DT];
  float rl = 0.f;
  auto load_tile = [&](int qt) {
#pragma unroll
    for (int e = 0; e < DT; ++e) {
      const int idx = e * 256 + tid, q = idx / C4, c4 = idx % C4;
      rq[e] = *reinterpret_cast<const f4*>(base + (int64_t)(qt * 32 + q) * row + a.qo + 4 * c4);
      ro[e] = *reinterpret_cast<const f4*>(dobase + (int64_t)(qt * 32 + q) * a.C + 4 * c4);
    }
    if (tid < 64) rl = (tid < 32 ? a.lse : a.dsum)[(int64_t)nh * a.T + qt * 32 + (tid & 31)];
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int e = 0; e < DT; ++e) {
      const int idx = e * 256 + tid, q = idx / C4, c4 = idx % C4;
      *reinterpret_cast<f4*>(&Qs[buf][q][4 * c4]) = rq[e];
      *reinterpret_cast<f4*>(&Os[buf][q][4 * c4]) = ro[e];
    }
    if (tid < 64) Ls[buf][tid >> 5][tid & 31] = rl;
  };
  floatx16 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
  const float sl2 = a.scale * kLog2e;
  const int nqt = a.T / 32;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int qt = 0; qt < nqt; ++qt) {
    const int buf = qt & 1;
    if (qt + 1 < nqt) load_tile(qt + 1);
    if (active) {
      floatx16 s = rows_times_regs<DH, KLD>(Qs[buf], lr, lh, kreg);   // S tile: rows = queries, column = this lane's key
      floatx16 dp = rows_times_regs<DH, KLD>(Os[buf], lr, lh, vreg);  // dP tile
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 l4 = *reinterpret_cast<const float4*>(&Ls[buf][0][8 * j + 4 * lh]);
        const float4 d4 = *reinterpret_cast<const float4*>(&Ls[buf][1][8 * j + 4 * lh]);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = ex2(s[4 * j + i] * sl2 - lq[i]);
          s[4 * j + i] = p;
          dp[4 * j + i] = p * (dp[4 * j + i] - dq_[i]);
        }
      }
      cols_times_acc<DT, KLD>(Os[buf], lr, lh, s, dv);   // dV^T += dO^T P
      cols_times_acc<DT, KLD>(Qs[buf], lr, lh, dp, dk);  // dK^T += Q^T dZ
    }
    if (qt + 1 < nqt) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (active) {
    float* rowp = a.dqkv + ((int64_t)n * a.T + k0 + lr) * row + h * a.hs;
    store_cols<DT>(dk, a.scale, rowp + a.ko, lh);
    store_cols<DT>(dv, 1.f, rowp + a.vo, lh);
  }
}

int geometry(AttnArgs& a, int T, int C, int heads, int new_order) {
  if (T < 32 || T % 32 != 0 || heads < 1 || C % heads != 0) return 0;
  const int ch = C / heads;
  if (ch != 32 && ch != 64) return 0;
  a.T = T, a.C = C, a.heads = heads;
  if (new_order)
    a.hs = ch, a.qo = 0, a.ko = C, a.vo = 2 * C;       // QKVAttention: [q heads | k heads | v heads]
  else
    a.hs = 3 * ch, a.qo = 0, a.ko = ch, a.vo = 2 * ch;  // QKVAttentionLegacy: per head [q | k | v]
  a.scale = 1.f / sqrtf((float)ch);
  return ch;
}

}  // namespace

extern "C" {

int fh_attention_supported(int T, int C, int heads) {
  AttnArgs a;
  return geometry(a, T, C, heads, 0) != 0;
}

int fh_attention_fwd(const float* qkv, float* out, float* lse, int N, int T, int C, int heads, int new_order, void* stream) {
  AttnArgs a{};
  if (!qkv || !out || !lse || N < 1 || (int64_t)N * heads > 65535) return FH_EINVAL;
  const int ch = geometry(a, T, C, heads, new_order);
  if (!ch) return FH_ESIZE;
  a.qkv = qkv, a.out = out, a.lse = lse;
  const dim3 grid((T + 127) / 128, N * heads);
  if (ch == 64)
    hipLaunchKernelGGL(k_attn_fwd<64>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_attn_fwd<32>, grid, dim3(256), 0, (hipStream_t)stream, a);
  FH_LAUNCH_CHECK();
  return 0;
}

int fh_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dsum, float* dqkv, int N,
                     int T, int C, int heads, int new_order, void* stream) {
  AttnArgs a{};
  if (!qkv || !out || !dout || !lse || !dsum || !dqkv || N < 1 || (int64_t)N * heads > 65535) return FH_EINVAL;
  const int ch = geometry(a, T, C, heads, new_order);
  if (!ch) return FH_ESIZE;
  a.qkv = qkv, a.o = out, a.dout = dout, a.lse = const_cast<float*>(lse), a.dsum = dsum, a.dqkv = dqkv;  // (lse is only read)
  const dim3 grid((T + 127) / 128, N * heads);
  hipStream_t st = (hipStream_t)stream;
  if (ch == 64) {
    hipLaunchKernelGGL(k_attn_bwd_dq<64>, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_attn_bwd_dkv<64>, grid, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(k_attn_bwd_dq<32>, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_attn_bwd_dkv<32>, grid, dim3(256), 0, st, a);
  }
  FH_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
