// Fused backward chains of one LocalAttention + ResidualNorm iteration, on 64-row tiles with the
// forward kernels' machinery: split-fp16 MFMA projections (scann_mma.h), everything between the GEMMs in the accumulator
// layout (lane = row, 16 of the wave's 32 columns).  At batch 128 a kernel of the modular backward (scann_train.hip) is one
// partially filled round of workgroups, i.e. a launch costs its latency chain whatever it computes; these kernels replace
//   rn_bwd_kernel   : [linear_sum of the layer above] + ln_bwd + dropout_copy + linear(Wf2^T, swish') + linear(Wf1^T)   (5 -> 1)
//   edge_bwd_kernel : linear(Wk^T) + ln_bwd_edge + linear(W2^T)                                                        (3 -> 1)
//   atom_gather3_kernel : gather_prod_sum + atom_sums                                                                  (2 -> 1)
// Gradients are not O(1) like activations, so every GEMM input row is multiplied by a power of two that brings its largest
// magnitude to 2^12 before the hi / lo fp16 split (exact), and the accumulators by the inverse: the split keeps 22 significant
// bits relative to the row's maximum whatever the loss scale.
// Formulas: the exact derivatives of attention.py:37-40 (ResidualNorm) and :141-163 (geometry update, gate, key), as in
// the modular kernels they replace (scann_train.hip: ln_bwd_kernel, linear_kernel, dropout_copy_kernel).
#include "scann_internal.h"
#include <hip/hip_ext.h>
#include "scann_mma.h"
#include "scann_train.h"

#include <algorithm>

namespace scann {

namespace {

__device__ __forceinline__ float sigm_(float x) { return 1.0f / (1.0f + expf(-x)); }
// d/dx [x * sigmoid(x)] = s * (1 + x * (1 - s))
__device__ __forceinline__ float dsw_(float x) {
  const float s = sigm_(x);
  return s * (1.0f + x * (1.0f - s));
}

__device__ __forceinline__ float f4sum_(float4 a) { return (a.x + a.y) + (a.z + a.w); }

constexpr float WINV = 1.0f / WSCALE;
constexpr int STAGE_STRIDE = D + 4;  // floats per row of the gamma / beta staging (aliases the plane buffer)

// scale = 2^(12 - e) for a row whose largest magnitude has binary exponent e (1 for an all-zero row); post = WINV / scale
__device__ __forceinline__ void row_scale(float rowmax, float& scale, float& post) {
  const int e = (__float_as_int(rowmax) >> 23) & 0xff;
  const int se = min(253, 266 - e);  // biased exponent of the scale; e <= 254 -> se >= 12
  scale = e == 0 ? 1.0f : __int_as_float(se << 23);
  post = e == 0 ? WINV : __int_as_float((254 - se) << 23) * WINV;
}

// largest magnitude of the row: this lane's 16 columns, its partner lane, then (through LDS) the row's four waves
template <int RT>
__device__ __forceinline__ void put_rowmax(float* __restrict__ sMax, const float4 (&v)[RT][4], int lrow, int lh, int wave) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v[rt][j].x), fabsf(v[rt][j].y))), fmaxf(fabsf(v[rt][j].z), fabsf(v[rt][j].w)));
    m = xor32_max(m);
    if (lh == 0) sMax[(lrow + 32 * rt) * 4 + wave] = m;
  }
}
__device__ __forceinline__ float get_rowmax(const float* __restrict__ sMax, int row) {
  const float4 m = *reinterpret_cast<const float4*>(&sMax[row * 4]);
  return fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w));
}

// the scaled row -> hi / lo planes
__device__ __forceinline__ void put_planes(_Float16* __restrict__ sH, _Float16* __restrict__ sL, const float4 (&v)[4], float scale, int row,
                                           int cbase) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f16x4 h, l;
    split4(make_float4(v[j].x * scale, v[j].y * scale, v[j].z * scale, v[j].w * scale), h, l);
    *reinterpret_cast<f16x4*>(sH + row * PLANE_STRIDE + cbase + 8 * j) = h;
    *reinterpret_cast<f16x4*>(sL + row * PLANE_STRIDE + cbase + 8 * j) = l;
  }
}

// LayerNorm statistics of the tile's rows (as in atom_kernel): per-lane partial (mean of 32, M2 of 32) -> LDS
template <int RT>
__device__ __forceinline__ void put_stats(float* __restrict__ sRed, const float4 (&x)[RT][4], int lrow, int lh, int wave) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += f4sum_(x[rt][j]);
    const float mean32 = xor32(s) * (1.0f / 32.0f);
    float v2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = x[rt][j].x - mean32, b = x[rt][j].y - mean32, c = x[rt][j].z - mean32, d = x[rt][j].w - mean32;
      v2 += (a * a + b * b) + (c * c + d * d);
    }
    const float m2 = xor32(v2);
    if (lh == 0) *reinterpret_cast<float2*>(&sRed[((lrow + 32 * rt) * 4 + wave) * 2]) = make_float2(mean32, m2);
  }
}
__device__ __forceinline__ void get_stats(const float* __restrict__ sRed, int row, float& mean, float& rstd) {
  const float4 sa = *reinterpret_cast<const float4*>(&sRed[row * 8]), sb = *reinterpret_cast<const float4*>(&sRed[row * 8 + 4]);
  mean = ((sa.x + sa.z) + (sb.x + sb.z)) * 0.25f;
  const float d0 = sa.x - mean, d1 = sa.z - mean, d2 = sb.x - mean, d3 = sb.z - mean;
  const float var = (((sa.y + sa.w) + (sb.y + sb.w)) + 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3))) * (1.0f / D);
  rstd = 1.0f / sqrtf(var + 1e-6f);
}

// LayerNorm backward in the accumulator layout, first half: x -> xhat, dy -> ax = dy * gamma; the row sums of ax and ax * xhat go
// to sRed, the tile's column sums of dy * xhat and dy (gamma / beta gradients) to the staging buffer (two row tiles pre-added).
template <int RT>
__device__ __forceinline__ void ln_bwd_head(float4 (&x)[RT][4], float4 (&dy)[RT][4], const float* __restrict__ sStat, const float* __restrict__ sGamma,
                                            float* __restrict__ sSum, float* __restrict__ stage, float (&rstd)[RT], int lrow, int lh, int wave,
                                            int cbase) {
  float4 dgm[4], dbt[4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float mean;
    get_stats(sStat, lrow + 32 * rt, mean, rstd[rt]);
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 g = *reinterpret_cast<const float4*>(&sGamma[cbase + 8 * j]);
      const float4 xh = make_float4((x[rt][j].x - mean) * rstd[rt], (x[rt][j].y - mean) * rstd[rt], (x[rt][j].z - mean) * rstd[rt],
                                    (x[rt][j].w - mean) * rstd[rt]);
      const float4 d = dy[rt][j];
      const float4 gx = make_float4(d.x * xh.x, d.y * xh.y, d.z * xh.z, d.w * xh.w);
      if (rt == 0) {
        dgm[j] = gx;
        dbt[j] = d;
      } else {
        dgm[j] = make_float4(dgm[j].x + gx.x, dgm[j].y + gx.y, dgm[j].z + gx.z, dgm[j].w + gx.w);
        dbt[j] = make_float4(dbt[j].x + d.x, dbt[j].y + d.y, dbt[j].z + d.z, dbt[j].w + d.w);
      }
      const float4 ax = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
      p1 += f4sum_(ax);
      p2 += (ax.x * xh.x + ax.y * xh.y) + (ax.z * xh.z + ax.w * xh.w);
      x[rt][j] = xh;
      dy[rt][j] = ax;
    }
    p1 = xor32(p1);
    p2 = xor32(p2);
    if (lh == 0) *reinterpret_cast<float2*>(&sSum[((lrow + 32 * rt) * 4 + wave) * 2]) = make_float2(p1, p2);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    *reinterpret_cast<float4*>(&stage[lrow * STAGE_STRIDE + cbase + 8 * j]) = dgm[j];
    *reinterpret_cast<float4*>(&stage[(32 + lrow) * STAGE_STRIDE + cbase + 8 * j]) = dbt[j];
  }
}
// second half (after the barrier): x := dx = rstd * (ax - mean(ax) - xhat * mean(ax * xhat)); the workgroup's gamma / beta slot
template <int RT>
__device__ __forceinline__ void ln_bwd_tail(float4 (&x)[RT][4], const float4 (&ax)[RT][4], const float* __restrict__ sSum, const float* __restrict__ stage,
                                            const float (&rstd)[RT], float* __restrict__ dgamma, float* __restrict__ dbeta, int lrow, int tid) {
  {
    const int which = tid >> 7, col = tid & (D - 1);
    float tot = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) tot += stage[(32 * which + r) * STAGE_STRIDE + col];
    (which ? dbeta : dgamma)[(size_t)blockIdx.x * D + col] = tot;  // summed in slot order by wgrad_reduce_kernel
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = lrow + 32 * rt;
    const float4 sa = *reinterpret_cast<const float4*>(&sSum[row * 8]), sb = *reinterpret_cast<const float4*>(&sSum[row * 8 + 4]);
    const float m1 = ((sa.x + sa.z) + (sb.x + sb.z)) * (1.0f / D), m2 = ((sa.y + sa.w) + (sb.y + sb.w)) * (1.0f / D);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 a = ax[rt][j], xh = x[rt][j];
      x[rt][j] = make_float4(rstd[rt] * (a.x - m1 - xh.x * m2), rstd[rt] * (a.y - m1 - xh.y * m2), rstd[rt] * (a.z - m1 - xh.z * m2),
                             rstd[rt] * (a.w - m1 - xh.w * m2));
    }
  }
}

}  // namespace

// ---- ResidualNorm backward (attention.py:37-40): c' = LN(T2), T2 = x + drop(Y), Y = H1 W2 + b2, H1 = swish(pre1), pre1 = x W1 + b1 ----
// in : dC = d loss / d c' (+ optionally the projections of the layer above: dC += X0 W0^T + X1 W1^T + X2 W2^T), T2, pre1
// out: dY (operand of dW2), dpre1 (operand of dW1), dCtx = dT2 + dpre1 W1^T, gamma / beta slots
template <int RT>
__global__ __launch_bounds__(256, RT == 2 ? 2 : 3) void rn_bwd_kernel(RnBwdArgs a) {
  constexpr int TR = 32 * RT;  // rows per tile
  __shared__ __attribute__((aligned(16))) unsigned char sTile[2 * 64 * PLANE_STRIDE * 2];  // hi / lo planes of <= 64 rows | gamma-beta staging (64 rows of fp32 either way)
  __shared__ __attribute__((aligned(16))) float sStat[TR * 8], sSum[TR * 8];
  __shared__ __attribute__((aligned(16))) float sMax[3][TR * 4];
  __shared__ __attribute__((aligned(16))) float sGamma[D];
  _Float16* const sH = reinterpret_cast<_Float16*>(sTile);
  _Float16* const sL = sH + TR * PLANE_STRIDE;
  float* const stage = reinterpret_cast<float*>(sTile);
  static_assert(64 * STAGE_STRIDE * 4 <= (int)sizeof(sTile), "gamma / beta staging must fit the plane buffer");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5, cbase = 32 * wave + 4 * lh;
  const int row0 = blockIdx.x * TR;
  const int nrows = min(TR, a.n_atom - row0);

  f16x8 whA[4], wlA[4], whB[4], wlB[4];
  const _Float16* const firstW = a.n_pre > 0 ? a.Wh[0] : a.Wf2Th;
  load_wsplit<4, 8>(firstW, wave, lane, whA, wlA, 0);
  load_wsplit<4, 8>(firstW, wave, lane, whB, wlB, 4);
  if (tid < D) sGamma[tid] = a.gamma[tid];

  unsigned off[RT];
  float4 x[RT][4], dy[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    off[rt] = ((unsigned)(row0 + min(lrow + 32 * rt, nrows - 1)) * D + cbase) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[rt][j] = ld4(a.T2, off[rt] + 32 * j);
      dy[rt][j] = a.dC ? ld4(a.dC, off[rt] + 32 * j) : make_float4(0.f, 0.f, 0.f, 0.f);  // null: the pre-terms are the whole gradient
    }
  }
  f32x16 acc[RT];
  // projections of the layer above that end in this tile's rows (scann_train.hip: linear_sum_kernel)
  for (int t = 0; t < a.n_pre; ++t) {
    float4 v[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[rt][j] = ld4(a.X[t], off[rt] + 32 * j);
    put_rowmax<RT>(sMax[0], v, lrow, lh, wave);
    __syncthreads();  // (also: every wave is done with the planes of the previous term)
    float post[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float scale;
      row_scale(get_rowmax(sMax[0], lrow + 32 * rt), scale, post[rt]);
      put_planes(sH, sL, v[rt], scale, lrow + 32 * rt, cbase);
    }
    __syncthreads();
    gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, t + 1 < a.n_pre ? a.Wh[t + 1] : a.Wf2Th, wave, lane, acc);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dy[rt][j].x = fmaf(acc[rt][4 * j], post[rt], dy[rt][j].x);
        dy[rt][j].y = fmaf(acc[rt][4 * j + 1], post[rt], dy[rt][j].y);
        dy[rt][j].z = fmaf(acc[rt][4 * j + 2], post[rt], dy[rt][j].z);
        dy[rt][j].w = fmaf(acc[rt][4 * j + 3], post[rt], dy[rt][j].w);
      }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
    if (lrow + 32 * rt >= nrows)
#pragma unroll
      for (int j = 0; j < 4; ++j) dy[rt][j] = make_float4(0.f, 0.f, 0.f, 0.f);  // rows past the end contribute exact zeros
  if (a.dC_out) {  // the complete d loss / d c' (operand of nothing here; kept for the layer's own consumers when asked)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      if (lrow + 32 * rt < nrows)
#pragma unroll
        for (int j = 0; j < 4; ++j) st4(a.dC_out, off[rt] + 32 * j, dy[rt][j]);
  }
  put_stats<RT>(sStat, x, lrow, lh, wave);
  __syncthreads();  // statistics (and sGamma) visible; the planes are free
  float rstd[RT];
  ln_bwd_head<RT>(x, dy, sStat, sGamma, sSum, stage, rstd, lrow, lh, wave, cbase);
  __syncthreads();
  ln_bwd_tail<RT>(x, dy, sSum, stage, rstd, a.dgamma, a.dbeta, lrow, tid);  // x = dT2
  // dY = dT2 through the Dropout mask of the forward (attention.py:29)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 v = x[rt][j];
      if (a.drop_p > 0.f) {
        const size_t e = (size_t)(row0 + lrow + 32 * rt) * D + cbase + 8 * j;
        v.x *= drop_scale(a.drop_seed, a.drop_tag, e, a.drop_p);
        v.y *= drop_scale(a.drop_seed, a.drop_tag, e + 1, a.drop_p);
        v.z *= drop_scale(a.drop_seed, a.drop_tag, e + 2, a.drop_p);
        v.w *= drop_scale(a.drop_seed, a.drop_tag, e + 3, a.drop_p);
      }
      dy[rt][j] = v;
    }
  put_rowmax<RT>(sMax[1], dy, lrow, lh, wave);
  __syncthreads();  // row maxima visible; every thread is done with the staging buffer
  float post[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = lrow + 32 * rt;
    float scale;
    row_scale(get_rowmax(sMax[1], row), scale, post[rt]);
    if (row < nrows)
#pragma unroll
      for (int j = 0; j < 4; ++j) st4(a.dY, off[rt] + 32 * j, dy[rt][j]);
    put_planes(sH, sL, dy[rt], scale, row, cbase);
  }
  // pre1 rows (for swish') arrive under the GEMM, in the registers dY just left
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) dy[rt][j] = ld4(a.pre1, off[rt] + 32 * j);
  __syncthreads();
  gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, a.Wf1Th, wave, lane, acc);  // dH1 = dY . W2^T
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 p = dy[rt][j];
      dy[rt][j] = make_float4(acc[rt][4 * j] * post[rt] * dsw_(p.x), acc[rt][4 * j + 1] * post[rt] * dsw_(p.y),
                              acc[rt][4 * j + 2] * post[rt] * dsw_(p.z), acc[rt][4 * j + 3] * post[rt] * dsw_(p.w));  // dpre1
    }
  put_rowmax<RT>(sMax[2], dy, lrow, lh, wave);
  __syncthreads();  // every wave is done reading the dY planes
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = lrow + 32 * rt;
    float scale;
    row_scale(get_rowmax(sMax[2], row), scale, post[rt]);
    if (row < nrows)
#pragma unroll
      for (int j = 0; j < 4; ++j) st4(a.dpre1, off[rt] + 32 * j, dy[rt][j]);
    put_planes(sH, sL, dy[rt], scale, row, cbase);
  }
  __syncthreads();
  gemm_tile<false, RT>(sH, sL, whA, wlA, whB, wlB, nullptr, wave, lane, acc);  // dpre1 . W1^T
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
    if (lrow + 32 * rt < nrows)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        st4(a.dCtx, off[rt] + 32 * j,
            make_float4(fmaf(acc[rt][4 * j], post[rt], x[rt][j].x), fmaf(acc[rt][4 * j + 1], post[rt], x[rt][j].y),
                        fmaf(acc[rt][4 * j + 2], post[rt], x[rt][j].z), fmaf(acc[rt][4 * j + 3], post[rt], x[rt][j].w)));
}

// ---- key / gate / geometry-update backward of one layer's edges (attention.py:141-163) -------------------------------------
// in : dK, centres c (gate ang = c[j] * G'), dG' of the layer below (null for the last layer), T (LayerNorm_g input), V
// out: dang = dK Wk^T, dV = dT * swish'(V), dG = dT + dV W2^T, gamma / beta slots      (dT = LN_g backward of dang * c[j] + dG')
// ATT (32-row tiles of the forward's plan: whole atoms, every degree <= 16): the tile's dK rows do not come from memory -- the waves
// first run attn_bwd16_kernel's arithmetic on the tile's atoms (softmax / LayerNorm backward per atom, one atom per wave at a time),
// leaving dq and dK in memory for the weight gradients and dK in a staging buffer for the chain below.  One launch less per layer.
#define SCANN_DPP_(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float red8_(float v) {
  v += SCANN_DPP_(v, 0xB1);   // lane ^ 1
  v += SCANN_DPP_(v, 0x4E);   // lane ^ 2
  v += SCANN_DPP_(v, 0x141);  // the other quad of the group of 8
  return v;
}
__device__ __forceinline__ float red64_(float v) {
  v = red8_(v);
  v += SCANN_DPP_(v, 0x128);
  v = xor16(v);  // (v_permlane16_swap / v_permlane32_swap: no LDS round trip, same additions)
  v = xor32(v);
  return v;
}

template <int RT, bool ATT = false>
__global__ __launch_bounds__(256, RT == 2 ? 2 : 3) void edge_bwd_kernel(EdgeBwdArgs a, AttnPart b) {
  static_assert(!ATT || RT == 1, "the fused attention backward runs on the 32-row tile plan");
  constexpr int TR = 32 * RT;  // rows per tile
  __shared__ __attribute__((aligned(16))) unsigned char sTile[2 * 64 * PLANE_STRIDE * 2];
  __shared__ __attribute__((aligned(16))) float sStat[TR * 8], sSum[TR * 8];
  __shared__ __attribute__((aligned(16))) float sMax[2][TR * 4];
  __shared__ __attribute__((aligned(16))) float sGamma[D];
  _Float16* const sH = reinterpret_cast<_Float16*>(sTile);
  _Float16* const sL = sH + TR * PLANE_STRIDE;
  float* const stage = reinterpret_cast<float*>(sTile);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5, cbase = 32 * wave + 4 * lh;
  EdgeTile tile = {0, 0, 0, 0};
  if (ATT) tile = b.tiles[blockIdx.x];
  const int row0 = ATT ? tile.edge_begin : blockIdx.x * TR;
  const int nrows = ATT ? tile.edge_end - tile.edge_begin : min(TR, a.n_edge - row0);
  float* const stageK = reinterpret_cast<float*>(sTile + 2 * 64 * PLANE_STRIDE);  // ATT: [32][STAGE_STRIDE] dK rows (upper half of the buffer)

  if (ATT) {
    __shared__ float sred[4][4 * 64];
    const float2 g = reinterpret_cast<const float2*>(b.gamma)[lane];
    float2 dg = make_float2(0.f, 0.f), dbt = dg;
    for (int at = tile.atom_begin + wave; at < tile.atom_end; at += 4) {  // attn_bwd16_kernel's body, atom by atom
      const int e0 = b.edge_offset[at], deg = b.edge_offset[at + 1] - e0;
      const float2 q2 = reinterpret_cast<const float2*>(b.q)[(size_t)at * 64 + lane];
      const float2 dyv = reinterpret_cast<const float2*>(b.dctx)[(size_t)at * 64 + lane];
      const float qx = q2.x * 0.25f, qy = q2.y * 0.25f;
      float2 k2[16];
      float ev[16];
      float m = -INFINITY;
      if (deg > 0) {
#pragma unroll
        for (int n = 0; n < 16; ++n) k2[n] = reinterpret_cast<const float2*>(b.K)[(size_t)(e0 + min(n, deg - 1)) * 64 + lane];
      }
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        if (n >= deg) k2[n] = make_float2(0.f, 0.f);
        float e = qx * k2[n].x + qy * k2[n].y;
        e = red8_(e);
        ev[n] = n < deg ? e : -INFINITY;
        m = fmaxf(m, ev[n]);
      }
      float ssum = 0.f;
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        ev[n] = n < deg ? __builtin_amdgcn_exp2f((ev[n] - m) * 1.44269504088896340736f) : 0.f;
        ssum += ev[n];
      }
      const float rsum = deg > 0 ? __builtin_amdgcn_rcpf(ssum) : 0.f;
      float px = q2.x, py = q2.y;
      float keep[16];
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        ev[n] = ev[n] * rsum;
        keep[n] = (b.drop_p > 0.f && n < deg) ? drop_scale(b.drop_seed, b.drop_tag, (size_t)(e0 + n) * NHEAD + (lane >> 3), b.drop_p) : 1.0f;
        px += ev[n] * keep[n] * k2[n].x;
        py += ev[n] * keep[n] * k2[n].y;
      }
      const float sm = red64_(px + py);
      const float mean = sm * (1.0f / D);
      const float cx = px - mean, cy = py - mean;
      const float v = red64_(cx * cx + cy * cy);
      const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
      const float hx = cx * rstd, hy = cy * rstd;
      const float ax = dyv.x * g.x, ay = dyv.y * g.y;
      const float m1 = red64_(ax + ay) * (1.0f / D), m2 = red64_(ax * hx + ay * hy) * (1.0f / D);
      const float dpx = rstd * (ax - m1 - hx * m2), dpy = rstd * (ay - m1 - hy * m2);
      dg.x += dyv.x * hx; dg.y += dyv.y * hy; dbt.x += dyv.x; dbt.y += dyv.y;
      float da[16];
      float dot = 0.f;
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        float d = dpx * k2[n].x + dpy * k2[n].y;
        d = red8_(d);
        da[n] = d * keep[n];
        dot += ev[n] * da[n];
      }
      float dqx = dpx, dqy = dpy;
#pragma unroll
      for (int n = 0; n < 16; ++n)
        if (n < deg) {
          const float de = ev[n] * (da[n] - dot);
          const float2 dk = make_float2(ev[n] * keep[n] * dpx + 0.25f * de * q2.x, ev[n] * keep[n] * dpy + 0.25f * de * q2.y);
          reinterpret_cast<float2*>(b.dK)[(size_t)(e0 + n) * 64 + lane] = dk;
          *reinterpret_cast<float2*>(&stageK[(e0 + n - row0) * STAGE_STRIDE + 2 * lane]) = dk;
          dqx += 0.25f * de * k2[n].x;
          dqy += 0.25f * de * k2[n].y;
        }
      reinterpret_cast<float2*>(b.dq)[(size_t)at * 64 + lane] = make_float2(dqx, dqy);
    }
    sred[wave][lane] = dg.x; sred[wave][64 + lane] = dg.y; sred[wave][128 + lane] = dbt.x; sred[wave][192 + lane] = dbt.y;
    __syncthreads();  // (also: every dK row of the tile is in the staging buffer)
    const float tot = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
    const int ln = tid & 63, which = tid >> 6;
    (which < 2 ? b.dgamma : b.dbeta)[(size_t)blockIdx.x * D + 2 * ln + (which & 1)] = tot;  // this workgroup's slot
  }

  f16x8 whA[4], wlA[4], whB[4], wlB[4];
  load_wsplit<4, 8>(a.WkTh, wave, lane, whA, wlA, 0);
  load_wsplit<4, 8>(a.WkTh, wave, lane, whB, wlB, 4);
  if (tid < D) sGamma[tid] = a.gamma[tid];

  unsigned off[RT], noff[RT];
  float4 x[RT][4], dy[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    // (ATT: a tile of atoms without edges has no rows -- every load below reads a valid row and is ignored)
    const int rc = min(row0 + min(lrow + 32 * rt, max(nrows - 1, 0)), a.n_edge - 1);
    off[rt] = ((unsigned)rc * D + cbase) * 4;
    noff[rt] = ((unsigned)a.nb[rc] * D + cbase) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dy[rt][j] = lrow + 32 * rt >= nrows ? make_float4(0.f, 0.f, 0.f, 0.f)
                  : ATT ? *reinterpret_cast<const float4*>(&stageK[(lrow + 32 * rt) * STAGE_STRIDE + cbase + 8 * j])
                        : ld4(a.dK, off[rt] + 32 * j);
      // T = swish(V) + G (attention.py:150-152): read back, or formed again from V and the geometry that entered the layer -- the
      // forward's own instruction (swish_plus), the same bits -- so that the training forward need not write it
      x[rt][j] = a.T ? ld4(a.T, off[rt] + 32 * j) : f4swish_plus(ld4(a.V, off[rt] + 32 * j), ld4(a.G, off[rt] + 32 * j));
    }
  }
  put_rowmax<RT>(sMax[0], dy, lrow, lh, wave);
  put_stats<RT>(sStat, x, lrow, lh, wave);
  __syncthreads();
  float post[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float scale;
    row_scale(get_rowmax(sMax[0], lrow + 32 * rt), scale, post[rt]);
    put_planes(sH, sL, dy[rt], scale, lrow + 32 * rt, cbase);
  }
  // the gate's centre rows c[j] arrive under the GEMM
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) dy[rt][j] = ld4(a.c, noff[rt] + 32 * j);
  __syncthreads();
  f32x16 acc[RT];
  gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, a.W2Th, wave, lane, acc);  // dang = dK . Wk^T
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const bool live = lrow + 32 * rt < nrows;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 da = make_float4(acc[rt][4 * j] * post[rt], acc[rt][4 * j + 1] * post[rt], acc[rt][4 * j + 2] * post[rt],
                                    acc[rt][4 * j + 3] * post[rt]);
      if (live) st4(a.dang, off[rt] + 32 * j, da);
      const float4 cn = dy[rt][j];
      float4 g = make_float4(da.x * cn.x, da.y * cn.y, da.z * cn.z, da.w * cn.w);  // d loss / d G' through the gate
      if (a.dG_in) {
        const float4 o = ld4(a.dG_in, off[rt] + 32 * j);
        g = make_float4(g.x + o.x, g.y + o.y, g.z + o.z, g.w + o.w);
      }
      dy[rt][j] = live ? g : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();  // every wave is done reading the dK planes: the staging buffer may be written
  float rstd[RT];
  ln_bwd_head<RT>(x, dy, sStat, sGamma, sSum, stage, rstd, lrow, lh, wave, cbase);
  __syncthreads();
  ln_bwd_tail<RT>(x, dy, sSum, stage, rstd, a.dgamma, a.dbeta, lrow, tid);  // x = dT
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = ld4(a.V, off[rt] + 32 * j), t = x[rt][j];
      dy[rt][j] = make_float4(t.x * dsw_(v.x), t.y * dsw_(v.y), t.z * dsw_(v.z), t.w * dsw_(v.w));  // dV
    }
  put_rowmax<RT>(sMax[1], dy, lrow, lh, wave);
  __syncthreads();  // row maxima visible; every thread is done with the staging buffer
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = lrow + 32 * rt;
    float scale;
    row_scale(get_rowmax(sMax[1], row), scale, post[rt]);
    if (row < nrows)
#pragma unroll
      for (int j = 0; j < 4; ++j) st4(a.dV, off[rt] + 32 * j, dy[rt][j]);
    put_planes(sH, sL, dy[rt], scale, row, cbase);
  }
  __syncthreads();
  gemm_tile<false, RT>(sH, sL, whA, wlA, whB, wlB, nullptr, wave, lane, acc);  // dV . W2^T
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
    if (lrow + 32 * rt < nrows)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        st4(a.dG, off[rt] + 32 * j,
            make_float4(fmaf(acc[rt][4 * j], post[rt], x[rt][j].x), fmaf(acc[rt][4 * j + 1], post[rt], x[rt][j].y),
                        fmaf(acc[rt][4 * j + 2], post[rt], x[rt][j].z), fmaf(acc[rt][4 * j + 3], post[rt], x[rt][j].w)));
}

// ---- the three atom-indexed sums of a layer in one launch ----------------------------------------------------------------
// dC[a] = sum over the edges e that point AT atom a of dang[e] * G'[e] (gate ang = c[j] * G'), dP3[a] = the same sum of dV[e],
// dP1[a] = sum of dV over atom a's own CSR row.  Fixed order, no atomics; loads of four edges in flight per step.
__global__ __launch_bounds__(256) void atom_gather3_kernel(const float4* __restrict__ dang, const float4* __restrict__ G, const float4* __restrict__ dV,
                                                           const int* __restrict__ edge_offset, const int* __restrict__ in_off,
                                                           const int* __restrict__ in_edge, float4* __restrict__ dC, float4* __restrict__ dP1,
                                                           float4* __restrict__ dP3, int n_atom) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_atom * 32) return;
  const int at = (int)(i >> 5), c4 = (int)(i & 31);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), t = s, u = s;
  const int e0 = edge_offset[at], e1 = edge_offset[at + 1], k0 = in_off[at], k1 = in_off[at + 1];
  int k = k0;
  for (; k + 4 <= k1; k += 4) {
    size_t o[4];
    float4 p[4], q[4], r[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) o[m] = (size_t)in_edge[k + m] * 32 + c4;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      p[m] = dang[o[m]];
      q[m] = G[o[m]];
      r[m] = dV[o[m]];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      s.x += p[m].x * q[m].x; s.y += p[m].y * q[m].y; s.z += p[m].z * q[m].z; s.w += p[m].w * q[m].w;
      t.x += r[m].x; t.y += r[m].y; t.z += r[m].z; t.w += r[m].w;
    }
  }
  for (; k < k1; ++k) {
    const size_t o = (size_t)in_edge[k] * 32 + c4;
    const float4 p = dang[o], q = G[o], r = dV[o];
    s.x += p.x * q.x; s.y += p.y * q.y; s.z += p.z * q.z; s.w += p.w * q.w;
    t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
  }
  int e = e0;
  for (; e + 4 <= e1; e += 4) {
    float4 r[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) r[m] = dV[(size_t)(e + m) * 32 + c4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { u.x += r[m].x; u.y += r[m].y; u.z += r[m].z; u.w += r[m].w; }
  }
  for (; e < e1; ++e) {
    const float4 r = dV[(size_t)e * 32 + c4];
    u.x += r.x; u.y += r.y; u.z += r.z; u.w += r.w;
  }
  dC[i] = s;
  dP3[i] = t;
  dP1[i] = u;
}

// 32-row tiles while the launch fits one round of workgroups (three per CU): it is then the latency chain of a tile, and a 32-row
// tile's chain is shorter (same choice as launch_atom / the edge-tile plan); 64-row tiles beyond
static int fused_tile_rows(int rows) { return rows <= 32 * 768 ? 32 : 64; }
int tile_slots(int rows) { return (rows + fused_tile_rows(rows) - 1) / fused_tile_rows(rows); }

void launch_rn_bwd(WgradCtx& ctx, RnBwdArgs a, float* dgamma, float* dbeta, hipStream_t s) {
  if (a.n_atom <= 0) return;
  const int n = tile_slots(a.n_atom);
  a.dgamma = reserve_vec(ctx, dgamma, n);
  a.dbeta = reserve_vec(ctx, dbeta, n);
  if (fused_tile_rows(a.n_atom) == 32) hipLaunchKernelGGL(rn_bwd_kernel<1>, dim3(n), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(rn_bwd_kernel<2>, dim3(n), dim3(256), 0, s, a);
}
void launch_edge_bwd(WgradCtx& ctx, EdgeBwdArgs a, float* dgamma, float* dbeta, hipStream_t s) {
  if (a.n_edge <= 0) return;
  const int n = tile_slots(a.n_edge);
  a.dgamma = reserve_vec(ctx, dgamma, n);
  a.dbeta = reserve_vec(ctx, dbeta, n);
  if (fused_tile_rows(a.n_edge) == 32) hipLaunchKernelGGL((edge_bwd_kernel<1, false>), dim3(n), dim3(256), 0, s, a, AttnPart{});
  else hipLaunchKernelGGL((edge_bwd_kernel<2, false>), dim3(n), dim3(256), 0, s, a, AttnPart{});
}
void launch_attn_edge_bwd(WgradCtx& ctx, EdgeBwdArgs a, AttnPart b, int n_tile, float* dgamma_g, float* dbeta_g, float* dgamma_ln,
                          float* dbeta_ln, hipStream_t s, hipEvent_t done) {
  if (a.n_edge <= 0 || n_tile <= 0) {
    if (done) (void)hipEventRecord(done, s);
    return;
  }
  a.dgamma = reserve_vec(ctx, dgamma_g, n_tile);
  a.dbeta = reserve_vec(ctx, dbeta_g, n_tile);
  b.dgamma = reserve_vec(ctx, dgamma_ln, n_tile);
  b.dbeta = reserve_vec(ctx, dbeta_ln, n_tile);
  // `done` (or null): recorded by the kernel's OWN completion signal (hipExtLaunchKernelGGL's stop event) -- a side stream can wait for
  // it without a marker packet behind the kernel on this stream: 4.0 instead of 6.7 us per fork (tools/fork_probe.hip, mode 3 vs 0)
  if (done) hipExtLaunchKernelGGL((edge_bwd_kernel<1, true>), dim3(n_tile), dim3(256), 0, s, nullptr, done, 0, a, b);
  else hipLaunchKernelGGL((edge_bwd_kernel<1, true>), dim3(n_tile), dim3(256), 0, s, a, b);
}
void launch_atom_gather3(const float* dang, const float* G, const float* dV, const int* edge_offset, const int* in_off, const int* in_edge,
                         float* dC, float* dP1, float* dP3, int n_atom, hipStream_t s, hipEvent_t done) {
  if (n_atom <= 0) {
    if (done) (void)hipEventRecord(done, s);
    return;
  }
  const dim3 grid((unsigned)(((size_t)n_atom * 32 + 255) / 256));
  if (done)
    hipExtLaunchKernelGGL(atom_gather3_kernel, grid, dim3(256), 0, s, nullptr, done, 0, (const float4*)dang, (const float4*)G, (const float4*)dV,
                          edge_offset, in_off, in_edge, (float4*)dC, (float4*)dP1, (float4*)dP3, n_atom);
  else
    hipLaunchKernelGGL(atom_gather3_kernel, grid, dim3(256), 0, s, (const float4*)dang, (const float4*)G, (const float4*)dV, edge_offset, in_off,
                       in_edge, (float4*)dC, (float4*)dP1, (float4*)dP3, n_atom);
}

}  // namespace scann
