// Training-path kernels (backward pass, loss, Adam) for the SCANN+ graph -- SURVEY.md section 8 row a17.
//
// Reference semantics: model.compile(loss=root_mean_squared_error, optimizer=Adam(lr, decay=1e-5)) + model.fit
// (scann_model.py:199-241); loss = sqrt(mean((y_pred - y_true)^2)) (losses.py:5-6) + sum of l2(1e-4) kernel
// regularisers (attention.py:27-28,95-109,260-265; scann_model.py:428,441).  Keras differentiates the forward graph
// of create_model; here the backward is written by hand as a sequence of small kernels over packed [rows,128]
// tensors (first version: modular, activations recomputed from the per-layer tensors the training forward keeps).
// Every formula is the exact derivative of the corresponding forward line cited next to it.
#include "scann_internal.h"
#include <algorithm>
#include <vector>

#include "scann_train.h"
#include "scann_mma.h"

namespace scann {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float swish_(float x) { return x * sigmoidf_(x); }
// d/dx [x * sigmoid(x)] = s * (1 + x * (1 - s))
__device__ __forceinline__ float dswish_(float x) {
  const float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ int acc_row_(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// ---- linear: Y = act(X . W + b) on 64-row tiles (same MFMA fragment scheme as the forward kernels) -----------------
// flags: bit0 accumulate into Y, bit1 swish, bit2 multiply by swish'(P) (P is then an INPUT: the fused swish backward).
// Otherwise P (optional) receives the pre-activation X.W + b.
template <int RT>  // 32-row MFMA row tiles per workgroup: 2 for large inputs, 1 otherwise (more workgroups)
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                     const float* __restrict__ bias, float* __restrict__ Y,
                                                     float* __restrict__ P, int rows, int flags) {
  constexpr int TR = 32 * RT;
  __shared__ __attribute__((aligned(16))) float sX[TR * LDS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * TR;
  const int nrows = min(TR, rows - row0);
  const float4* __restrict__ wsrc = reinterpret_cast<const float4*>(Wp) + wave * (16 * 64) + lane;
  float4 w[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) w[t] = wsrc[t * 64];
  {  // all row pieces of a thread are requested together from clamped rows and masked afterwards: a load under a
     // per-thread guard costs one full memory round trip each (branch + load + s_waitcnt vmcnt(0) + store)
    const int c4 = tid & 31, r0 = tid >> 5;
    float4 xv[4 * RT];
#pragma unroll
    for (int k = 0; k < 4 * RT; ++k)
      xv[k] = reinterpret_cast<const float4*>(X)[(size_t)(row0 + min(r0 + 8 * k, nrows - 1)) * 32 + c4];
#pragma unroll
    for (int k = 0; k < 4 * RT; ++k) {
      const int r = r0 + 8 * k;
      *reinterpret_cast<float4*>(&sX[r * LDS_STRIDE + 4 * c4]) = r < nrows ? xv[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[rt][i] = 0.f;
  const float* xrow = sX + (lane & 31) * LDS_STRIDE + 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      // operands swapped (weights as A, rows as B): transposed product -- lane l holds row l & 31 of the row tile and the
      // columns 32 wave + 8 j + 4 (l >> 5) + 0..3, j = 0..3 -- so the epilogue moves 16-byte pieces (same k order, same sums)
      const float4 a = *reinterpret_cast<const float4*>(xrow + rt * 32 * LDS_STRIDE + 8 * t);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].x, a.x, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].y, a.y, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].z, a.z, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].w, a.w, acc[rt], 0, 0, 0);
    }
  const int cb = 32 * wave + 4 * (lane >> 5);  // first column of run j: cb + 8 j
  float4 bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = bias ? *reinterpret_cast<const float4*>(bias + cb + 8 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 yold[RT][4];  // accumulate mode: the old values, requested together (clamped rows) before any of them is used
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) yold[rt][j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (flags & 1) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        yold[rt][j] = *reinterpret_cast<const float4*>(Y + (size_t)(row0 + min(32 * rt + (lane & 31), nrows - 1)) * D + cb + 8 * j);
  }
  float4 preact[RT][4];  // flags & 4: pre-activation rows (requested with the other epilogue operands)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < 4; ++j) preact[rt][j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (flags & 4) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        preact[rt][j] = *reinterpret_cast<const float4*>(P + (size_t)(row0 + min(32 * rt + (lane & 31), nrows - 1)) * D + cb + 8 * j);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int r = 32 * rt + (lane & 31);
    if (r < nrows) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t o = (size_t)(row0 + r) * D + cb + 8 * j;
        float4 v = make_float4(acc[rt][4 * j] + bv[j].x, acc[rt][4 * j + 1] + bv[j].y, acc[rt][4 * j + 2] + bv[j].z, acc[rt][4 * j + 3] + bv[j].w);
        if (flags & 4) {  // P is an input here: the pre-activation whose swish this product is the gradient of
          const float4 pr = preact[rt][j];
          v = make_float4(v.x * dswish_(pr.x), v.y * dswish_(pr.y), v.z * dswish_(pr.z), v.w * dswish_(pr.w));
        } else if (P) {
          *reinterpret_cast<float4*>(P + o) = v;
        }
        if (flags & 2) v = make_float4(swish_(v.x), swish_(v.y), swish_(v.z), swish_(v.w));
        if (flags & 1) v = make_float4(v.x + yold[rt][j].x, v.y + yold[rt][j].y, v.z + yold[rt][j].z, v.w + yold[rt][j].w);
        *reinterpret_cast<float4*>(Y + o) = v;
      }
    }
  }
}

void launch_linear(const float* X, const float* Wp, const float* bias, float* Y, float* P, int rows, int flags,
                   hipStream_t s) {
  if (rows <= 0) return;
  if (rows >= 65536) hipLaunchKernelGGL(linear_kernel<2>, dim3((rows + 63) / 64), dim3(256), 0, s, X, Wp, bias, Y, P, rows, flags);
  else hipLaunchKernelGGL(linear_kernel<1>, dim3((rows + 31) / 32), dim3(256), 0, s, X, Wp, bias, Y, P, rows, flags);
}

// ---- Y (+)= sum_i X_i . W_i for up to three (X_i, W_i) pairs on 32-row tiles: the data gradients that meet in one
// tensor (d centres = dP1.W1^T + dP3.W3^T + dq.Wq^T, attention.py:142-160) in one launch and one pass over Y -----------------
struct LinearSumArgs {
  const float* X[3];
  const float* Wp[3];
  int n;
};
__global__ __launch_bounds__(256) void linear_sum_kernel(LinearSumArgs a, float* __restrict__ Y, int rows, int accumulate,
                                                         const float* __restrict__ P) {
  __shared__ __attribute__((aligned(16))) float sX[32 * LDS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * 32;
  const int nrows = min(32, rows - row0);
  const int c4 = tid & 31, r0 = tid >> 5;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int cb = 32 * wave + 4 * (lane >> 5);
  float4 yold[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) yold[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (accumulate) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      yold[j] = *reinterpret_cast<const float4*>(Y + (size_t)(row0 + min(lane & 31, nrows - 1)) * D + cb + 8 * j);
  }
  for (int p = 0; p < a.n; ++p) {
    const float4* __restrict__ wsrc = reinterpret_cast<const float4*>(a.Wp[p]) + wave * (16 * 64) + lane;
    float4 w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = wsrc[t * 64];
    float4 xv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      xv[k] = reinterpret_cast<const float4*>(a.X[p])[(size_t)(row0 + min(r0 + 8 * k, nrows - 1)) * 32 + c4];
    if (p) __syncthreads();  // every wave is done with the previous operand tile
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + 8 * k;
      *reinterpret_cast<float4*>(&sX[r * LDS_STRIDE + 4 * c4]) = r < nrows ? xv[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const float* xrow = sX + (lane & 31) * LDS_STRIDE + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float4 x4 = *reinterpret_cast<const float4*>(xrow + 8 * t);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].x, x4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].y, x4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].z, x4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].w, x4.w, acc, 0, 0, 0);
    }
  }
  if ((lane & 31) < nrows) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t o = (size_t)(row0 + (lane & 31)) * D + cb + 8 * j;
      float4 v = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
      if (P) {  // fused swish backward: the sum is d loss / d swish(P)
        const float4 pr = *reinterpret_cast<const float4*>(P + o);
        v = make_float4(v.x * dswish_(pr.x), v.y * dswish_(pr.y), v.z * dswish_(pr.z), v.w * dswish_(pr.w));
      }
      *reinterpret_cast<float4*>(Y + o) = make_float4(v.x + yold[j].x, v.y + yold[j].y, v.z + yold[j].z, v.w + yold[j].w);
    }
  }
}
void launch_linear_sum(const float* X0, const float* W0, const float* X1, const float* W1, const float* X2, const float* W2, float* Y,
                       int rows, int accumulate, hipStream_t s, const float* swish_pre) {
  if (rows <= 0) return;
  LinearSumArgs a{};
  a.X[0] = X0; a.Wp[0] = W0; a.X[1] = X1; a.Wp[1] = W1; a.X[2] = X2; a.Wp[2] = W2;
  a.n = X2 ? 3 : (X1 ? 2 : 1);
  hipLaunchKernelGGL(linear_sum_kernel, dim3((rows + 31) / 32), dim3(256), 0, s, a, Y, rows, accumulate, swish_pre);
}

// ---- weight gradient: dW[i][j] += sum_rows X[row][i] dY[row][j];  db[j] += sum_rows dY[row][j] ---------------------
// Two stages, no atomics, bit-reproducible: a workgroup reduces one slab of rows with MFMA (A = X^T read column-wise from
// LDS) and STORES its 128x128 partial (and its bias partial) to a slot of its own; wgrad_reduce_kernel, launched once at the
// end of the backward pass for ALL the weight gradients of the step, adds the slots of every gradient in slab order.
struct WgradSet {  // up to WGRAD_MAX_JOBS independent gradients in one launch (blockIdx.y)
  const float* X[WGRAD_MAX_JOBS];
  const float* dY[WGRAD_MAX_JOBS];
  float* part[WGRAD_MAX_JOBS];   // [n_slab][128*128] partial sums of the job
  float* bpart[WGRAD_MAX_JOBS];  // [n_slab][128] bias partials or null
  int32_t rows[WGRAD_MAX_JOBS], chunks[WGRAD_MAX_JOBS];
  const int32_t* nb[WGRAD_MAX_JOBS];  // gated X operand (WgradCtx::Job) or null
  const float* X2[WGRAD_MAX_JOBS];
};
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradSet set) {
  // blockIdx.y selects the job: all weight gradients of one LocalAttention / ResidualNorm layer go out in ONE launch.
  // Split-fp16 MFMA with K = rows (v_mfma_f32_32x32x16_f16, three products per operand pair, scann_kernels.hip): a lane's operand is
  // 8 consecutive ROWS of one column, so every lane fetches its operands straight from global memory in that shape -- wave w the
  // X^T fragments of features 32 w .. 32 w + 31 (shared with the other waves through LDS, fragment order, conflict-free b128) and
  // the dY fragments of its own 32 output columns (registers only).  dY is a gradient: each wave scales its dY block by a power
  // of two that brings the largest magnitude seen so far to 2^12 and rescales its accumulators (exactly) when a later chunk is
  // larger -- block floating point with a monotone exponent, so the split keeps 22 bits relative to the largest rows, which
  // dominate the sum.
  const int rows = set.rows[blockIdx.y], chunks = set.chunks[blockIdx.y];
  if (blockIdx.x * 64 * chunks >= rows) return;  // jobs over fewer rows than the largest one
  const float* __restrict__ X = set.X[blockIdx.y];
  const float* __restrict__ dY = set.dY[blockIdx.y];
  const int32_t* __restrict__ nb = set.nb[blockIdx.y];
  const float* __restrict__ X2 = set.X2[blockIdx.y];
  float* __restrict__ part = set.part[blockIdx.y] + (size_t)blockIdx.x * D * D;
  float* __restrict__ bpart = set.bpart[blockIdx.y];
  __shared__ f16x8 sA[4][4][2][64];  // [k-step][feature tile][hi | lo][lane]: the chunk's X^T fragments (32 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = 32 * wave + (lane & 31), kh = lane >> 5;
  const int slab0 = blockIdx.x * 64 * chunks;
  f32x16 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
  float bsum = 0.f;  // column `col` of db over this lane's rows
  float C = 0.f;     // the wave's dY scale (a power of two; 0 = nothing but zeros so far)
  // operands of a chunk: 8 consecutive rows x this lane's column, for each of the 4 k-steps; clamped rows, masked afterwards, all 64
  // loads in flight together -- and requested one chunk ahead, under the MFMAs of the current chunk
  float xa[4][8], db[4][8];
  const unsigned voff = ((unsigned)(8 * kh) * D + col) * 4;  // byte offset of (row 8 kh, this lane's column) inside a k-step's 16 rows
  auto fetch_full = [&](int row0) {  // uniform base + one offset register + immediates (saddr form): no per-load address registers
    if (nb) {  // gated operand: X[nb[r]] * X2[r], the product the forward formed (one rounding, never contracted)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const int r0 = row0 + 16 * s2 + 8 * kh;
        const int4 i0 = *reinterpret_cast<const int4*>(nb + r0), i1 = *reinterpret_cast<const int4*>(nb + r0 + 4);
        const int idx[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
        const char* gs = reinterpret_cast<const char*>(X2 + (size_t)(row0 + 16 * s2) * D);
        const char* ds = reinterpret_cast<const char*>(dY + (size_t)(row0 + 16 * s2) * D);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xa[s2][j] = __fmul_rn(X[(size_t)idx[j] * D + col], *reinterpret_cast<const float*>(gs + voff + j * D * 4));
          db[s2][j] = *reinterpret_cast<const float*>(ds + voff + j * D * 4);
        }
      }
      return;
    }
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const char* xs = reinterpret_cast<const char*>(X + (size_t)(row0 + 16 * s2) * D);
      const char* ds = reinterpret_cast<const char*>(dY + (size_t)(row0 + 16 * s2) * D);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xa[s2][j] = *reinterpret_cast<const float*>(xs + voff + j * D * 4);
        db[s2][j] = *reinterpret_cast<const float*>(ds + voff + j * D * 4);
      }
    }
  };
  // one chunk: scale bookkeeping, X^T fragments -> LDS, dY fragments -> registers, [next chunk's loads], MFMAs
  auto consume = [&](int nrows, bool prefetch, int next_row0) {
    float mx = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool live = 16 * s2 + 8 * kh + j < nrows;
        xa[s2][j] = live ? xa[s2][j] : 0.f;
        db[s2][j] = live ? db[s2][j] : 0.f;
        bsum += db[s2][j];
        mx = fmaxf(mx, fabsf(db[s2][j]));
      }
    mx = wave_max64(mx);
    {
      const int e = (__float_as_int(mx) >> 23) & 0xff;
      if (e != 0) {
        const float cn = __int_as_float(min(253, 266 - e) << 23);  // 2^(12 - exponent of the largest magnitude)
        if (C == 0.f) {
          C = cn;
        } else if (cn < C) {  // larger values than any chunk before: bring the accumulators to the new scale (exact)
          const float f = cn / C;
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][i] *= f;
          C = cn;
        }
      }
    }
    const float cs = C == 0.f ? 1.0f : C;
    __syncthreads();  // every wave is done reading the previous chunk's fragments
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      f16x8 h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        h[j] = (_Float16)xa[s2][j];
        l[j] = (_Float16)(xa[s2][j] - (float)h[j]);
      }
      sA[s2][wave][0][lane] = h;
      sA[s2][wave][1][lane] = l;
    }
    f16x8 bh[4], bl[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = db[s2][j] * cs;
        bh[s2][j] = (_Float16)v;
        bl[s2][j] = (_Float16)(v - (float)bh[s2][j]);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (prefetch) fetch_full(next_row0);  // the next chunk's operands arrive under the MFMAs below
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f16x8 ah = sA[s2][m][0][lane], al = sA[s2][m][1][lane];
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s2], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s2], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s2], acc[m], 0, 0, 0);
      }
    }
  };
  const int slab_rows = min(64 * chunks, rows - slab0), n_full = slab_rows >> 6, rem = slab_rows & 63;
  if (n_full > 0) fetch_full(slab0);
  for (int c = 0; c < n_full; ++c) consume(64, c + 1 < n_full, slab0 + 64 * (c + 1));
  if (rem) {  // the ragged last chunk of the tensor: rows clamped one by one, not pipelined
    const int row0 = slab0 + 64 * n_full;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t o = (size_t)(row0 + min(16 * s2 + 8 * kh + j, rem - 1)) * D + col;
        xa[s2][j] = nb ? __fmul_rn(X[(size_t)nb[o / D] * D + col], X2[o]) : X[o];
        db[s2][j] = dY[o];
      }
    consume(rem, false, 0);
  }
  const float inv = C == 0.f ? 0.f : 1.0f / C;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) part[(size_t)(32 * m + acc_row_(i, lane)) * D + col] = acc[m][i] * inv;
  bsum = xor32(bsum);
  if (bpart && kh == 0) bpart[(size_t)blockIdx.x * D + col] = bsum;
}

// dst[i] += part[0][i] + part[1][i] + ... in slab order, for every gradient tensor of the step (blockIdx.y)
struct WgradReduceSet {
  WgradReduceEntry e[WGRAD_REDUCE_MAX];
};
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceSet set) {
  const WgradReduceEntry e = set.e[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= e.numel) return;
  const float* __restrict__ p = e.part + idx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four interleaved chains (fixed association), independent loads in flight
  int k = 0;
  for (; k + 4 <= e.n_slab; k += 4) {
    s0 += p[(size_t)k * e.numel];
    s1 += p[(size_t)(k + 1) * e.numel];
    s2 += p[(size_t)(k + 2) * e.numel];
    s3 += p[(size_t)(k + 3) * e.numel];
  }
  for (; k < e.n_slab; ++k) s0 += p[(size_t)k * e.numel];
  e.dst[idx] += (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(WgradReduceSet set) {
  constexpr int U = 8;  // independent 16-byte loads in flight
  const WgradReduceEntry e = set.e[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;  // float4 index (numel is 128 or 128 * 128)
  if (idx * 4 >= e.numel) return;
  const size_t stride = (size_t)(e.numel >> 2);
  const float4* __restrict__ p = reinterpret_cast<const float4*>(e.part) + idx;
  float4 s[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) s[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + U <= e.n_slab; k += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[(size_t)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s[u & 3].x += v[u].x; s[u & 3].y += v[u].y; s[u & 3].z += v[u].z; s[u & 3].w += v[u].w;
    }
  }
  for (; k < e.n_slab; ++k) {
    const float4 v = p[(size_t)k * stride];
    s[k & 3].x += v.x; s[k & 3].y += v.y; s[k & 3].z += v.z; s[k & 3].w += v.w;
  }
  const float4 t = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                               (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
  float* __restrict__ d = e.dst + 4 * (size_t)idx;
  if ((reinterpret_cast<size_t>(e.dst) & 15) == 0) {
    const float4 o = *reinterpret_cast<const float4*>(d);
    *reinterpret_cast<float4*>(d) = make_float4(o.x + t.x, o.y + t.y, o.z + t.z, o.w + t.w);
  } else {
    d[0] += t.x; d[1] += t.y; d[2] += t.z; d[3] += t.w;
  }
}

int wgrad_chunks(int rows) {
  // slab per workgroup: at least 4 chunks of 64 rows and <= 80 slabs per gradient (a layer's launch is ~7 gradients: a few hundred
  // workgroups), so that the partial slots (64 KB each) the reduce re-reads stay small whatever the batch size.  Measured at batch
  // 128 (profiles/r02_notes.md): 1 -> 4 chunks for the atom-row gradients, 36 -> 9 slabs each, -20 us per step.
  const int tiles = (rows + 63) / 64;
  return std::max(std::min(4, tiles), (tiles + 79) / 80);
}
int wgrad_slabs(int rows) {
  const int chunks = wgrad_chunks(rows);
  return (rows + 64 * chunks - 1) / (64 * chunks);
}

// the 128-wide gradient vectors (biases, LayerNorm gamma / beta): one workgroup of 1,024 threads per vector, eight groups of 128
// walking the slots k = group, group + 8, ... (four loads in flight each), then the eight group sums added in group order --
// the slot count of these is the WORKGROUP count of the producing kernel (hundreds), far more than the <= 80 slabs of a matrix
__global__ __launch_bounds__(1024) void vec_reduce_kernel(WgradReduceSet set) {
  const WgradReduceEntry e = set.e[blockIdx.x];
  __shared__ float sg[8][D];
  const int col = threadIdx.x & (D - 1), grp = threadIdx.x >> 7;
  const float* __restrict__ p = e.part + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = grp;
  for (; k + 24 < e.n_slab; k += 32) {
    s0 += p[(size_t)k * D];
    s1 += p[(size_t)(k + 8) * D];
    s2 += p[(size_t)(k + 16) * D];
    s3 += p[(size_t)(k + 24) * D];
  }
  for (; k < e.n_slab; k += 8) s0 += p[(size_t)k * D];
  sg[grp][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x < D) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) t += sg[g2][col];
    e.dst[col] += t;
  }
}

// both of the above in ONE launch (a layer's flush at batch 128: 7 matrices + <= 17 vectors): blocks [0, 16 n_mat) take the matrices
// (1,024 elements each, wgrad_reduce_kernel's sum), the rest a vector each (vec_reduce_kernel's sum) -- the two used to be two
// latency-bound launches one behind the other on the side stream, ~11 us each, and the step ends when the side stream does
__global__ __launch_bounds__(1024) void wgrad_reduce_all_kernel(WgradReduceSet set, int n_mat) {
  __shared__ float sg[8][D];
  const int b = blockIdx.x;
  if (b < 16 * n_mat) {
    const WgradReduceEntry e = set.e[b >> 4];
    const int idx = (b & 15) * 1024 + threadIdx.x;
    if (idx >= e.numel) return;
    const float* __restrict__ p = e.part + idx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= e.n_slab; k += 4) {
      s0 += p[(size_t)k * e.numel];
      s1 += p[(size_t)(k + 1) * e.numel];
      s2 += p[(size_t)(k + 2) * e.numel];
      s3 += p[(size_t)(k + 3) * e.numel];
    }
    for (; k < e.n_slab; ++k) s0 += p[(size_t)k * e.numel];
    e.dst[idx] += (s0 + s1) + (s2 + s3);
    return;
  }
  const WgradReduceEntry e = set.e[n_mat + (b - 16 * n_mat)];
  const int col = threadIdx.x & (D - 1), grp = threadIdx.x >> 7;
  const float* __restrict__ p = e.part + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = grp;
  for (; k + 24 < e.n_slab; k += 32) {
    s0 += p[(size_t)k * D];
    s1 += p[(size_t)(k + 8) * D];
    s2 += p[(size_t)(k + 16) * D];
    s3 += p[(size_t)(k + 24) * D];
  }
  for (; k < e.n_slab; k += 8) s0 += p[(size_t)k * D];
  sg[grp][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x < D) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) t += sg[g2][col];
    e.dst[col] += t;
  }
}

void wgrad_flush(WgradCtx& ctx, hipStream_t s) {
  // the (destination, slots) records travel as kernel arguments: no table copy, so a flush can follow each layer's gradient
  // launch on the side stream
  std::vector<WgradReduceEntry> mats, vecs;
  for (const WgradReduceEntry& e : ctx.entries) (e.numel == D ? vecs : mats).push_back(e);
  {
    size_t bytes = 0;
    bool fits = !mats.empty() && !vecs.empty() && mats.size() + vecs.size() <= (size_t)WGRAD_REDUCE_MAX;
    for (const WgradReduceEntry& e : mats) {
      bytes += (size_t)e.n_slab * e.numel * 4;
      fits = fits && e.numel <= 16 * 1024;
    }
    if (fits && bytes < ((size_t)24 << 20)) {  // (beyond that the 16-byte-load shape of the matrix sum pays: below)
      WgradReduceSet set{};
      int n = 0;
      for (const WgradReduceEntry& e : mats) set.e[n++] = e;
      for (const WgradReduceEntry& e : vecs) set.e[n++] = e;
      hipLaunchKernelGGL(wgrad_reduce_all_kernel, dim3(16 * (unsigned)mats.size() + (unsigned)vecs.size()), dim3(1024), 0, s, set, (int)mats.size());
      ctx.entries.clear();
      return;
    }
  }
  for (size_t e0 = 0; e0 < mats.size(); e0 += WGRAD_REDUCE_MAX) {
    const int n = (int)std::min<size_t>(WGRAD_REDUCE_MAX, mats.size() - e0);
    WgradReduceSet set{};
    for (int k = 0; k < n; ++k) set.e[k] = mats[e0 + k];
    // two shapes of the same fixed-order sum: many light threads while the launch is a latency chain beside the data-gradient
    // kernels (batch 128: 1.18 vs 1.19 ms per step), 16-byte loads with eight in flight once it is bandwidth that counts
    // (batch 1024: 4.81 vs 5.36 ms)
    size_t bytes = 0;
    for (int k = 0; k < n; ++k) bytes += (size_t)set.e[k].n_slab * set.e[k].numel * 4;
    if (bytes >= ((size_t)24 << 20)) hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(D * D / 4 / 256, n), dim3(256), 0, s, set);
    else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(D * D / 256, n), dim3(256), 0, s, set);
  }
  for (size_t e0 = 0; e0 < vecs.size(); e0 += WGRAD_REDUCE_MAX) {
    const int n = (int)std::min<size_t>(WGRAD_REDUCE_MAX, vecs.size() - e0);
    WgradReduceSet set{};
    for (int k = 0; k < n; ++k) set.e[k] = vecs[e0 + k];
    hipLaunchKernelGGL(vec_reduce_kernel, dim3(n), dim3(1024), 0, s, set);
  }
  ctx.entries.clear();
}

// queue one gradient dW += X^T dY (db += column sums of dY when db) for the next wgrad_launch
void wgrad_add(WgradCtx& ctx, const float* X, const float* dY, float* dW, float* db, int rows, const int32_t* nb, const float* X2) {
  if (rows <= 0) return;
  WgradCtx::Job j{};
  j.X = X; j.dY = dY; j.rows = rows; j.chunks = wgrad_chunks(rows);
  j.nb = nb; j.X2 = X2;
  const int n_slab = wgrad_slabs(rows);
  j.part = ctx.arena + ctx.off;
  ctx.entries.push_back(WgradReduceEntry{dW, j.part, n_slab, D * D});
  ctx.off += (size_t)n_slab * D * D;
  if (db) {
    j.bpart = ctx.arena + ctx.off;
    ctx.entries.push_back(WgradReduceEntry{db, j.bpart, n_slab, D});
    ctx.off += (size_t)n_slab * D;
  }
  ctx.jobs.push_back(j);
}
// one launch (per WGRAD_MAX_JOBS) for everything queued
void wgrad_launch(WgradCtx& ctx, hipStream_t s) {
  for (size_t j0 = 0; j0 < ctx.jobs.size(); j0 += WGRAD_MAX_JOBS) {
    const int n = (int)std::min<size_t>(WGRAD_MAX_JOBS, ctx.jobs.size() - j0);
    WgradSet set{};
    int max_slab = 0;
    for (int k = 0; k < n; ++k) {
      const WgradCtx::Job& j = ctx.jobs[j0 + k];
      set.X[k] = j.X; set.dY[k] = j.dY; set.part[k] = j.part; set.bpart[k] = j.bpart; set.rows[k] = j.rows; set.chunks[k] = j.chunks;
      set.nb[k] = j.nb; set.X2[k] = j.X2;
      max_slab = std::max(max_slab, wgrad_slabs(j.rows));
    }
    hipLaunchKernelGGL(wgrad_kernel, dim3(max_slab, n), dim3(256), 0, s, set);
  }
  ctx.jobs.clear();
}
// ---- elementwise ---------------------------------------------------------------------------------------------------

__global__ void swish_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dout,
                                 float* __restrict__ dpre, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dpre[i] = dout[i] * dswish_(pre[i]);
}
void launch_swish_bwd(const float* pre, const float* dout, float* dpre, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(swish_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, dout, dpre, n);
}

__global__ void dropout_kernel(float* __restrict__ x, size_t n, unsigned long long seed, unsigned tag, float p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= drop_scale(seed, tag, i, p);
}
// dst = src * mask: the Dropout backward into a buffer of its own (the un-dropped gradient is still needed)
__global__ void dropout_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n, unsigned long long seed, unsigned tag,
                                    float p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] * drop_scale(seed, tag, i, p);
}
void launch_dropout_copy(float* dst, const float* src, size_t n, unsigned long long seed, unsigned tag, float p, hipStream_t s) {
  if (n) hipLaunchKernelGGL(dropout_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, src, n, seed, tag, p);
}
void launch_dropout(float* x, size_t n, unsigned long long seed, unsigned tag, float p, hipStream_t s) {
  if (n && p > 0.f) hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, seed, tag, p);
}

// ---- LayerNorm backward (rows of 128; 8 threads per row) ---------------------------------------------------------------
// y = xhat*gamma + beta, xhat = (x - mean) * rstd  (attention.py:35,111,113; eps 1e-6)
// dxhat = dy*gamma ; dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat*xhat)) ; dgamma += dy*xhat ; dbeta += dy
// Optional fusions of the LocalAttention backward (null when unused): the incoming gradient is formed on the fly as
// dy = dang * c[nb] + dg_in (the gate ang = c[j] * G', attention.py:157, plus the geometry gradient of the next layer), and
// dV = dx * swish'(V) (attention.py:151) leaves with dx.
struct LnBwdFuse {
  const float* dang;   // [rows,128] or null: dy is read as is
  const float* c;      // [n_atom,128]
  const int* nb;       // [rows]
  const float* dg_in;  // [rows,128] or null
  const float* V;      // [rows,128] or null
  float* dV;           // [rows,128]
};
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ dy, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                                     int groups, int accumulate, LnBwdFuse f) {
  // 8 threads per row (float4 chunks sub, sub+8, sub+16, sub+24 like the forward LayerNorm), 32 rows per pass, `groups`
  // passes per workgroup; all loads of a pass are requested together from clamped rows.
  const int tid = threadIdx.x, r_in = tid >> 3, sub = tid & 7;
  const int row_base = blockIdx.x * 32 * groups;
  float4 g4[4], dg[4], dbt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    g4[i] = reinterpret_cast<const float4*>(gamma)[sub + 8 * i];
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    dbt[i] = dg[i];
  }
  for (int gi = 0; gi < groups; ++gi) {
    const int r = row_base + 32 * gi + r_in;
    if (row_base + 32 * gi >= rows) break;
    const int rc = min(r, rows - 1);
    float4 xv[4], dv[4], old[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xv[i] = reinterpret_cast<const float4*>(x)[(size_t)rc * 32 + sub + 8 * i];
      dv[i] = reinterpret_cast<const float4*>(f.dang ? f.dang : dy)[(size_t)rc * 32 + sub + 8 * i];
      old[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (f.dang) {
      const int j = f.nb[rc];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 cn = reinterpret_cast<const float4*>(f.c)[(size_t)j * 32 + sub + 8 * i];
        dv[i] = make_float4(dv[i].x * cn.x, dv[i].y * cn.y, dv[i].z * cn.z, dv[i].w * cn.w);
        if (f.dg_in) {
          const float4 o = reinterpret_cast<const float4*>(f.dg_in)[(size_t)rc * 32 + sub + 8 * i];
          dv[i].x += o.x; dv[i].y += o.y; dv[i].z += o.z; dv[i].w += o.w;
        }
      }
    }
    if (accumulate) {
#pragma unroll
      for (int i = 0; i < 4; ++i) old[i] = reinterpret_cast<const float4*>(dx)[(size_t)rc * 32 + sub + 8 * i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    const float mean = s * (1.0f / D);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;
      v += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
    }
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
    float m1 = 0.f, m2 = 0.f;
    float4 ax[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xv[i].x *= rstd; xv[i].y *= rstd; xv[i].z *= rstd; xv[i].w *= rstd;  // xhat
      ax[i] = make_float4(dv[i].x * g4[i].x, dv[i].y * g4[i].y, dv[i].z * g4[i].z, dv[i].w * g4[i].w);
      m1 += (ax[i].x + ax[i].y) + (ax[i].z + ax[i].w);
      m2 += (ax[i].x * xv[i].x + ax[i].y * xv[i].y) + (ax[i].z * xv[i].z + ax[i].w * xv[i].w);
    }
    m1 += __shfl_xor(m1, 1); m1 += __shfl_xor(m1, 2); m1 += __shfl_xor(m1, 4);
    m2 += __shfl_xor(m2, 1); m2 += __shfl_xor(m2, 2); m2 += __shfl_xor(m2, 4);
    m1 *= (1.0f / D);
    m2 *= (1.0f / D);
    if (r < rows) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 dxv = make_float4(rstd * (ax[i].x - m1 - xv[i].x * m2) + old[i].x, rstd * (ax[i].y - m1 - xv[i].y * m2) + old[i].y,
                                       rstd * (ax[i].z - m1 - xv[i].z * m2) + old[i].z, rstd * (ax[i].w - m1 - xv[i].w * m2) + old[i].w);
        reinterpret_cast<float4*>(dx)[(size_t)r * 32 + sub + 8 * i] = dxv;
        if (f.V) {
          const float4 vv = reinterpret_cast<const float4*>(f.V)[(size_t)r * 32 + sub + 8 * i];
          reinterpret_cast<float4*>(f.dV)[(size_t)r * 32 + sub + 8 * i] =
              make_float4(dxv.x * dswish_(vv.x), dxv.y * dswish_(vv.y), dxv.z * dswish_(vv.z), dxv.w * dswish_(vv.w));
        }
        dg[i].x += dv[i].x * xv[i].x; dg[i].y += dv[i].y * xv[i].y; dg[i].z += dv[i].z * xv[i].z; dg[i].w += dv[i].w * xv[i].w;
        dbt[i].x += dv[i].x; dbt[i].y += dv[i].y; dbt[i].z += dv[i].z; dbt[i].w += dv[i].w;
      }
    }
  }
  // column sums over the 32 row slots of the workgroup, then one set of atomics per workgroup
  __shared__ float sred[2][32][D + 4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<float4*>(&sred[0][r_in][4 * (sub + 8 * i)]) = dg[i];
    *reinterpret_cast<float4*>(&sred[1][r_in][4 * (sub + 8 * i)]) = dbt[i];
  }
  __syncthreads();
  const int which = tid >> 7, col = tid & (D - 1);
  float tot = 0.f;
#pragma unroll 8
  for (int rr = 0; rr < 32; ++rr) tot += sred[which][rr][col];
  (which ? dbeta : dgamma)[(size_t)blockIdx.x * D + col] = tot;  // this workgroup's slot (summed in slot order by wgrad_reduce_kernel)
}
void wgrad_flush(WgradCtx& ctx, hipStream_t s);
// row groups of 32 per workgroup: with per-workgroup partial slots instead of atomics the launch no longer has to stay small
// (round 1 capped it at ~160 workgroups); about 1,500 workgroups fill the chip several times over
static int ln_bwd_groups(int rows) { return std::min(32, std::max(1, (rows + 32 * 1536 - 1) / (32 * 1536))); }
int ln_bwd_slots(int rows) {
  const int groups = ln_bwd_groups(rows);
  return (rows + 32 * groups - 1) / (32 * groups);
}
// records (dst, slots) of one parameter-gradient vector whose per-workgroup partial sums a kernel is about to store
float* reserve_vec(WgradCtx& ctx, float* dst, int n_slot) {
  float* part = ctx.arena + ctx.off;
  ctx.entries.push_back(WgradReduceEntry{dst, part, n_slot, D});
  ctx.off += (size_t)n_slot * D;
  return part;
}
void launch_ln_bwd(WgradCtx& ctx, const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, int rows,
                   int accumulate, hipStream_t s) {
  if (rows <= 0) return;
  const int groups = ln_bwd_groups(rows), n_wg = ln_bwd_slots(rows);
  float* gp = reserve_vec(ctx, dgamma, n_wg);
  float* bp = reserve_vec(ctx, dbeta, n_wg);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(n_wg), dim3(256), 0, s, x, gamma, dy, dx, gp, bp, rows, groups, accumulate, LnBwdFuse{});
}
// LayerNorm_g backward of LocalAttention with its neighbours fused: dy = dang * c[nb] (+ dg_in) in, dx = dT and dV = dT * swish'(V) out
void launch_ln_bwd_edge(WgradCtx& ctx, const float* T, const float* gamma, const float* dang, const float* c, const int* nb,
                        const float* dg_in, const float* V, float* dT, float* dV, float* dgamma, float* dbeta, int rows, hipStream_t s) {
  if (rows <= 0) return;
  const int groups = ln_bwd_groups(rows), n_wg = ln_bwd_slots(rows);
  float* gp = reserve_vec(ctx, dgamma, n_wg);
  float* bp = reserve_vec(ctx, dbeta, n_wg);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(n_wg), dim3(256), 0, s, T, gamma, (const float*)nullptr, dT, gp, bp, rows, groups, 0,
                     LnBwdFuse{dang, c, nb, dg_in, V, dV});
}

// ---- edge elementwise kernels (thread = float4 chunk of an edge row) -------------------------------------------------

// backward of ang = cn * G':  dcn (per edge; summed over the edges that point AT an atom by gather_sum_kernel) = dang * G',
// dG'(total) = dang * cn + dG'(from the next layer)
__global__ void edge_dang_kernel(const float4* __restrict__ c, const int* __restrict__ nb, const float4* __restrict__ g,
                                 const float4* __restrict__ dang, const float4* __restrict__ dg_in,
                                 float4* __restrict__ dcn, float4* __restrict__ dg_tot, int n_edge) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_edge * 32) return;
  const int e = (int)(i >> 5), c4 = (int)(i & 31);
  const int j = nb[e];
  const float4 cn = c[(size_t)j * 32 + c4], gg = g[i], da = dang[i];
  float4 t = make_float4(da.x * cn.x, da.y * cn.y, da.z * cn.z, da.w * cn.w);
  if (dg_in) {
    const float4 o = dg_in[i];
    t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
  }
  dg_tot[i] = t;
  dcn[i] = make_float4(da.x * gg.x, da.y * gg.y, da.z * gg.z, da.w * gg.w);
}
// out[a] (+)= sum of val[e] over the edges e whose NEIGHBOUR is atom a (reverse adjacency built at upload: in_off, in_edge):
// the neighbour-indexed sums of the backward pass in a fixed order, without float atomics
__global__ void gather_sum_kernel(const float4* __restrict__ val, const int* __restrict__ in_off, const int* __restrict__ in_edge,
                                  float4* __restrict__ out, int n_atom, int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_atom * 32) return;
  const int a = (int)(i >> 5), c4 = (int)(i & 31);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (accumulate) s = out[i];
  for (int k = in_off[a]; k < in_off[a + 1]; ++k) {
    const float4 v = val[(size_t)in_edge[k] * 32 + c4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  out[i] = s;
}
// out[a] (+)= sum over the edges e that point AT atom a of x[e] * y[e]: the neighbour-centre gradient dC[j] = sum dang * G' without
// materialising the per-edge product
__global__ void gather_prod_sum_kernel(const float4* __restrict__ x, const float4* __restrict__ y, const int* __restrict__ in_off,
                                       const int* __restrict__ in_edge, float4* __restrict__ out, int n_atom, int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_atom * 32) return;
  const int a = (int)(i >> 5), c4 = (int)(i & 31);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (accumulate) s = out[i];
  for (int k = in_off[a]; k < in_off[a + 1]; ++k) {
    const size_t o = (size_t)in_edge[k] * 32 + c4;
    const float4 u = x[o], v = y[o];
    s.x += u.x * v.x; s.y += u.y * v.y; s.z += u.z * v.z; s.w += u.w * v.w;
  }
  out[i] = s;
}
void launch_gather_prod_sum(const float* x, const float* y, const int* in_off, const int* in_edge, float* out, int n_atom, int accumulate,
                            hipStream_t s) {
  if (n_atom > 0)
    hipLaunchKernelGGL(gather_prod_sum_kernel, dim3((unsigned)(((size_t)n_atom * 32 + 255) / 256)), dim3(256), 0, s, (const float4*)x,
                       (const float4*)y, in_off, in_edge, (float4*)out, n_atom, accumulate);
}
// both atom-indexed sums of dV in one launch: own[a] = sum over atom a's CSR row (dP1), in[a] = sum over the edges that point at a (dP3)
__global__ void atom_sums_kernel(const float4* __restrict__ dV, const int* __restrict__ edge_offset, const int* __restrict__ in_off,
                                 const int* __restrict__ in_edge, float4* __restrict__ own, float4* __restrict__ in, int n_atom) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_atom * 32) return;
  const int a = (int)(i >> 5), c4 = (int)(i & 31);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), t = s;
  for (int e = edge_offset[a]; e < edge_offset[a + 1]; ++e) {
    const float4 v = dV[(size_t)e * 32 + c4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  for (int k = in_off[a]; k < in_off[a + 1]; ++k) {
    const float4 v = dV[(size_t)in_edge[k] * 32 + c4];
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
  }
  own[i] = s;
  in[i] = t;
}
void launch_atom_sums(const float* dV, const int* edge_offset, const int* in_off, const int* in_edge, float* own, float* in, int n_atom,
                      hipStream_t s) {
  if (n_atom > 0)
    hipLaunchKernelGGL(atom_sums_kernel, dim3((unsigned)(((size_t)n_atom * 32 + 255) / 256)), dim3(256), 0, s, (const float4*)dV,
                       edge_offset, in_off, in_edge, (float4*)own, (float4*)in, n_atom);
}
void launch_gather_sum(const float* val, const int* in_off, const int* in_edge, float* out, int n_atom, int accumulate, hipStream_t s) {
  if (n_atom > 0)
    hipLaunchKernelGGL(gather_sum_kernel, dim3((unsigned)(((size_t)n_atom * 32 + 255) / 256)), dim3(256), 0, s, (const float4*)val,
                       in_off, in_edge, (float4*)out, n_atom, accumulate);
}
#define EDGE_GRID(n_edge) dim3((unsigned)(((size_t)(n_edge) * 32 + 255) / 256)), dim3(256)
void launch_edge_dang(const float* c, const int* nb, const float* g, const float* dang, const float* dg_in, float* dcn,
                      float* dg_tot, int n_edge, hipStream_t s) {
  if (n_edge > 0)
    hipLaunchKernelGGL(edge_dang_kernel, EDGE_GRID(n_edge), 0, s, (const float4*)c, nb, (const float4*)g,
                       (const float4*)dang, (const float4*)dg_in, (float4*)dcn, (float4*)dg_tot, n_edge);
}
// ---- attention backward: one wave per atom, lane l owns features 2l, 2l+1 (head l>>3) -----------------------------
// forward (attention.py:180-214): e[n,h] = 0.25 sum_d q[h,d] K[n,h,d]; attn = softmax_n(e); pre = sum_n attn K[n] + q;
// ctx = LN(pre).  Given dctx: dpre = LN'(dctx); dq += dpre; dattn[n,h] = sum_d dpre[h,d] K[n,h,d];
// de = attn*(dattn - sum_m attn[m] dattn[m]); dK[n] = attn*dpre + 0.25*de*q ; dq[h,:] += 0.25 sum_n de[n,h] K[n,h,:].
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ K,
                                                       const int* __restrict__ edge_offset,
                                                       const float* __restrict__ dctx, const float* __restrict__ gamma,
                                                       float* __restrict__ dq, float* __restrict__ dK,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int n_atom,
                                                       int atoms_per_wave, float drop_p, unsigned drop_tag,
                                                       unsigned long long drop_seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a0 = (blockIdx.x * 4 + wave) * atoms_per_wave;
  const float2 g = reinterpret_cast<const float2*>(gamma)[lane];
  float2 dg = make_float2(0.f, 0.f), dbt = dg;
  for (int at = a0; at < min(n_atom, a0 + atoms_per_wave); ++at) {
    const int e0 = edge_offset[at], e1 = edge_offset[at + 1];
    const float2 q2 = reinterpret_cast<const float2*>(q)[(size_t)at * 64 + lane];
    const float qx = q2.x * 0.25f, qy = q2.y * 0.25f;
    float m = -INFINITY;
    for (int n = e0; n < e1; ++n) {
      const float2 k2 = reinterpret_cast<const float2*>(K)[(size_t)n * 64 + lane];
      float e = qx * k2.x + qy * k2.y;
      e += __shfl_xor(e, 1); e += __shfl_xor(e, 2); e += __shfl_xor(e, 4);
      m = fmaxf(m, e);
    }
    float ssum = 0.f;
    for (int n = e0; n < e1; ++n) {
      const float2 k2 = reinterpret_cast<const float2*>(K)[(size_t)n * 64 + lane];
      float e = qx * k2.x + qy * k2.y;
      e += __shfl_xor(e, 1); e += __shfl_xor(e, 2); e += __shfl_xor(e, 4);
      ssum += expf(e - m);
    }
    float px = 0.f, py = 0.f;  // pre = sum attn K + q
    for (int n = e0; n < e1; ++n) {
      const float2 k2 = reinterpret_cast<const float2*>(K)[(size_t)n * 64 + lane];
      float e = qx * k2.x + qy * k2.y;
      e += __shfl_xor(e, 1); e += __shfl_xor(e, 2); e += __shfl_xor(e, 4);
      float attn = expf(e - m) / ssum;
      if (drop_p > 0.f) attn *= drop_scale(drop_seed, drop_tag, (size_t)n * NHEAD + (lane >> 3), drop_p);
      px += attn * k2.x; py += attn * k2.y;
    }
    px += q2.x; py += q2.y;
    // LayerNorm backward
    float s = px + py;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / D);
    const float cx = px - mean, cy = py - mean;
    float v = cx * cx + cy * cy;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
    const float hx = cx * rstd, hy = cy * rstd;
    const float2 dyv = reinterpret_cast<const float2*>(dctx)[(size_t)at * 64 + lane];
    const float ax = dyv.x * g.x, ay = dyv.y * g.y;
    float m1 = ax + ay, m2 = ax * hx + ay * hy;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      m1 += __shfl_xor(m1, o);
      m2 += __shfl_xor(m2, o);
    }
    m1 *= (1.0f / D); m2 *= (1.0f / D);
    const float dpx = rstd * (ax - m1 - hx * m2), dpy = rstd * (ay - m1 - hy * m2);
    dg.x += dyv.x * hx; dg.y += dyv.y * hy; dbt.x += dyv.x; dbt.y += dyv.y;
    // softmax backward: dot = sum_m attn[m] dattn[m] per head
    float dot = 0.f;
    for (int n = e0; n < e1; ++n) {
      const float2 k2 = reinterpret_cast<const float2*>(K)[(size_t)n * 64 + lane];
      float e = qx * k2.x + qy * k2.y, da = dpx * k2.x + dpy * k2.y;
      e += __shfl_xor(e, 1); e += __shfl_xor(e, 2); e += __shfl_xor(e, 4);
      da += __shfl_xor(da, 1); da += __shfl_xor(da, 2); da += __shfl_xor(da, 4);
      if (drop_p > 0.f) da *= drop_scale(drop_seed, drop_tag, (size_t)n * NHEAD + (lane >> 3), drop_p);  // d(attn) through the mask
      dot += (expf(e - m) / ssum) * da;
    }
    float dqx = dpx, dqy = dpy;  // residual path (attention.py:212)
    for (int n = e0; n < e1; ++n) {
      const float2 k2 = reinterpret_cast<const float2*>(K)[(size_t)n * 64 + lane];
      float e = qx * k2.x + qy * k2.y, da = dpx * k2.x + dpy * k2.y;
      e += __shfl_xor(e, 1); e += __shfl_xor(e, 2); e += __shfl_xor(e, 4);
      da += __shfl_xor(da, 1); da += __shfl_xor(da, 2); da += __shfl_xor(da, 4);
      const float attn = expf(e - m) / ssum;
      const float keep = drop_p > 0.f ? drop_scale(drop_seed, drop_tag, (size_t)n * NHEAD + (lane >> 3), drop_p) : 1.0f;
      const float de = attn * (da * keep - dot);
      reinterpret_cast<float2*>(dK)[(size_t)n * 64 + lane] =
          make_float2(attn * keep * dpx + 0.25f * de * q2.x, attn * keep * dpy + 0.25f * de * q2.y);
      dqx += 0.25f * de * k2.x; dqy += 0.25f * de * k2.y;
    }
    reinterpret_cast<float2*>(dq)[(size_t)at * 64 + lane] = make_float2(dqx, dqy);
  }
  const size_t slot = (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * D;  // one slot per wave
  reinterpret_cast<float2*>(dgamma + slot)[lane] = dg;
  reinterpret_cast<float2*>(dbeta + slot)[lane] = dbt;
}
// Fast path when no atom of the batch has more than 16 neighbours (QM9: <= 12): the atom's key rows, logits and
// attention gradients live in registers, so K is read once and the 8-lane dot-product reductions run once per edge.
// Lane reductions of the attention backward through DPP (no LDS round trips): the wave-per-atom kernels are chains of such
// sums, so their latency is the kernel time.  red8: sum over each aligned group of 8 lanes (one head), in every lane of it.
#define SCANN_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float red8(float v) {
  v += SCANN_DPP(v, 0xB1);   // quad_perm [1,0,3,2]: lane ^ 1
  v += SCANN_DPP(v, 0x4E);   // quad_perm [2,3,0,1]: lane ^ 2
  v += SCANN_DPP(v, 0x141);  // row_half_mirror: the other quad of the group of 8
  return v;
}
__device__ __forceinline__ float red64(float v) {
  v = red8(v);
  v += SCANN_DPP(v, 0x128);  // row_ror:8 -- the other half of the row of 16
  v = xor16(v);  // (lane swaps: no LDS round trip, same additions)
  v = xor32(v);
  return v;
}

__global__ __launch_bounds__(256) void attn_bwd16_kernel(const float* __restrict__ q, const float* __restrict__ K,
                                                         const int* __restrict__ edge_offset,
                                                         const float* __restrict__ dctx, const float* __restrict__ gamma,
                                                         float* __restrict__ dq, float* __restrict__ dK,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int n_atom,
                                                         int atoms_per_wave, float drop_p, unsigned drop_tag,
                                                         unsigned long long drop_seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int at0 = (blockIdx.x * 4 + wave) * atoms_per_wave;
  const float2 g = reinterpret_cast<const float2*>(gamma)[lane];
  float2 dg = make_float2(0.f, 0.f), dbt = dg;
  // several atoms per wave: the LayerNorm parameter gradients of a workgroup end in 256 atomics on the SAME 256 addresses for
  // every workgroup of the launch, so the number of workgroups, not the arithmetic, set this kernel's time
  for (int at = at0; at < min(n_atom, at0 + atoms_per_wave); ++at) {
    const int e0 = edge_offset[at], deg = edge_offset[at + 1] - e0;
    const float2 q2 = reinterpret_cast<const float2*>(q)[(size_t)at * 64 + lane];
    const float2 dyv = reinterpret_cast<const float2*>(dctx)[(size_t)at * 64 + lane];
    const float qx = q2.x * 0.25f, qy = q2.y * 0.25f;
    float2 k2[16];
    float ev[16];
    float m = -INFINITY;
    // the sixteen key rows are requested together from clamped edge rows and masked afterwards (a load under a per-lane
    // guard is compiled as one full memory round trip each)
    if (deg > 0) {
#pragma unroll
      for (int n = 0; n < 16; ++n) k2[n] = reinterpret_cast<const float2*>(K)[(size_t)(e0 + min(n, deg - 1)) * 64 + lane];
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      if (n >= deg) k2[n] = make_float2(0.f, 0.f);
      float e = qx * k2[n].x + qy * k2[n].y;
      e = red8(e);
      ev[n] = n < deg ? e : -INFINITY;
      m = fmaxf(m, ev[n]);
    }
    float ssum = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      ev[n] = n < deg ? __builtin_amdgcn_exp2f((ev[n] - m) * 1.44269504088896340736f) : 0.f;  // v_exp_f32, as in the forward
      ssum += ev[n];
    }
    const float rsum = deg > 0 ? __builtin_amdgcn_rcpf(ssum) : 0.f;
    float px = q2.x, py = q2.y;
    float keep[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      ev[n] = ev[n] * rsum;  // attention weight (rsum = 1/sum through v_rcp_f32, as in the forward; 0 for an atom without edges)
      keep[n] = (drop_p > 0.f && n < deg) ? drop_scale(drop_seed, drop_tag, (size_t)(e0 + n) * NHEAD + (lane >> 3), drop_p) : 1.0f;
      px += ev[n] * keep[n] * k2[n].x;
      py += ev[n] * keep[n] * k2[n].y;
    }
    const float s = red64(px + py);
    const float mean = s * (1.0f / D);
    const float cx = px - mean, cy = py - mean;
    const float v = red64(cx * cx + cy * cy);
    const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
    const float hx = cx * rstd, hy = cy * rstd;
    const float ax = dyv.x * g.x, ay = dyv.y * g.y;
    const float m1 = red64(ax + ay) * (1.0f / D), m2 = red64(ax * hx + ay * hy) * (1.0f / D);
    const float dpx = rstd * (ax - m1 - hx * m2), dpy = rstd * (ay - m1 - hy * m2);
    dg.x += dyv.x * hx; dg.y += dyv.y * hy; dbt.x += dyv.x; dbt.y += dyv.y;
    float da[16];
    float dot = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      float d = dpx * k2[n].x + dpy * k2[n].y;
      d = red8(d);
      da[n] = d * keep[n];
      dot += ev[n] * da[n];
    }
    float dqx = dpx, dqy = dpy;
#pragma unroll
    for (int n = 0; n < 16; ++n)
      if (n < deg) {
        const float de = ev[n] * (da[n] - dot);
        reinterpret_cast<float2*>(dK)[(size_t)(e0 + n) * 64 + lane] =
            make_float2(ev[n] * keep[n] * dpx + 0.25f * de * q2.x, ev[n] * keep[n] * dpy + 0.25f * de * q2.y);
        dqx += 0.25f * de * k2[n].x;
        dqy += 0.25f * de * k2[n].y;
      }
    reinterpret_cast<float2*>(dq)[(size_t)at * 64 + lane] = make_float2(dqx, dqy);
  }
  __shared__ float sred[4][4 * 64];
  sred[wave][lane] = dg.x; sred[wave][64 + lane] = dg.y; sred[wave][128 + lane] = dbt.x; sred[wave][192 + lane] = dbt.y;
  __syncthreads();
  const int t = threadIdx.x;
  const float tot = (sred[0][t] + sred[1][t]) + (sred[2][t] + sred[3][t]);
  const int ln = t & 63, which = t >> 6;
  (which < 2 ? dgamma : dbeta)[(size_t)blockIdx.x * D + 2 * ln + (which & 1)] = tot;  // this workgroup's slot
}

// atoms per wave of attn_bwd16_kernel: the wave walks its atoms one after the other (each a chain of loads and lane reductions), so
// as few as fill the chip a few times over (~1,000 workgroups); the per-workgroup gamma / beta slots are tiny
static int attn_apw16(int n_atom) { return std::min(8, std::max(1, (n_atom + 4095) / 4096)); }
int attn_bwd_slots(int n_atom, int max_degree) {
  if (max_degree <= 16) {
    const int apw16 = attn_apw16(n_atom);
    return (n_atom + 4 * apw16 - 1) / (4 * apw16);  // one slot per workgroup
  }
  return 4 * ((n_atom + 4 * 2 - 1) / (4 * 2));      // one slot per wave
}
void launch_attn_bwd(WgradCtx& ctx, const float* q, const float* K, const int* edge_offset, const float* dctx, const float* gamma, float* dq,
                     float* dK, float* dgamma, float* dbeta, int n_atom, int max_degree, float drop_p, unsigned drop_tag,
                     unsigned long long drop_seed, hipStream_t s) {
  if (n_atom <= 0) return;
  const int n_slot = attn_bwd_slots(n_atom, max_degree);
  float* gp = reserve_vec(ctx, dgamma, n_slot);
  float* bp = reserve_vec(ctx, dbeta, n_slot);
  if (max_degree <= 16) {
    const int apw16 = attn_apw16(n_atom);
    hipLaunchKernelGGL(attn_bwd16_kernel, dim3((n_atom + 4 * apw16 - 1) / (4 * apw16)), dim3(256), 0, s, q, K, edge_offset, dctx,
                       gamma, dq, dK, gp, bp, n_atom, apw16, drop_p, drop_tag, drop_seed);
    return;
  }
  const int apw = 2;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((n_atom + 4 * apw - 1) / (4 * apw)), dim3(256), 0, s, q, K, edge_offset, dctx,
                     gamma, dq, dK, gp, bp, n_atom, apw, drop_p, drop_tag, drop_seed);
}

// ---- readout backward: one workgroup per structure ------------------------------------------------------------------
// forward (attention.py:279-316, scann_model.py:437-447): a_i = sum_{j!=i} gk_i.gq_j ; an = a/|a| ; at = softmax(an);
// rep = sum_i at_i gk_i ; pre = rep.Wb + bb ; h = swish(pre) ; y = h.wo + bo (mrelu has an identity gradient).
__global__ __launch_bounds__(128) void readout_bwd_kernel(ReadoutBwdArgs a) {
  extern __shared__ float sd[];  // [n] attn, [n] agg, [n] da
  __shared__ float sRep[D], sV[D], sS[D], sW[D], sRed[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mol = blockIdx.x;
  const int a0 = a.mol_offset[mol];
  const int n = a.mol_offset[mol + 1] - a0;
  float* sAt = sd;
  float* sAgg = sd + n;
  float* sDa = sd + 2 * n;
  const float dy = a.dy[mol];
  // rep, S = sum_j gq_j
  // (every per-atom / per-feature sum of this kernel keeps its order and requests its operands eight or sixteen at a time: one
  // workgroup per structure is a chain of such sums, and a load per iteration made each of them n or 128 memory round trips)
  float rep = 0.f, S = 0.f;
  for (int i0 = 0; i0 < n; i0 += 8) {
    float g[8], k[8], q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + u, n - 1);
      g[u] = a.ga[a0 + i];
      k[u] = a.gk[(size_t)(a0 + i) * D + tid];
      q[u] = a.gq[(size_t)(a0 + i) * D + tid];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < n) {
        rep += g[u] * k[u];
        S += q[u];
      }
  }
  sRep[tid] = rep;
  sS[tid] = S;
  for (int i = tid; i < n; i += 128) sAt[i] = a.ga[a0 + i];
  __syncthreads();
  // head: pre_j, h_j
  float pre = a.bb[tid];
  for (int k0 = 0; k0 < D; k0 += 16) {
    float w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = a.Wb[(k0 + u) * D + tid];
#pragma unroll
    for (int u = 0; u < 16; ++u) pre += sRep[k0 + u] * w[u];
  }
  const float hj = swish_(pre);
  const float dh = dy * a.wo[tid];
  const float dpre = dh * dswish_(pre);
  a.dwo[(size_t)mol * D + tid] = dy * hj;  // this structure's slot (vec_reduce_kernel adds the slots in order)
  if (tid == 0) atomicAdd(a.dbo, dy);
  a.rep_out[(size_t)mol * D + tid] = rep;
  a.dpre_out[(size_t)mol * D + tid] = dpre;
  sV[tid] = dpre;
  __syncthreads();
  // drep_k = sum_j dpre_j Wb[k][j]
  float drep = 0.f;
  for (int j0 = 0; j0 < D; j0 += 16) {  // this thread's ROW of Wb: four 16-byte loads per step
    float4 w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const float4*>(a.Wb + (size_t)tid * D + j0 + 4 * u);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      drep += sV[j0 + 4 * u] * w[u].x;
      drep += sV[j0 + 4 * u + 1] * w[u].y;
      drep += sV[j0 + 4 * u + 2] * w[u].z;
      drep += sV[j0 + 4 * u + 3] * w[u].w;
    }
  }
  __syncthreads();
  sV[tid] = drep;  // now drep
  __syncthreads();
  // per atom: dat_i = drep.gk_i ; a_i = gk_i.(S - gq_i)
  for (int i = wave; i < n; i += 2) {
    const float2 k2 = reinterpret_cast<const float2*>(a.gk)[(size_t)(a0 + i) * 64 + lane];
    const float2 q2 = reinterpret_cast<const float2*>(a.gq)[(size_t)(a0 + i) * 64 + lane];
    float d1 = sV[2 * lane] * k2.x + sV[2 * lane + 1] * k2.y;
    float d2 = k2.x * (sS[2 * lane] - q2.x) + k2.y * (sS[2 * lane + 1] - q2.y);
    d1 = wave_sum64(d1);  // (the lane ^ 1 ... ^ 32 butterfly's additions as DPP / lane swaps: no LDS round trips)
    d2 = wave_sum64(d2);
    if (lane == 0) {
      sDa[i] = d1;   // dat_i
      sAgg[i] = d2;  // a_i
    }
  }
  __syncthreads();
  if (wave == 0) {  // softmax and normalisation backward: one wave, a lane per atom (strided beyond 64), sums across the lanes
    // (round 5: thread 0 used to walk the atoms three times on its own -- ~50 dependent LDS reads and divisions per structure)
    float dot = 0.f, nrm2 = 0.f;
    for (int i = lane; i < n; i += 64) {
      dot += sAt[i] * sDa[i];
      nrm2 += sAgg[i] * sAgg[i];
    }
    dot = wave_sum64(dot);
    nrm2 = wave_sum64(nrm2);
    const float nrm = sqrtf(nrm2);
    float proj = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float dan = sAt[i] * (sDa[i] - dot);  // dan_i
      sDa[i] = dan;
      if (a.use_ga_norm) proj += (sAgg[i] / nrm) * dan;
    }
    if (a.use_ga_norm) {
      proj = wave_sum64(proj);
      for (int i = lane; i < n; i += 64) sDa[i] = (sDa[i] - (sAgg[i] / nrm) * proj) / nrm;  // da_i
    }
    if (lane == 0) sRed[0] = 0.f;
  }
  __syncthreads();
  // W = sum_i da_i gk_i ; dgk_i = at_i*drep + da_i (S - gq_i) ; dgq_j = W - da_j gk_j
  float Wacc = 0.f;
  for (int i0 = 0; i0 < n; i0 += 8) {
    float k[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) k[u] = a.gk[(size_t)(a0 + min(i0 + u, n - 1)) * D + tid];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < n) Wacc += sDa[i0 + u] * k[u];
  }
  sW[tid] = Wacc;
  for (int i0 = 0; i0 < n; i0 += 8) {
    float k[8], q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t o = (size_t)(a0 + min(i0 + u, n - 1)) * D + tid;
      k[u] = a.gk[o];
      q[u] = a.gq[o];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < n) {
        const size_t o = (size_t)(a0 + i0 + u) * D + tid;
        a.dgk[o] = sAt[i0 + u] * sV[tid] + sDa[i0 + u] * (sS[tid] - q[u]);
        a.dgq[o] = Wacc - sDa[i0 + u] * k[u];
      }
  }
}
void launch_readout_bwd(const ReadoutBwdArgs& a, hipStream_t s) {
  if (a.n_struct <= 0) return;
  hipLaunchKernelGGL(readout_bwd_kernel, dim3(a.n_struct), dim3(128), (size_t)3 * a.max_atoms * sizeof(float), s, a);
}

// ---- basis backward: geom0 = swish(gd.Wd + bd) * swish(gw.Ww + bw)  (scann_model.py:378-389) ---------------------------
__device__ __forceinline__ float gauss_(float x, float c) {
  const float d = x - c;
  return expf(-(d * d) / 0.25f);
}
__global__ __launch_bounds__(128) void basis_bwd_kernel(BasisParams p, const float* __restrict__ dist,
                                                        const float* __restrict__ weight, const float* __restrict__ dgeom,
                                                        int n_edge, int chunks, float* dWd, float* dbd, float* dWw, float* dbw) {
  // `chunks` blocks of 32 edges per workgroup, gradients accumulated in registers: the 42 x 128 atomics at the end go to the
  // same addresses for every workgroup of the launch, so their number is what this kernel's time scales with
  __shared__ float sG[32][2 * NG];
  const int tid = threadIdx.x;
  float wd[NG], ww[NG], gdw[NG], gww[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    wd[k] = p.Wd[k * D + tid];
    ww[k] = p.Ww[k * D + tid];
    gdw[k] = 0.f;
    gww[k] = 0.f;
  }
  const float bd = p.bd[tid], bw = p.bw[tid];
  float gbd = 0.f, gbw = 0.f;
  for (int ch = 0; ch < chunks; ++ch) {
  const int e0 = (blockIdx.x * chunks + ch) * 32;
  if (e0 >= n_edge) break;
  const int ne = min(32, n_edge - e0);
  __syncthreads();
  for (int i = tid; i < 32 * 2 * NG; i += 128) {
    const int e = i / (2 * NG), k = i % (2 * NG);
    float v = 0.f;
    if (e < ne) v = k < NG ? gauss_(dist[e0 + e], p.cd[k]) : gauss_(weight[e0 + e], p.cw[k - NG]);
    sG[e][k] = v;
  }
  __syncthreads();
  for (int e = 0; e < ne; ++e) {
    float ad = 0.f, aw = 0.f;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      ad += sG[e][k] * wd[k];
      aw += sG[e][NG + k] * ww[k];
    }
    ad += bd; aw += bw;
    const float dg = dgeom[(size_t)(e0 + e) * D + tid];
    const float dpd = dg * swish_(aw) * dswish_(ad), dpw = dg * swish_(ad) * dswish_(aw);
    gbd += dpd; gbw += dpw;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      gdw[k] += sG[e][k] * dpd;
      gww[k] += sG[e][NG + k] * dpw;
    }
  }
  }
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    atomicAdd(&dWd[k * D + tid], gdw[k]);
    atomicAdd(&dWw[k * D + tid], gww[k]);
  }
  atomicAdd(&dbd[tid], gbd);
  atomicAdd(&dbw[tid], gbw);
}
void launch_basis_bwd(const BasisParams& p, const float* dist, const float* weight, const float* dgeom, int n_edge, float* dWd,
                      float* dbd, float* dWw, float* dbw, hipStream_t s) {
  if (n_edge > 0)
  {
    const int chunks = n_edge >= (1 << 20) ? 4 : 1;  // the per-edge arithmetic, not the final atomics, bounds this kernel: 4 x fewer workgroups measured 105 vs 37 us at 18 k edges
    hipLaunchKernelGGL(basis_bwd_kernel, dim3((n_edge + 32 * chunks - 1) / (32 * chunks)), dim3(128), 0, s, p, dist, weight, dgeom,
                       n_edge, chunks, dWd, dbd, dWw, dbw);
  }
}

// ---- base SCANN branch: geomL = swish(gd.Wf + bf) * weight  (attention.py:155; gd = raw Gaussian basis [E,20]) -------------
// backward: dpre = dgeomL * weight * swish'(pre); dWf[k][col] += gd[e][k] dpre; dbf[col] += dpre   (32 edges per workgroup)
__global__ __launch_bounds__(128) void base_geom_bwd_kernel(const float* __restrict__ gd, const float* __restrict__ Wf,
                                                            const float* __restrict__ bf, const float* __restrict__ wgt,
                                                            const float* __restrict__ dgeomL, int n_edge, float* dWf, float* dbf) {
  __shared__ float sG[32][NG];
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * 32;
  const int ne = min(32, n_edge - e0);
  for (int i = tid; i < 32 * NG; i += 128) sG[i / NG][i % NG] = (i / NG) < ne ? gd[(size_t)(e0 + i / NG) * NG + (i % NG)] : 0.f;
  __syncthreads();
  float w[NG], gw[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    w[k] = Wf[k * D + tid];
    gw[k] = 0.f;
  }
  const float b = bf[tid];
  float gb = 0.f;
  for (int e = 0; e < ne; ++e) {
    float pre = b;
#pragma unroll
    for (int k = 0; k < NG; ++k) pre += sG[e][k] * w[k];
    const float dp = dgeomL[(size_t)(e0 + e) * D + tid] * wgt[e0 + e] * dswish_(pre);
    gb += dp;
#pragma unroll
    for (int k = 0; k < NG; ++k) gw[k] += sG[e][k] * dp;
  }
#pragma unroll
  for (int k = 0; k < NG; ++k) atomicAdd(&dWf[k * D + tid], gw[k]);
  atomicAdd(&dbf[tid], gb);
}
void launch_base_geom_bwd(const float* gd, const float* Wf, const float* bf, const float* wgt, const float* dgeomL, int n_edge,
                          float* dWf, float* dbf, hipStream_t s) {
  if (n_edge > 0)
    hipLaunchKernelGGL(base_geom_bwd_kernel, dim3((n_edge + 31) / 32), dim3(128), 0, s, gd, Wf, bf, wgt, dgeomL, n_edge, dWf, dbf);
}

// ---- embedding backward (Embedding + dense_embed, scann_model.py:362,373) -------------------------------------------------
// c0[a] = drop(swish(pre[Z_a])), pre[s] = E[s].W + b.   dlut[s] = sum_{a: Z_a = s} dc0[a] (dropout already applied to dc0).
// One workgroup (128 threads = columns) per run of `run` atoms, with a private [n_species][128] table in LDS (thread `col` is
// the only writer of column `col`): the global table has only n_species x 128 addresses and every atomic on them is serialised
// behind all the others of the launch, so each workgroup adds its totals once.
__global__ __launch_bounds__(128) void embed_scatter_kernel(const float* __restrict__ dc0, const int* __restrict__ atomic,
                                                            float* __restrict__ dlut, int n_atom, int run, int n_species,
                                                            unsigned long long drop_seed, unsigned drop_tag, float drop_p) {
  extern __shared__ float sTab[];  // [n_species][D]
  const int col = threadIdx.x, a0 = blockIdx.x * run, a1 = min(n_atom, a0 + run);
  for (int sp = 0; sp < n_species; ++sp) sTab[sp * D + col] = 0.f;
  // (the Dropout(0.1) mask of the embedding rows, scann_model.py:374, is applied on the way in: no dropout launch before this one)
  // (eight rows requested per step, added in row order: a load per iteration made this a chain of `run` memory round trips)
  for (int ab = a0; ab < a1; ab += 8) {
    float v[8];
    int z[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int a = min(ab + u, a1 - 1);
      v[u] = dc0[(size_t)a * D + col];
      z[u] = atomic[a];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (ab + u < a1) {
        const size_t i = (size_t)(ab + u) * D + col;
        sTab[z[u] * D + col] += drop_p > 0.f ? v[u] * drop_scale(drop_seed, drop_tag, i, drop_p) : v[u];
      }
  }
  for (int sp = 0; sp < n_species; ++sp) {
    const float v = sTab[sp * D + col];
    if (v != 0.f) atomicAdd(&dlut[(size_t)sp * D + col], v);
  }
}
__global__ __launch_bounds__(128) void embed_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ W,
                                                        const float* __restrict__ b, float* __restrict__ dlut,
                                                        int n_species, int emb_dim, float* dEmb, float* dW, float* db) {
  // one workgroup per species; thread = output column
  extern __shared__ float sdp[];  // [128] dpre
  const int sp = blockIdx.x, col = threadIdx.x;
  const float dl = dlut[sp * D + col];
  dlut[sp * D + col] = 0.f;  // consumed: the table is zero again for the next step's embed_scatter_kernel (no memset per step)
  // a species absent from the batch has a zero row: nothing to add anywhere (QM9: 5 of the table's species occur; their workgroups'
  // atomics on the shared dW / db addresses were queueing behind ~20 x as many adds of zero)
  if (!__syncthreads_or(dl != 0.f)) return;
  float pre = b[col];
  for (int k0 = 0; k0 < emb_dim; k0 += 16) {  // the same sum in the same order, operands requested sixteen at a time
    float e[16], w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = min(k0 + u, emb_dim - 1);
      e[u] = emb[sp * emb_dim + k];
      w[u] = W[k * D + col];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (k0 + u < emb_dim) pre += e[u] * w[u];
  }
  const float dpre = dl * dswish_(pre);
  sdp[col] = dpre;
  if (dpre != 0.f) {
    atomicAdd(&db[col], dpre);
    for (int k = 0; k < emb_dim; ++k) atomicAdd(&dW[k * D + col], emb[sp * emb_dim + k] * dpre);
  }
  __syncthreads();
  for (int k = col; k < emb_dim; k += 128) {
    float acc = 0.f;
    for (int j0 = 0; j0 < D; j0 += 16) {  // this thread's row of W, sixteen elements per step (same order of additions)
      float4 w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const float4*>(W + (size_t)k * D + j0 + 4 * u);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc += sdp[j0 + 4 * u] * w[u].x;
        acc += sdp[j0 + 4 * u + 1] * w[u].y;
        acc += sdp[j0 + 4 * u + 2] * w[u].z;
        acc += sdp[j0 + 4 * u + 3] * w[u].w;
      }
    }
    dEmb[sp * emb_dim + k] += acc;  // accumulate; each species row is touched by exactly one workgroup
  }
}
void launch_embed_bwd(const float* dc0, const int* atomic, int n_atom, const float* emb, const float* W, const float* b,
                      float* dlut, int n_species, int emb_dim, float* dEmb, float* dW, float* db, unsigned long long drop_seed,
                      unsigned drop_tag, float drop_p, hipStream_t s) {
  if (n_atom <= 0) return;
  const int run = std::max(32, (n_atom + 127) / 128);  // <= 128 workgroups
  hipLaunchKernelGGL(embed_scatter_kernel, dim3((n_atom + run - 1) / run), dim3(128), (size_t)n_species * D * sizeof(float), s, dc0,
                     atomic, dlut, n_atom, run, n_species, drop_seed, drop_tag, drop_p);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(n_species), dim3(128), D * sizeof(float), s, emb, W, b, dlut, n_species, emb_dim,
                     dEmb, dW, db);
}

// ---- general embedding backward (use_ring / feature="cgcnn", scann_model.py:361-374) --------------------------------------
// forward: v = concat[E[Z] | X.We + be, ring.Wr + br] (cin values), c0 = swish(v.Wde + bde).  8 atoms per 128-thread workgroup;
// dc0 already carries the dropout scale.  Small tensors: every gradient goes through float atomics.
__global__ __launch_bounds__(128) void embed_general_bwd_kernel(EmbedArgs a, const float* __restrict__ dc0, float* dEmb, float* dWe,
                                                                float* dbe, float* dWr, float* dbr, float* dWde, float* dbde) {
  __shared__ float sV[8][160], sDp[8][D], sDv[8][160];
  const int tid = threadIdx.x;
  const int a0 = blockIdx.x * 8;
  const int na = min(8, a.n_atom - a0);
  const int cin = a.emb_dim + (a.ring ? 10 : 0);
  for (int i = tid; i < na * cin; i += 128) {
    const int la = i / cin, k = i % cin, at = a0 + la;
    float v;
    if (k < a.emb_dim) {
      if (a.cgcnn) {
        float acc = 0.f;
        for (int j = 0; j < 92; ++j) acc += a.cgcnn[(size_t)at * 92 + j] * a.We[j * a.emb_dim + k];
        v = acc + a.be[k];
      } else {
        v = a.emb[(size_t)a.atomic[at] * a.emb_dim + k];
      }
    } else {
      const int r = k - a.emb_dim;
      v = (a.ring[(size_t)at * 2] * a.Wr[r] + a.ring[(size_t)at * 2 + 1] * a.Wr[10 + r]) + a.br[r];
    }
    sV[la][k] = v;
  }
  __syncthreads();
  float gb = 0.f;
  for (int la = 0; la < na; ++la) {  // thread = output column
    float pre = a.bde[tid];
    for (int k = 0; k < cin; ++k) pre += sV[la][k] * a.Wde[k * D + tid];
    const float dp = dc0[(size_t)(a0 + la) * D + tid] * dswish_(pre);
    sDp[la][tid] = dp;
    gb += dp;
  }
  atomicAdd(&dbde[tid], gb);
  __syncthreads();
  for (int k = 0; k < cin; ++k) {  // dWde[k][col] += sum_la v[la][k] dpre[la][col]
    float g = 0.f;
    for (int la = 0; la < na; ++la) g += sV[la][k] * sDp[la][tid];
    atomicAdd(&dWde[k * D + tid], g);
  }
  for (int i = tid; i < na * cin; i += 128) {  // dv[la][k] = sum_col dpre[la][col] Wde[k][col]
    const int la = i / cin, k = i % cin;
    float acc = 0.f;
    for (int c = 0; c < D; ++c) acc += sDp[la][c] * a.Wde[k * D + c];
    sDv[la][k] = acc;
  }
  __syncthreads();
  for (int i = tid; i < na * cin; i += 128) {
    const int la = i / cin, k = i % cin, at = a0 + la;
    const float dv = sDv[la][k];
    if (k < a.emb_dim) {
      if (a.cgcnn) {
        atomicAdd(&dbe[k], dv);
        for (int j = 0; j < 92; ++j) {
          const float x = a.cgcnn[(size_t)at * 92 + j];
          if (x != 0.f) atomicAdd(&dWe[j * a.emb_dim + k], x * dv);
        }
      } else {
        atomicAdd(&dEmb[(size_t)a.atomic[at] * a.emb_dim + k], dv);
      }
    } else {
      const int r = k - a.emb_dim;
      atomicAdd(&dbr[r], dv);
      atomicAdd(&dWr[r], a.ring[(size_t)at * 2] * dv);
      atomicAdd(&dWr[10 + r], a.ring[(size_t)at * 2 + 1] * dv);
    }
  }
}
void launch_embed_general_bwd(const EmbedArgs& a, const float* dc0, float* dEmb, float* dWe, float* dbe, float* dWr, float* dbr,
                              float* dWde, float* dbde, hipStream_t s) {
  if (a.n_atom > 0)
    hipLaunchKernelGGL(embed_general_bwd_kernel, dim3((a.n_atom + 7) / 8), dim3(128), 0, s, a, dc0, dEmb, dWe, dbe, dWr, dbr, dWde, dbde);
}

// ---- loss ------------------------------------------------------------------------------------------------------------------
// out = {sum (y - t)^2, n, sum |y - t|}.  Optional extras of scann_train_step (null otherwise): t may be pinned host memory, copied to t_dev on the
// way; host_stat (pinned) receives the pair without a copy operation; dy = d rmse / d y for a single-rank step (the batch IS the
// global batch, losses.py:5-6), which saves the separate dy launch.
__global__ void sse_kernel(const float* __restrict__ y, const float* __restrict__ t, int n, double* __restrict__ out,
                           float* __restrict__ t_dev, float* __restrict__ dy, double* __restrict__ host_stat) {
  double s = 0.0, ab = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float ti = t[i];
    if (t_dev) t_dev[i] = ti;
    const double d = (double)y[i] - (double)ti;
    s += d * d;
    ab += fabs(d);
  }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = ab;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  const double sse = sh[0][0];
  if (threadIdx.x == 0) {
    out[0] = sse;
    out[1] = (double)n;  // the count and the absolute-error sum travel with the sum of squares (scann_train_step all-reduces the three)
    out[2] = sh[1][0];
    if (host_stat) {
      host_stat[0] = sse;
      host_stat[1] = (double)n;
      host_stat[2] = sh[1][0];
    }
  }
  if (dy) {
    const double rmse = sqrt(sse / (double)n);
    const float scale = rmse > 0 ? (float)(1.0 / ((double)n * rmse)) : 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dy[i] = (y[i] - t[i]) * scale;
  }
}
void launch_sse(const float* y, const float* t, int n, double* out, float* t_dev, float* dy, double* host_stat, hipStream_t s) {
  hipLaunchKernelGGL(sse_kernel, dim3(1), dim3(256), 0, s, y, t, n, out, t_dev, dy, host_stat);
}
// d rmse / d y_i = (y_i - t_i) / (count * rmse)   (losses.py:5-6 over the GLOBAL batch)
__global__ void dy_kernel(const float* __restrict__ y, const float* __restrict__ t, int n, float scale, const double* __restrict__ stat,
                          float* __restrict__ dy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (stat) {  // {global sse, global count} on the device: the same double arithmetic as the host path
    const double rmse = sqrt(stat[0] / stat[1]);
    scale = rmse > 0 ? (float)(1.0 / (stat[1] * rmse)) : 0.f;
  }
  if (i < n) dy[i] = (y[i] - t[i]) * scale;
}
void launch_dy(const float* y, const float* t, int n, float scale, const double* stat, float* dy, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(dy_kernel, dim3((n + 255) / 256), dim3(256), 0, s, y, t, n, scale, stat, dy);
}

// ---- Adam (tf.keras.optimizers.Adam, epsilon 1e-7) + l2 regulariser gradient ------------------------------------------------
// g += 2*l2*w on regularised kernels; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_hat * m / (sqrt(v) + eps)
// with lr_hat = lr_t * sqrt(1 - b2^t) / (1 - b1^t) computed by the caller.
__global__ void adam_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ l2mask, size_t n, float lr_hat, float b1, float b2, float eps,
                            float l2, int zero_g) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] + 2.0f * l2 * l2mask[i] * w[i];
  if (zero_g) g[i] = 0.f;  // scann_train_step: the next step starts from a zeroed gradient vector without a memset
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  w[i] -= lr_hat * mi / (sqrtf(vi) + eps);
}
void launch_adam(float* w, float* g, float* m, float* v, const float* l2mask, size_t n, float lr_hat, float b1, float b2,
                 float eps, float l2, int zero_g, hipStream_t s) {
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, g, m, v, l2mask, n, lr_hat, b1, b2, eps, l2, zero_g);
}

// ---- master weights -> MFMA fragment order (after every optimiser step) --------------------------------------------------------
// dst[((w*16 + t)*64 + lane)*4 + i] = M[8t + 4(lane>>5) + i][32w + (lane&31)], M = W (ld 128) or W^T.
__global__ void repack_kernel(const RepackDesc* __restrict__ descs, const float* __restrict__ master, float* __restrict__ arena,
                              int32_t* __restrict__ range_flag) {
  const RepackDesc d = descs[blockIdx.y];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 16383
  if (idx >= D * D) return;
  if (d.raw > 0) {
    if (idx < d.raw) arena[d.dst + idx] = master[d.src + idx];
    return;
  }
  if (d.raw < 0) {  // split-fp16 image (pack_weight_f16): this thread writes one 4-byte slot = halfs j, j + 1 of one lane
    const int ks = d.raw == -1 ? 8 : 2, k_real = d.raw == -1 ? D : NG;
    if (idx >= ks * 2048) return;
    const int jp = idx & 3, lane = (idx >> 2) & 63, plane = (idx >> 8) & 1, sw = idx >> 9, st = sw % ks, w = sw / ks;
    _Float16 out[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = 16 * st + 8 * (lane >> 5) + 2 * jp + u;
      const int col = 32 * w + (lane & 31);
      const float x = k < k_real ? WSCALE * master[d.src + (d.transpose ? (size_t)col * D + k : (size_t)k * D + col)] : 0.f;
      const _Float16 hi = (_Float16)x;
      if (!(fabsf(x) < 65504.f)) flag_range(range_flag, 5, -1);  // the optimiser pushed a weight past 65504 / 2^8 (or made it NaN)
      out[u] = plane ? (_Float16)(x - (float)hi) : hi;
    }
    reinterpret_cast<_Float16*>(arena + d.dst)[2 * idx] = out[0];
    reinterpret_cast<_Float16*>(arena + d.dst)[2 * idx + 1] = out[1];
    return;
  }
  const int i = idx & 3, lane = (idx >> 2) & 63, t = (idx >> 8) & 15, w = idx >> 12;
  const int k = 8 * t + 4 * (lane >> 5) + i, j = 32 * w + (lane & 31);
  arena[d.dst + idx] = d.transpose ? master[d.src + (size_t)j * D + k] : master[d.src + (size_t)k * D + j];
}
void launch_repack(const RepackDesc* descs, int n, const float* master, float* arena, int32_t* range_flag, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(repack_kernel, dim3(D * D / 256, n), dim3(256), 0, s, descs, master, arena, range_flag);
}

}  // namespace scann
