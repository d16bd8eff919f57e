// Generic-width training: the backward pass of create_model (scann_model.py:329-453) for ANY local_dim / num_head / global_dim /
// dense_out, as plain fp32 kernels -- one formula per kernel, like the forward of scann_generic.hip whose kept tensors it reads.  The
// reference compiles and fits whatever create_model built (scann_model.py:199-241); every shipped yaml is 128 / 8 and trains on the
// MFMA kernels of scann_train.hip / scann_train_fused.hip, which are written for exactly those widths.
//
// No atomics anywhere: every sum has a fixed order (weight gradients: 32-row chunks in row order, row slabs in slab order; per-atom sums: the atom's own CSR
// row, then its incoming edges in reverse-adjacency order; LayerNorm gamma / beta: row chunks, then chunk order), so a step is
// bit-reproducible like the 128-wide one.
//
//   gen_transpose_kernel      W^T images of the kernels (d x = d z . W^T runs through gen_dense_kernel on them)
//   gen_act_bwd_kernel        d pre = d y * swish'(pre) [* row scale] [* Dropout mask]
//   gen_mul_gather_kernel     out = a * B[idx] [+ c]                                (geometry product, gated neighbour rows: attention.py:157)
//   gen_dense_dw_kernel       dW += X^T . d z, db += column sums of d z, X assembled from gathered segments like the forward's
//   gen_layernorm_bwd_kernel  d x of LayerNormalization(epsilon=1e-6) + per-row statistics; gen_ln_param_* : d gamma, d beta
//   gen_attn_bwd_kernel       per atom: softmax / logits / context backward (attention.py:170-212), Dropout on the attention weights
//   gen_pool_bwd_kernel       per structure: GlobalAttention pooling backward (attention.py:279-316)
//   gen_edge_to_atom_kernel   d c[a] = [d c[a]] + sum over a's own edges + sum over the edges that have a as their neighbour
//   gen_table_part / _final   Embedding gradient: rows of d v summed per species (per 64-atom chunk, then chunk order)
#include "scann_internal.h"
#include "scann_mma.h"

namespace scann {

namespace {

__device__ __forceinline__ float gen_dswish(float x) {  // d/dx x sigmoid(x) = s (1 + x (1 - s))
  const float s = 1.0f / (1.0f + expf(-x));
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float gt_wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// sum of p[0], p[stride], ... p[(n - 1) * stride] in that order; eight loads in flight (the additions keep their order)
__device__ __forceinline__ float gt_ordered_sum(const float* __restrict__ p, int n, size_t stride) {
  float s = 0.f;
  int c = 0;
  for (; c + 8 <= n; c += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(c + j) * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; c < n; ++c) s += p[(size_t)c * stride];
  return s;
}

// dst[o * kn + kk] = src[(k0 + kk) * N + o]: rows [k0, k0 + kn) of a row-major [*, N] kernel, transposed
__global__ void gen_transpose_kernel(const GenTransDesc* __restrict__ descs, const float* __restrict__ W, float* __restrict__ WT) {
  const GenTransDesc t = descs[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.kn * t.N) return;
  const int o = i / t.kn, kk = i - o * t.kn;
  WT[t.dst + i] = W[t.src + (size_t)(t.k0 + kk) * t.N + o];
}

__global__ void gen_act_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ pre, const float* __restrict__ row_scale, size_t n, int N,
                                   float drop_p, unsigned drop_tag, unsigned long long drop_seed, float* __restrict__ out) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = dY[i];
  if (row_scale) v = v * row_scale[i / (size_t)N];
  if (drop_p > 0.f) v = v * drop_scale(drop_seed, drop_tag, i, drop_p);
  if (pre) v = v * gen_dswish(pre[i]);
  out[i] = v;
}

__global__ void gen_mul_gather_kernel(const float* __restrict__ a, const float* __restrict__ B, const int32_t* __restrict__ idx, const float* __restrict__ c,
                                      size_t n, int N, float* __restrict__ out) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t r = i / (size_t)N, k = i - r * (size_t)N;
  float v = a[i] * B[(size_t)(idx ? idx[r] : (int32_t)r) * N + k];
  if (c) v = v + c[i];
  out[i] = v;
}

__global__ void gen_relu_kernel(float* __restrict__ y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fmaxf(y[i], 0.f);
}

// dW[k][o] = sum_r X[r][k] dZ[r][o]: a workgroup owns a 32 x 32 tile of dW and ONE SLAB of the rows (blockIdx.z), walked in 32-row
// chunks staged in LDS (X assembled from its gathered segments, or as the product of two; the next chunk's elements are requested
// before the current chunk is multiplied).  A chunk's partial sum is formed first and added to the running sum second (two-level:
// the rounding error of a long sum stays that of a few dozen additions), the slab's tile goes to part[slab][K][N] (+ [slab][N] column
// sums of dZ behind the tiles, written by the first tile row) and gen_dw_reduce_kernel adds the slabs in order into dW / db.
__global__ __launch_bounds__(256) void gen_dense_dw_kernel(GenDwArgs a) {
#pragma clang fp contract(off)
  __shared__ float sXc[32][33], sZc[32][33];
  const int tid = threadIdx.x, k0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
  const int tk = tid >> 4, to = tid & 15;  // this thread's outputs: k = k0 + tk + {0, 16}, o = o0 + to + {0, 16}
  const int rbeg = blockIdx.z * a.slab_rows, rend = min(a.rows, rbeg + a.slab_rows);
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, bacc[2] = {0.f, 0.f};
  float xr[4], zr[4];
  auto fetch = [&](int r0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * j, rr = i >> 5, kk = i & 31;
      const int r = r0 + rr, k = k0 + kk, o = o0 + kk;
      float xv = 0.f, zv = 0.f;
      if (r < rend) {
        if (k < a.K) {
          if (a.prod) {
            xv = a.seg[0].p[(size_t)(a.seg[0].idx ? a.seg[0].idx[r] : r) * a.K + k] * a.seg[1].p[(size_t)(a.seg[1].idx ? a.seg[1].idx[r] : r) * a.K + k];
          } else {
            int kq = k, sg = 0;
            while (sg < a.n_seg - 1 && kq >= a.seg[sg].w) {
              kq -= a.seg[sg].w;
              ++sg;
            }
            xv = a.seg[sg].p[(size_t)(a.seg[sg].idx ? a.seg[sg].idx[r] : r) * a.seg[sg].w + kq];
          }
        }
        if (o < a.N) zv = a.dZ[(size_t)r * a.N + o];
      }
      xr[j] = xv;
      zr[j] = zv;
    }
  };
  if (rbeg < rend) fetch(rbeg);
  for (int r0 = rbeg; r0 < rend; r0 += 32) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * j;
      sXc[i >> 5][i & 31] = xr[j];
      sZc[i >> 5][i & 31] = zr[j];
    }
    __syncthreads();
    if (r0 + 32 < rend) fetch(r0 + 32);
    float p[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, bp[2] = {0.f, 0.f};
#pragma unroll 8
    for (int rr = 0; rr < 32; ++rr) {
      const float x0 = sXc[rr][tk], x1 = sXc[rr][tk + 16], z0 = sZc[rr][to], z1 = sZc[rr][to + 16];
      p[0][0] = fmaf(x0, z0, p[0][0]);
      p[0][1] = fmaf(x0, z1, p[0][1]);
      p[1][0] = fmaf(x1, z0, p[1][0]);
      p[1][1] = fmaf(x1, z1, p[1][1]);
      bp[0] += z0;
      bp[1] += z1;
    }
    acc[0][0] += p[0][0]; acc[0][1] += p[0][1]; acc[1][0] += p[1][0]; acc[1][1] += p[1][1];
    bacc[0] += bp[0]; bacc[1] += bp[1];
    __syncthreads();
  }
  float* pt = a.part + (size_t)blockIdx.z * a.K * a.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + tk + 16 * i, o = o0 + to + 16 * j;
      if (k < a.K && o < a.N) pt[(size_t)k * a.N + o] = acc[i][j];
    }
  if (a.db && blockIdx.x == 0 && tk == 0) {
    float* pb = a.part + (size_t)gridDim.z * a.K * a.N + (size_t)blockIdx.z * a.N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int o = o0 + to + 16 * j;
      if (o < a.N) pb[o] = bacc[j];
    }
  }
}
// dW[i] += sum over the slabs, in slab order, of part[slab][i]; db likewise
__global__ void gen_dw_reduce_kernel(const float* __restrict__ part, int n_slab, int KN, int N, float* __restrict__ dW, float* __restrict__ db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < KN) dW[i] += gt_ordered_sum(part + i, n_slab, (size_t)KN);
  else if (db && i < KN + N) db[i - KN] += gt_ordered_sum(part + (size_t)n_slab * KN + (i - KN), n_slab, (size_t)N);
}

// one wave per row.  x = X (+ res) is the LayerNorm's input; xhat = (x - mean) rstd; g = dY gamma;
// dX = rstd (g - mean(g) - xhat mean(g xhat)); stats[r] = {mean, rstd} for the gamma / beta sums
__global__ __launch_bounds__(256) void gen_layernorm_bwd_kernel(const float* __restrict__ X, const float* __restrict__ res, const float* __restrict__ gamma,
                                                                const float* __restrict__ dY, int rows, int N, float* __restrict__ dX,
                                                                float* __restrict__ stats) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* x = X + (size_t)r * N;
  const float* q = res ? res + (size_t)r * N : nullptr;
  const float* dy = dY + (size_t)r * N;
  float s = 0.f;
  for (int k = lane; k < N; k += 64) s += q ? x[k] + q[k] : x[k];
  const float mean = gt_wave_sum64(s) / (float)N;
  float v = 0.f;
  for (int k = lane; k < N; k += 64) {
    const float d = (q ? x[k] + q[k] : x[k]) - mean;
    v += d * d;
  }
  const float rstd = 1.0f / sqrtf(gt_wave_sum64(v) / (float)N + 1e-6f);
  float sg = 0.f, sgx = 0.f;
  for (int k = lane; k < N; k += 64) {
    const float xh = ((q ? x[k] + q[k] : x[k]) - mean) * rstd, g = dy[k] * gamma[k];
    sg += g;
    sgx += g * xh;
  }
  const float mg = gt_wave_sum64(sg) / (float)N, mgx = gt_wave_sum64(sgx) / (float)N;
  for (int k = lane; k < N; k += 64) {
    const float xh = ((q ? x[k] + q[k] : x[k]) - mean) * rstd, g = dy[k] * gamma[k];
    dX[(size_t)r * N + k] = rstd * (g - mg - xh * mgx);
  }
  if (lane == 0) {
    stats[2 * (size_t)r] = mean;
    stats[2 * (size_t)r + 1] = rstd;
  }
}

// part[chunk][0][k] = sum over the chunk's rows of dY xhat, part[chunk][1][k] = ... of dY (a thread per column, rows in order)
__global__ void gen_ln_param_kernel(const float* __restrict__ X, const float* __restrict__ res, const float* __restrict__ dY, const float* __restrict__ stats,
                                    int rows, int N, int rows_per_chunk, float* __restrict__ part) {
#pragma clang fp contract(off)
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float sg = 0.f, sb = 0.f;
  int r = r0;
  for (; r + 4 <= r1; r += 4) {  // four rows requested together; the additions keep the row order
    float x[4], dd[4], mu[4], rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t i = (size_t)(r + j) * N + k;
      x[j] = res ? X[i] + res[i] : X[i];
      dd[j] = dY[i];
      mu[j] = stats[2 * (size_t)(r + j)];
      rs[j] = stats[2 * (size_t)(r + j) + 1];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sg += dd[j] * ((x[j] - mu[j]) * rs[j]);
      sb += dd[j];
    }
  }
  for (; r < r1; ++r) {
    const size_t i = (size_t)r * N + k;
    const float x = res ? X[i] + res[i] : X[i];
    const float xh = (x - stats[2 * (size_t)r]) * stats[2 * (size_t)r + 1], d = dY[i];
    sg += d * xh;
    sb += d;
  }
  part[((size_t)blockIdx.y * 2) * N + k] = sg;
  part[((size_t)blockIdx.y * 2 + 1) * N + k] = sb;
}
__global__ void gen_ln_param_final_kernel(const float* __restrict__ part, int n_chunk, int N, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  dgamma[k] += gt_ordered_sum(part + k, n_chunk, (size_t)2 * N);
  dbeta[k] += gt_ordered_sum(part + N + k, n_chunk, (size_t)2 * N);
}

// one workgroup per atom (the forward: gen_attn_kernel).  P = softmax of the scaled logits per head, P' = P * Dropout scale;
// context[o] = sum_n P'[n][h(o)] K[n][o] + q[o].  Given d context:
//   dP'[n][h] = sum over the head's columns of d ctx[o] K[n][o];  dP = dP' * scale;  d logit = P (dP - sum_m P[m] dP[m]);
//   dK[n][o] = P'[n][h] d ctx[o] + hd^-0.5 d logit[n][h] q[o];   dq[o] = d ctx[o] + hd^-0.5 sum_n d logit[n][h] K[n][o]
__global__ __launch_bounds__(256) void gen_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ K, const int32_t* __restrict__ edge_offset,
                                                           int d, int H, const float* __restrict__ dctx, float drop_p, unsigned drop_tag,
                                                           unsigned long long drop_seed, float* __restrict__ dq, float* __restrict__ dK) {
#pragma clang fp contract(off)
  extern __shared__ float sm[];  // [deg][H] P, [deg][H] P', [deg][H] dP -> d logit
  const int at = blockIdx.x, tid = threadIdx.x;
  const int e0 = edge_offset[at], deg = edge_offset[at + 1] - e0, hd = d / H;
  float* sP = sm;
  float* sPd = sP + (size_t)deg * H;
  float* sD = sPd + (size_t)deg * H;
  const float dk = 1.0f / sqrtf((float)hd);
  const float* qa = q + (size_t)at * d;
  const float* dc = dctx + (size_t)at * d;
  for (int i = tid; i < deg * H; i += 256) {
    const int n = i / H, h = i - n * H;
    const float* kr = K + (size_t)(e0 + n) * d + h * hd;
    float s = 0.f, g = 0.f;
    for (int j = 0; j < hd; ++j) {
      s = fmaf(qa[h * hd + j] * dk, kr[j], s);
      g = fmaf(dc[h * hd + j], kr[j], g);
    }
    sP[i] = s;
    sD[i] = g;  // dP'
  }
  __syncthreads();
  for (int h = tid; h < H; h += 256) {
    float m = -INFINITY;
    for (int n = 0; n < deg; ++n) m = fmaxf(m, sP[n * H + h]);
    float ss = 0.f;
    for (int n = 0; n < deg; ++n) {
      const float e = expf(sP[n * H + h] - m);
      sP[n * H + h] = e;
      ss += e;
    }
    float dot = 0.f;
    for (int n = 0; n < deg; ++n) {
      const float p = sP[n * H + h] / ss;
      const float sc = drop_p > 0.f ? drop_scale(drop_seed, drop_tag, (size_t)(e0 + n) * H + h, drop_p) : 1.0f;
      const float dp = sD[n * H + h] * sc;
      sP[n * H + h] = p;
      sPd[n * H + h] = p * sc;
      sD[n * H + h] = dp;
      dot += p * dp;
    }
    for (int n = 0; n < deg; ++n) sD[n * H + h] = sP[n * H + h] * (sD[n * H + h] - dot);
  }
  __syncthreads();
  for (int o = tid; o < d; o += 256) {
    const int h = o / hd;
    const float dco = dc[o], qo = qa[o];
    float s = 0.f;
    for (int n = 0; n < deg; ++n) {
      const float dl = sD[n * H + h];
      const size_t i = (size_t)(e0 + n) * d + o;
      s = fmaf(dl, K[i], s);
      dK[i] = sPd[n * H + h] * dco + dk * dl * qo;
    }
    dq[(size_t)at * d + o] = dco + dk * s;
  }
}

// one workgroup per structure (the forward: gen_readout_kernel up to the pooled row).  rep = sum_i at_i gk_i, at = softmax(u),
// u = agg / |agg| (use_ga_norm) or agg, agg_i = sum over j != i of gk_i . gq_j.  Given d rep:
//   d at_i = d rep . gk_i;  du = at (d at - sum_j at_j d at_j);  d agg = (du - u sum_j u_j du_j) / |agg|  (or du);
//   d gk_i = at_i d rep + d agg_i sum_{j != i} gq_j;   d gq_j = sum_{i != j} d agg_i gk_i
__global__ __launch_bounds__(256) void gen_pool_bwd_kernel(const int32_t* __restrict__ mol_offset, const float* __restrict__ gq, const float* __restrict__ gk, int dg,
                                                           int use_ga_norm, const float* __restrict__ drep, float* __restrict__ dgq, float* __restrict__ dgk) {
#pragma clang fp contract(off)
  // The per-structure scalars -- scores, normalisation, softmax and their derivatives -- in fp64 (n values per structure: no cost), on the
  // scores of gen_pool_score, the forward's own: d agg = (du - u (u . du)) / |agg| cancels ten-fold for a two-atom structure and is divided
  // by a norm that can be a thousandth of the score's terms, so every fp32 rounding in this chain is a 1e-4 error of the structure's
  // whole gradient (and through it of every tensor upstream of the readout).  With the chain exact, what is left is what the fp32
  // STORAGE of gq / gk / d rep carries in -- the floor of any fp32 graph (tools/debug_plain_grads.py, profiles/r06_notes.md).
  extern __shared__ double smd[];  // [n] u, [n] at, [n] d at -> d agg, [4] reductions
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a0 = mol_offset[blockIdx.x], n = mol_offset[blockIdx.x + 1] - a0;
  double* sU = smd;
  double* sA = sU + n;
  double* sD = sA + n;
  double* sRed = sD + n;
  const float* dr = drep + (size_t)blockIdx.x * dg;
  for (int i = tid; i < n; i += 256) {
    const float* ki = gk + (size_t)(a0 + i) * dg;
    sU[i] = gen_pool_score(gq, gk, a0, n, i, dg);
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
    int k = 0;
    for (; k + 4 <= dg; k += 4) {
      d0 = fma((double)dr[k], (double)ki[k], d0);
      d1 = fma((double)dr[k + 1], (double)ki[k + 1], d1);
      d2 = fma((double)dr[k + 2], (double)ki[k + 2], d2);
      d3 = fma((double)dr[k + 3], (double)ki[k + 3], d3);
    }
    for (; k < dg; ++k) d0 = fma((double)dr[k], (double)ki[k], d0);
    sD[i] = (d0 + d1) + (d2 + d3);
  }
  __syncthreads();
  auto wave_sum = [&](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  auto block_sum = [&](double v) {
    v = wave_sum(v);
    if (lane == 0) sRed[wave] = v;
    __syncthreads();
    const double t = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
    __syncthreads();
    return t;
  };
  auto block_max = [&](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    if (lane == 0) sRed[wave] = v;
    __syncthreads();
    const double t = fmax(fmax(sRed[0], sRed[1]), fmax(sRed[2], sRed[3]));
    __syncthreads();
    return t;
  };
  double nrm = 1.0;
  if (use_ga_norm) {
    double ss = 0.0;
    for (int i = tid; i < n; i += 256) ss += sU[i] * sU[i];
    nrm = sqrt(block_sum(ss));
    for (int i = tid; i < n; i += 256) sU[i] = sU[i] / nrm;
    __syncthreads();
  }
  double m = -INFINITY;
  for (int i = tid; i < n; i += 256) m = fmax(m, sU[i]);
  m = block_max(m);
  double ss = 0.0;
  for (int i = tid; i < n; i += 256) {
    const double e = exp(sU[i] - m);
    sA[i] = e;
    ss += e;
  }
  ss = block_sum(ss);
  double dot = 0.0;
  for (int i = tid; i < n; i += 256) {
    const double at = sA[i] / ss;
    sA[i] = at;
    dot += at * sD[i];
  }
  dot = block_sum(dot);
  double dotu = 0.0;
  for (int i = tid; i < n; i += 256) {
    const double du = sA[i] * (sD[i] - dot);
    sD[i] = du;
    dotu += sU[i] * du;
  }
  dotu = block_sum(dotu);
  if (use_ga_norm)
    for (int i = tid; i < n; i += 256) sD[i] = (sD[i] - sU[i] * dotu) / nrm;
  __syncthreads();
  for (int t = tid; t < n * dg; t += 256) {
    const int i = t / dg, k = t - i * dg;
    double sq = 0.0, sk = 0.0;
    for (int j = 0; j < n; ++j) {
      if (j == i) continue;
      sq += (double)gq[(size_t)(a0 + j) * dg + k];
      sk = fma(sD[j], (double)gk[(size_t)(a0 + j) * dg + k], sk);
    }
    dgk[(size_t)(a0 + i) * dg + k] = (float)(sA[i] * (double)dr[k] + sD[i] * sq);
    dgq[(size_t)(a0 + i) * dg + k] = (float)sk;
  }
}

// out[a][k] = [acc[a][k]] + sum over a's own edges of S_out[e][k] + sum over the edges with neighbour a of (S_in[e][k] + P_a[e][k] P_b[e][k])
__global__ void gen_edge_to_atom_kernel(const int32_t* __restrict__ edge_offset, const int32_t* __restrict__ in_off, const int32_t* __restrict__ in_edge,
                                        const float* __restrict__ S_out, const float* __restrict__ S_in, const float* __restrict__ P_a,
                                        const float* __restrict__ P_b, const float* __restrict__ acc, int n_atom, int d, float* __restrict__ out) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_atom * d) return;
  const int a = (int)(i / (size_t)d), k = (int)(i - (size_t)a * d);
  float s = acc ? acc[i] : 0.f;
  if (S_out)
    for (int e = edge_offset[a]; e < edge_offset[a + 1]; ++e) s += S_out[(size_t)e * d + k];
  if (S_in || P_a)
    for (int t = in_off[a]; t < in_off[a + 1]; ++t) {
      const size_t j = (size_t)in_edge[t] * d + k;
      if (S_in) s += S_in[j];
      if (P_a) s += P_a[j] * P_b[j];
    }
  out[i] = s;
}

// Embedding gradient, two stages.  part[chunk][z][:] = sum over the atoms of species z among the chunk's 64 atoms of dV[a][:], atoms in
// order (a wave per (species, chunk): one ballot over the chunk's atomic numbers, lanes over the columns); then
// dTable[z][:] += the chunks' sums in chunk order.
__global__ __launch_bounds__(64) void gen_table_part_kernel(const int32_t* __restrict__ atomic, int n_atom, const float* __restrict__ dV, int emb, int n_species,
                                                           float* __restrict__ part) {
  const int z = blockIdx.x, a0 = blockIdx.y * 64, lane = threadIdx.x;
  const unsigned long long m0 = __ballot(a0 + lane < n_atom && atomic[a0 + lane] == z);
  float* out = part + ((size_t)blockIdx.y * n_species + z) * emb;
  for (int k = lane; k < emb; k += 64) {
    float s = 0.f;
    unsigned long long m = m0;
    while (m) {
      const int a = a0 + __ffsll((long long)m) - 1;
      m &= m - 1;
      s += dV[(size_t)a * emb + k];
    }
    out[k] = s;
  }
}
__global__ void gen_table_final_kernel(const float* __restrict__ part, int n_chunk, int n, float* __restrict__ dTable) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dTable[i] += gt_ordered_sum(part + i, n_chunk, (size_t)n);
}

}  // namespace

void launch_gen_transpose(const GenTransDesc* descs, int n_desc, int max_elems, const float* W, float* WT, hipStream_t s) {
  if (n_desc <= 0 || max_elems <= 0) return;
  hipLaunchKernelGGL(gen_transpose_kernel, dim3((max_elems + 255) / 256, n_desc), dim3(256), 0, s, descs, W, WT);
}
void launch_gen_act_bwd(const float* dY, const float* pre, const float* row_scale, int rows, int N, float drop_p, unsigned drop_tag,
                        unsigned long long drop_seed, float* out, hipStream_t s) {
  const size_t n = (size_t)rows * N;
  if (!n) return;
  hipLaunchKernelGGL(gen_act_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dY, pre, row_scale, n, N, drop_p, drop_tag, drop_seed, out);
}
void launch_gen_mul_gather(const float* a, const float* B, const int32_t* idx, const float* c, int rows, int N, float* out, hipStream_t s) {
  const size_t n = (size_t)rows * N;
  if (!n) return;
  hipLaunchKernelGGL(gen_mul_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, B, idx, c, n, N, out);
}
void launch_gen_relu(float* y, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gen_relu_kernel, dim3((n + 255) / 256), dim3(256), 0, s, y, n);
}
// slabs of a weight gradient over `rows` rows: enough workgroups to fill the chip (~1024 with the K x N tiles), at least one 32-row
// chunk per slab, and <= 32 Mi floats of partial tiles
int gen_dw_slabs(int rows, int K, int N) {
  const int tiles = ((K + 31) / 32) * ((N + 31) / 32), chunks = (rows + 31) / 32;
  int s = std::max(1, std::min(std::min(64, chunks), (1024 + tiles - 1) / tiles));
  while (s > 1 && (size_t)s * ((size_t)K * N + N) > ((size_t)32 << 20)) --s;
  return s;
}
size_t gen_dw_part_floats(int rows, int K, int N) { return (size_t)gen_dw_slabs(rows, K, N) * ((size_t)K * N + N); }
void launch_gen_dense_dw(const GenDwArgs& a0, hipStream_t s) {
  if (a0.rows <= 0 || a0.K <= 0 || a0.N <= 0) return;
  GenDwArgs a = a0;
  const int n_slab = gen_dw_slabs(a.rows, a.K, a.N);
  a.slab_rows = (((a.rows + n_slab - 1) / n_slab) + 31) / 32 * 32;
  hipLaunchKernelGGL(gen_dense_dw_kernel, dim3((a.K + 31) / 32, (a.N + 31) / 32, n_slab), dim3(256), 0, s, a);
  const int n = a.K * a.N + (a.db ? a.N : 0);
  hipLaunchKernelGGL(gen_dw_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a.part, n_slab, a.K * a.N, a.N, a.dW, a.db);
}
int gen_ln_chunks(int rows) { return std::max(1, std::min(512, (rows + 63) / 64)); }
void launch_gen_layernorm_bwd(const float* X, const float* res, const float* gamma, const float* dY, int rows, int N, float* dX, float* stats,
                              float* part, float* dgamma, float* dbeta, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(gen_layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, X, res, gamma, dY, rows, N, dX, stats);
  const int n_chunk = gen_ln_chunks(rows), per = (rows + n_chunk - 1) / n_chunk;
  hipLaunchKernelGGL(gen_ln_param_kernel, dim3((N + 63) / 64, n_chunk), dim3(64), 0, s, X, res, dY, stats, rows, N, per, part);
  hipLaunchKernelGGL(gen_ln_param_final_kernel, dim3((N + 63) / 64), dim3(64), 0, s, part, n_chunk, N, dgamma, dbeta);
}
void launch_gen_attn_bwd(const float* q, const float* K, const int32_t* edge_offset, int n_atom, int d, int H, int max_degree, const float* dctx,
                         float drop_p, unsigned drop_tag, unsigned long long drop_seed, float* dq, float* dK, hipStream_t s) {
  if (n_atom <= 0) return;
  const size_t lds = (size_t)3 * std::max(1, max_degree) * H * sizeof(float);
  hipLaunchKernelGGL(gen_attn_bwd_kernel, dim3(n_atom), dim3(256), lds, s, q, K, edge_offset, d, H, dctx, drop_p, drop_tag, drop_seed, dq, dK);
}
void launch_gen_pool_bwd(const int32_t* mol_offset, int n_struct, int max_atoms, const float* gq, const float* gk, int dg, int use_ga_norm,
                         const float* drep, float* dgq, float* dgk, hipStream_t s) {
  if (n_struct <= 0) return;
  const size_t lds = ((size_t)3 * max_atoms + 4) * sizeof(double);
  hipLaunchKernelGGL(gen_pool_bwd_kernel, dim3(n_struct), dim3(256), lds, s, mol_offset, gq, gk, dg, use_ga_norm, drep, dgq, dgk);
}
void launch_gen_edge_to_atom(const int32_t* edge_offset, const int32_t* in_off, const int32_t* in_edge, const float* S_out, const float* S_in,
                             const float* P_a, const float* P_b, const float* acc, int n_atom, int d, float* out, hipStream_t s) {
  const size_t n = (size_t)n_atom * d;
  if (!n) return;
  hipLaunchKernelGGL(gen_edge_to_atom_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, edge_offset, in_off, in_edge, S_out, S_in, P_a, P_b, acc,
                     n_atom, d, out);
}
void launch_gen_table_grad(const int32_t* atomic, int n_atom, const float* dV, int emb, int n_species, float* part, float* dTable, hipStream_t s) {
  if (n_atom <= 0) return;
  const int n_chunk = (n_atom + 63) / 64, n = n_species * emb;  // part: n_chunk * n_species * emb floats
  hipLaunchKernelGGL(gen_table_part_kernel, dim3(n_species, n_chunk), dim3(64), 0, s, atomic, n_atom, dV, emb, n_species, part);
  hipLaunchKernelGGL(gen_table_final_kernel, dim3((n + 255) / 256), dim3(256), 0, s, part, n_chunk, n, dTable);
}

}  // namespace scann
