// Generic-width forward: create_model (scann_model.py:329-453) for ANY local_dim / num_head / global_dim / dense_out the reference
// accepts (scann_model.py:330-434 reads them from the yaml; every shipped config is 128 / 8 / 128 / 128 and runs on the MFMA kernels of
// scann_kernels.hip, which are written for exactly those).  A handle created with other widths evaluates here: plain fp32 FMA kernels on
// the same packed CSR batch, one formula of the reference per kernel, no matrix instructions, no tuning -- the point is that such a
// checkpoint evaluates on the GPU at all, with the reference's arithmetic (fp32 products, fp32 sums); it is one to two orders of
// magnitude slower than the 128-wide path.  Its training counterpart (the same kernels keeping their pre-activations and applying the
// Dropout layers, and one plain kernel per backward formula) is scann_generic_train.hip.
//
//   gen_dense_kernel      y = act(x . W + b) [+ residual] [* row scale], x assembled per row from up to three gathered segments
//                         (concat[c_i, g_ij, c_j], attention.py:142-150) or as the product of two (c_j * g'_ij, :157)
//   gen_layernorm_kernel  LayerNormalization(epsilon=1e-6) over the last axis of x [+ residual]        (attention.py:40,152,214)
//   gen_gauss_kernel      GaussianExpansion (custom_layers.py:55-65)
//   gen_attn_kernel       per atom: scaled logits per head, softmax over its neighbours, context + unscaled query (attention.py:170-212)
//   gen_readout_kernel    per structure: GlobalAttention pooling, bf_property, predict_property, mrelu (attention.py:279-316,
//                         scann_model.py:437-447)
#include "scann_internal.h"
#include "scann_mma.h"

namespace scann {

namespace {

// GR rows per workgroup: every weight element fetched serves GR rows (8 while the staged rows fit 48 KB of LDS, else 4).  A thread
// owns RPT of them for one output column at a time, so that narrow layers (N <= 128, <= 64) still use all 256 threads: the workgroup
// covers 256 / (GR / RPT) columns per pass.  A row's sum over k is the same sequence in every shape (four interleaved partial sums).
template <int GR, int RPT>
__global__ __launch_bounds__(256) void gen_dense_kernel(GenDenseArgs a) {
#pragma clang fp contract(off)
  extern __shared__ float sX[];  // [GR][Ks], Ks = K rounded up to 4: 16-byte fragment reads
  const int r0 = blockIdx.x * GR, tid = threadIdx.x;
  const int Ks = (a.K + 3) & ~3;
  for (int i = tid; i < GR * a.K; i += 256) {
    const int rr = i / a.K, k = i - rr * a.K;
    const int r = min(r0 + rr, a.rows - 1);
    float v;
    if (a.prod) {
      const float x0 = a.seg[0].p[(size_t)(a.seg[0].idx ? a.seg[0].idx[r] : r) * a.K + k];
      const float x1 = a.seg[1].p[(size_t)(a.seg[1].idx ? a.seg[1].idx[r] : r) * a.K + k];
      v = x0 * x1;
    } else {
      int kk = k, s = 0;
      while (s < a.n_seg - 1 && kk >= a.seg[s].w) {
        kk -= a.seg[s].w;
        ++s;
      }
      v = a.seg[s].p[(size_t)(a.seg[s].idx ? a.seg[s].idx[r] : r) * a.seg[s].w + kk];
    }
    sX[rr * Ks + k] = v;
  }
  __syncthreads();
  const int K4 = a.K & ~3;
  constexpr int COLS = 256 / (GR / RPT);
  const int rb = (tid / COLS) * RPT;  // this thread's first staged row
  for (int o = tid % COLS; o < a.N; o += COLS) {
#if defined(SCANN_DIAG_DENSE64)  // diagnostic: exact products, fp64 sums, one rounding per output
    double acc[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) acc[rr] = 0.0;
    const float* __restrict__ wp = a.W + o;
    for (int k = 0; k < a.K; ++k) {
      const double w = wp[(size_t)k * a.N];
#pragma unroll
      for (int rr = 0; rr < RPT; ++rr) acc[rr] = fma((double)sX[(rb + rr) * Ks + k], w, acc[rr]);
    }
    const double bias = a.b ? a.b[o] : 0.f;
#elif defined(SCANN_DIAG_DENSE1)  // diagnostic: round 5's single fmaf chain over k
    float acc[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) acc[rr] = 0.f;
    const float* __restrict__ wp = a.W + o;
    for (int k = 0; k < a.K; ++k) {
      const float w = wp[(size_t)k * a.N];
#pragma unroll
      for (int rr = 0; rr < RPT; ++rr) acc[rr] = fmaf(sX[(rb + rr) * Ks + k], w, acc[rr]);
    }
    const float bias = a.b ? a.b[o] : 0.f;
#else
    // four interleaved partial sums (k mod 4), combined pairwise at the end: one fixed summation order for every shape, dependent chains
    // a quarter as long as one running sum's, and a rounding error that grows with K / 4 + 2 additions instead of K (a 128-term fmaf
    // chain sat a factor ~3 above the blocked sums of a BLAS or MFMA product -- visible where the graph is ill-conditioned:
    // profiles/r06_notes.md, 'plain backward')
    float acc4[RPT][4];
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) acc4[rr][0] = acc4[rr][1] = acc4[rr][2] = acc4[rr][3] = 0.f;
    const float* __restrict__ wp = a.W + o;
    for (int k = 0; k < K4; k += 4) {
      const float w0 = wp[(size_t)k * a.N], w1 = wp[(size_t)(k + 1) * a.N], w2 = wp[(size_t)(k + 2) * a.N], w3 = wp[(size_t)(k + 3) * a.N];
#pragma unroll
      for (int rr = 0; rr < RPT; ++rr) {
        const float4 x = *reinterpret_cast<const float4*>(&sX[(rb + rr) * Ks + k]);
        acc4[rr][0] = fmaf(x.x, w0, acc4[rr][0]);
        acc4[rr][1] = fmaf(x.y, w1, acc4[rr][1]);
        acc4[rr][2] = fmaf(x.z, w2, acc4[rr][2]);
        acc4[rr][3] = fmaf(x.w, w3, acc4[rr][3]);
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {  // K mod 4 trailing terms: term K4 + t continues partial sum t (a compile-time index: registers)
      if (K4 + t < a.K) {
        const float w = wp[(size_t)(K4 + t) * a.N];
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) acc4[rr][t] = fmaf(sX[(rb + rr) * Ks + K4 + t], w, acc4[rr][t]);
      }
    }
    float acc[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) acc[rr] = (acc4[rr][0] + acc4[rr][1]) + (acc4[rr][2] + acc4[rr][3]);
    const float bias = a.b ? a.b[o] : 0.f;
#endif
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
      const int r = r0 + rb + rr;
      if (r >= a.rows) break;
      float y = (float)(acc[rr] + bias);
      if (a.pre) a.pre[(size_t)r * a.N + o] = y;  // training forward: the pre-activation the backward differentiates swish at
      if (a.act) y = swish_exact(y);
      if (a.drop_p > 0.f) y = y * drop_scale(a.drop_seed, a.drop_tag, (size_t)r * a.N + o, a.drop_p);  // Dropout on the layer's OUTPUT
      if (a.res) y = y + a.res[(size_t)(a.res_idx ? a.res_idx[r] : r) * a.N + o];
      if (a.row_scale) y = y * a.row_scale[r];
      a.Y[(size_t)r * a.N + o] = y;
    }
  }
}

// (xor butterflies of its own, 32 down to 1: the plain-fp32 path keeps its summation order whatever scann_mma.h does)
__device__ __forceinline__ float gen_wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float gen_wave_max64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// one wave per row: mean, then the mean of the squared deviations (two passes, as the fp32 restatement does), then the affine map
__global__ __launch_bounds__(256) void gen_layernorm_kernel(const float* __restrict__ X, const float* __restrict__ res, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int rows, int N, float* __restrict__ Y) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* x = X + (size_t)r * N;
  const float* q = res ? res + (size_t)r * N : nullptr;
  float s = 0.f;
  for (int k = lane; k < N; k += 64) s += q ? x[k] + q[k] : x[k];
  const float mean = gen_wave_sum64(s) / (float)N;
  float v = 0.f;
  for (int k = lane; k < N; k += 64) {
    const float d = (q ? x[k] + q[k] : x[k]) - mean;
    v += d * d;
  }
  const float rstd = 1.0f / sqrtf(gen_wave_sum64(v) / (float)N + 1e-6f);
  for (int k = lane; k < N; k += 64) Y[(size_t)r * N + k] = ((q ? x[k] + q[k] : x[k]) - mean) * rstd * gamma[k] + beta[k];
}

__global__ void gen_gauss_kernel(const float* __restrict__ x, const float* __restrict__ centres, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * NG) return;
  const int e = i / NG, k = i - e * NG;
  const float d = x[e] - centres[k];
  out[i] = expf(-(d * d) / 0.25f);  // width = 0.5 ** 2 (custom_layers.py:48-51)
}

__global__ void gen_mul_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
}

// one workgroup per atom.  logits[n][h] = (q[h] * hd^-0.5) . K[n][h] over the atom's CSR row, softmax over n per head, context[o] =
// sum_n attn[n][head(o)] K[n][o] + q[o] (the residual is the UNSCALED query, attention.py:198-212).  No edges: context = q.
__global__ __launch_bounds__(256) void gen_attn_kernel(const float* __restrict__ q, const float* __restrict__ K, const int32_t* __restrict__ edge_offset,
                                                       int n_atom, int d, int H, float* __restrict__ ctx, float drop_p, unsigned drop_tag,
                                                       unsigned long long drop_seed) {
#pragma clang fp contract(off)
  extern __shared__ float sL[];  // [deg][H] logits -> attention
  const int at = blockIdx.x, tid = threadIdx.x;
  const int e0 = edge_offset[at], deg = edge_offset[at + 1] - e0, hd = d / H;
  const float dk = 1.0f / sqrtf((float)hd);
  const float* qa = q + (size_t)at * d;
  for (int i = tid; i < deg * H; i += 256) {
    const int n = i / H, h = i - n * H;
    const float* kr = K + (size_t)(e0 + n) * d + h * hd;
    float s = 0.f;
    for (int j = 0; j < hd; ++j) s = fmaf(qa[h * hd + j] * dk, kr[j], s);
    sL[i] = s;
  }
  __syncthreads();
  for (int h = tid; h < H; h += 256) {
    float m = -INFINITY;
    for (int n = 0; n < deg; ++n) m = fmaxf(m, sL[n * H + h]);
    float ss = 0.f;
    for (int n = 0; n < deg; ++n) {
      const float e = expf(sL[n * H + h] - m);
      sL[n * H + h] = e;
      ss += e;
    }
    for (int n = 0; n < deg; ++n) {
      float at = sL[n * H + h] / ss;
      if (drop_p > 0.f) at = at * drop_scale(drop_seed, drop_tag, (size_t)(e0 + n) * H + h, drop_p);  // Dropout(0.05), attention.py:191 (training)
      sL[n * H + h] = at;
    }
  }
  __syncthreads();
  for (int o = tid; o < d; o += 256) {
    const int h = o / hd;
    float s = 0.f;
    for (int n = 0; n < deg; ++n) s = fmaf(sL[n * H + h], K[(size_t)(e0 + n) * d + o], s);
    ctx[(size_t)at * d + o] = s + qa[o];
  }
}

// one workgroup per structure (attention.py:279-316; scann_model.py:437-447)
__global__ __launch_bounds__(256) void gen_readout_kernel(const int32_t* __restrict__ mol_offset, const float* __restrict__ gq, const float* __restrict__ gk, int dg,
                                                          int dout, int use_ga_norm, int relu_out, const float* __restrict__ Wb, const float* __restrict__ bb,
                                                          const float* __restrict__ wo, const float* __restrict__ bo, float* __restrict__ ga_attn,
                                                          float* __restrict__ y, float* __restrict__ rep_out) {
#pragma clang fp contract(off)
  extern __shared__ float sm[];  // [n] scores -> attention, [dg] pooled rows, [dout] hidden, [4] reductions
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a0 = mol_offset[blockIdx.x], n = mol_offset[blockIdx.x + 1] - a0;
  float* sA = sm;
  float* sRep = sA + n;
  float* sHid = sRep + dg;
  float* sRed = sHid + dout;
  // agg_i = sum over j != i of k_i . q_j  (the literal form: no S - q_i cancellation) -- products and sums in fp64, rounded once.  The
  // score of a small structure can be a thousandth of its terms (a two-atom molecule with nearly orthogonal key and query rows:
  // |k_0 . q_1| = 0.03 against sum |terms| = 22, tools/debug_plain_grads.py), the normalisation divides by it, and gen_pool_bwd_kernel
  // forms the same scores again: in fp32 the two evaluations differ by 6e-5 of the score and the pooling gradient by ten times that.
  for (int i = tid; i < n; i += 256) sA[i] = (float)gen_pool_score(gq, gk, a0, n, i, dg);
  __syncthreads();
  auto block_sum = [&](float v) {
    v = gen_wave_sum64(v);
    if (lane == 0) sRed[wave] = v;
    __syncthreads();
    const float t = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
    __syncthreads();
    return t;
  };
  auto block_max = [&](float v) {
    v = gen_wave_max64(v);
    if (lane == 0) sRed[wave] = v;
    __syncthreads();
    const float t = fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3]));
    __syncthreads();
    return t;
  };
  if (use_ga_norm) {  // tf.linalg.normalize over the atoms: no epsilon (a one-atom structure is the reference's 0 / 0)
    float ss = 0.f;
    for (int i = tid; i < n; i += 256) ss += sA[i] * sA[i];
    const float nrm = sqrtf(block_sum(ss));
    for (int i = tid; i < n; i += 256) sA[i] = sA[i] / nrm;
    __syncthreads();
  }
  float m = -INFINITY;
  for (int i = tid; i < n; i += 256) m = fmaxf(m, sA[i]);
  m = block_max(m);
  float ss = 0.f;
  for (int i = tid; i < n; i += 256) {
    const float e = expf(sA[i] - m);
    sA[i] = e;
    ss += e;
  }
  ss = block_sum(ss);
  for (int i = tid; i < n; i += 256) {
    const float at = sA[i] / ss;
    sA[i] = at;
    ga_attn[a0 + i] = at;
  }
  __syncthreads();
  for (int o = tid; o < dg; o += 256) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s = fmaf(sA[i], gk[(size_t)(a0 + i) * dg + o], s);
    sRep[o] = s;
    if (rep_out) rep_out[(size_t)blockIdx.x * dg + o] = s;
  }
  if (rep_out) return;  // training forward: bf_property / predict_property run as dense launches that keep their pre-activations
  __syncthreads();
  float part = 0.f;
  for (int o = tid; o < dout; o += 256) {
    float s = 0.f;
    for (int k = 0; k < dg; ++k) s = fmaf(sRep[k], Wb[(size_t)k * dout + o], s);
    part += swish_exact(s + bb[o]) * wo[o];
  }
  part = block_sum(part);
  if (tid == 0) {
    float out = part + bo[0];
    if (relu_out) out = fmaxf(out, 0.f);  // mrelu forward (custom_layers.py:15)
    y[blockIdx.x] = out;
  }
}

}  // namespace

void launch_gen_dense(const GenDenseArgs& a, hipStream_t s) {
  if (a.rows <= 0) return;
  // rows per workgroup: 8 (4 when eight staged rows would not fit 48 KB of LDS).  Measured on the 128 / 8 config: 4 -> 8 rows +7 %;
  // 32 rows -25 % (the loop is bound by the broadcast LDS reads of x, not by the weight stream from L2); this is the plain path, not
  // a tuned SGEMM
  const size_t Ks = (size_t)((a.K + 3) & ~3);
  const int gr = 8 * Ks * 4 <= 48 * 1024 ? 8 : 4;
  const int ngrp = a.N <= 64 ? 4 : a.N <= 128 ? 2 : 1;  // row groups side by side: 256 / ngrp columns per pass
  const dim3 grid((a.rows + gr - 1) / gr), block(256);
  const size_t lds = gr * Ks * sizeof(float);
#define SCANN_GEN_CASE(GR_)                                                                     \
  do {                                                                                          \
    if (ngrp == 4) hipLaunchKernelGGL((gen_dense_kernel<GR_, GR_ / 4>), grid, block, lds, s, a);  \
    else if (ngrp == 2) hipLaunchKernelGGL((gen_dense_kernel<GR_, GR_ / 2>), grid, block, lds, s, a); \
    else hipLaunchKernelGGL((gen_dense_kernel<GR_, GR_>), grid, block, lds, s, a);               \
  } while (0)
  if (gr == 8) SCANN_GEN_CASE(8);
  else SCANN_GEN_CASE(4);
#undef SCANN_GEN_CASE
}
void launch_gen_layernorm(const float* X, const float* res, const float* gamma, const float* beta, int rows, int N, float* Y, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(gen_layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, X, res, gamma, beta, rows, N, Y);
}
void launch_gen_gauss(const float* x, const float* centres, int n, float* out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gen_gauss_kernel, dim3((n * NG + 255) / 256), dim3(256), 0, s, x, centres, n, out);
}
void launch_gen_mul(const float* a, const float* b, size_t n, float* out, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(gen_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, n, out);
}
void launch_gen_attn(const float* q, const float* K, const int32_t* edge_offset, int n_atom, int d, int H, int max_degree, float* ctx, hipStream_t s,
                     float drop_p, unsigned drop_tag, unsigned long long drop_seed) {
  if (n_atom <= 0) return;
  hipLaunchKernelGGL(gen_attn_kernel, dim3(n_atom), dim3(256), (size_t)std::max(1, max_degree) * H * sizeof(float), s, q, K, edge_offset, n_atom, d, H, ctx,
                     drop_p, drop_tag, drop_seed);
}
void launch_gen_readout(const int32_t* mol_offset, int n_struct, int max_atoms, const float* gq, const float* gk, int dg, int dout, int use_ga_norm,
                        int relu_out, const float* Wb, const float* bb, const float* wo, const float* bo, float* ga_attn, float* y, hipStream_t s,
                        float* rep_out) {
  if (n_struct <= 0) return;
  const size_t lds = ((size_t)max_atoms + dg + dout + 4) * sizeof(float);
  hipLaunchKernelGGL(gen_readout_kernel, dim3(n_struct), dim3(256), lds, s, mol_offset, gq, gk, dg, dout, use_ga_norm, relu_out, Wb, bb, wo, bo, ga_attn, y,
                     rep_out);
}

}  // namespace scann
