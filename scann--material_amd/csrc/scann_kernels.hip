// Hand-written CDNA4 (gfx950) kernels for the SCANN / SCANN+ forward hot path.
//
// Layout: structures are packed -- atoms [n_atom,128] and CSR edges [n_edge,128] -- so
// the reference's padded [B,M,N,d] tensors (attention.py:136-212) are never materialised.  Every
// dense projection is an fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32, one k-ordered fma chain)
// of a row tile staged in LDS against a 128x128 weight that each wave streams straight from
// L2/HBM into registers in fragment order (no LDS staging of weights: a wave owns 32 output columns,
// so every weight element is read by exactly one wave of the workgroup).
//
// Kernels (one launch each per LocalAttention iteration, see scann_runtime.cpp: run_forward):
//   atom_kernel   : [ResidualNorm of previous layer] -> centres c; P1 = c W1 + bg, P3 = c W3, q = c Wq + bq
//                   (attention.py:37-40, :160, and the centre/neighbour thirds of filter_geo :142-151)
//   edge_kernel   : per tile of <=64 edges (whole atoms): U = G W2 (MFMA); geom' = LN_g(swish(U+P1[i]+P3[j])+G)
//                   (:141-153); ang = c[j]*geom' (:136,:157); K = ang Wk + bk (MFMA, :163);
//                   per (atom, head) softmax over its edges and ctx = LN(sum attn K + q) (:180-214)
//   readout_kernel: GlobalAttention + bf_property + predict_property (attention.py:267-318,
//                   scann_model.py:437-447), one workgroup per structure
//   basis_kernel  : Gaussian expansion + neighbor_d/neighbor_w MLP (custom_layers.py:63-65,
//                   scann_model.py:378-389)
#include "scann_internal.h"

namespace scann {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Diagnostic build only (-DSCANN_STAMPS): per-workgroup phase timestamps, written to a buffer nothing else reads.
#ifdef SCANN_STAMPS
#define STAMP(buf, slot)                                                                      \
  do {                                                                                        \
    if ((buf) && threadIdx.x == 0) {                                                          \
      unsigned long long t_;                                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
      (buf)[(size_t)blockIdx.x * 16 + (slot)] = t_;                                            \
    }                                                                                         \
  } while (0)
#define STAMP_IF(buf, slot, cond)                                                             \
  do {                                                                                        \
    if ((buf) && (cond)) {                                                                    \
      unsigned long long t_;                                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
      (buf)[(size_t)blockIdx.x * 16 + (slot)] = t_;                                            \
    }                                                                                         \
  } while (0)
#define STAMP_REAL(buf, slot)                                                                 \
  do {                                                                                        \
    if ((buf) && threadIdx.x == 0) {                                                          \
      unsigned long long t_;                                                                  \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      (buf)[(size_t)blockIdx.x * 16 + (slot)] = t_;                                            \
    }                                                                                         \
  } while (0)
#else
#define STAMP(buf, slot) do {} while (0)
#define STAMP_IF(buf, slot, cond) do {} while (0)
#define STAMP_REAL(buf, slot) do {} while (0)
#endif

// swish(x) = x * sigmoid(x) on the hardware transcendental units: v_exp_f32 (2^x) and v_rcp_f32, 1 ulp each.
// Measured effect on the end-to-end parity error: DESIGN.md "numerics".
__device__ __forceinline__ float swishf(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

__device__ __forceinline__ float swish_exact(float x) { return x * (1.0f / (1.0f + expf(-x))); }

// e^x through v_exp_f32 (2^x, 1 ulp); used where the argument is <= 0 (softmax numerators).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4swish(float4 a) { return make_float4(swishf(a.x), swishf(a.y), swishf(a.z), swishf(a.w)); }
__device__ __forceinline__ float f4sum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

// ---- MFMA GEMM of a staged row tile against one packed 128x128 weight ---------------------------
//
// acc[rt] (32 rows x 32 cols per wave) += X[32*rt .. 32*rt+31][0..127] . W[0..127][32*wave .. +31]
//
// v_mfma_f32_32x32x2_f32 operand map: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].
// The k order is ours to choose as long as A and B agree: k-step (t, i), lane half h <-> k = 8t + 4h + i,
// so a lane's A operands for four consecutive steps are one 16-byte LDS read and its B operands one
// 16-byte global read.  Packed weight element ((w*16 + t)*64 + lane)*4 + i = W[8t + 4(lane>>5) + i][32w + (lane&31)].
// load_w: one wave's 128x32 slab of a packed weight, 16 coalesced 1-KiB reads, issued early so the
// L2/HBM latency hides under whatever precedes the MFMAs.
__device__ __forceinline__ void load_w(const float* __restrict__ Wp, int wave, int lane, float4 (&w)[16]) {
  const float4* __restrict__ wsrc = reinterpret_cast<const float4*>(Wp) + wave * (16 * 64) + lane;
#pragma unroll
  for (int t = 0; t < 16; ++t) w[t] = wsrc[t * 64];
}

template <int RT>
__device__ __forceinline__ void mma128(const float* __restrict__ sX, const float4 (&w)[16], int lane,
                                       f32x16 (&acc)[RT]) {
  const float* xrow = sX + (lane & 31) * LDS_STRIDE + 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float4 a = *reinterpret_cast<const float4*>(xrow + rt * 32 * LDS_STRIDE + 8 * t);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w[t].x, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w[t].y, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w[t].z, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w[t].w, acc[rt], 0, 0, 0);
    }
  }
}

template <int RT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[RT]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[rt][i] = 0.0f;
}

// Operands swapped (weights as A, rows as B): the product comes out transposed -- lane l holds ROW l & 31 and the sixteen
// columns (i & 3) + 8 (i >> 2) + 4 (l >> 5) of the wave's 32, four runs of four consecutive columns -- so every epilogue
// moves 16-byte pieces (ds_write_b128 / global_store_dwordx4) instead of sixteen 4-byte ones.  Same k order, same sums.
__device__ __forceinline__ void mma128T(const float* __restrict__ sX, const float4 (&w)[16], int lane, f32x16& acc) {
  const float* xrow = sX + (lane & 31) * LDS_STRIDE + 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(xrow + 8 * t);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].x, a.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].y, a.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].z, a.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].w, a.w, acc, 0, 0, 0);
  }
}
// column of the j-th run of a lane in the transposed layout
__device__ __forceinline__ int tcol(int wave, int lane, int j) { return 32 * wave + 8 * j + 4 * (lane >> 5); }

// C/D map of the 32x32 MFMA: register i of lane l holds row (i&3) + 8*(i>>2) + 4*(l>>5), column l&31.
__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// ---- atom-tile kernel ------------------------------------------------------------------------------

// One weight slab (64 VGPRs) is live at a time: each slab is requested right after the previous GEMM's MFMAs and pinned
// there with sched_barrier (hipcc otherwise hoists it above them and keeps two slabs live: 202 VGPRs, 2 waves/SIMD).
// 158 VGPRs -> 3 waves per SIMD = 3 workgroups per CU; +1.7 % end to end.  (4 waves/SIMD spills.)
template <bool FFN, int MODE>
__global__ __launch_bounds__(256, 3) void atom_kernel(AtomArgs a) {
  __shared__ __attribute__((aligned(16))) float sX[TA * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float sH[TA * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float sLN[2 * D];  // ResidualNorm LayerNorm gamma, beta
  __shared__ __attribute__((aligned(16))) float sBias[5 * D];  // rows: bf1, bf2, bA, bC, bD
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * TA;
  const int nrows = min(TA, a.n_atom - row0);
  const int trow = lane & 31;  // this lane's row in the transposed accumulator layout (mma128T)
  // first projection after the (optional) ResidualNorm: W1 (mode 0), Wq (mode 1), after_Lc (mode 2)
  const float* const firstW = MODE == 1 ? a.WCp : a.WAp;

  STAMP(a.stamps, 0);
  // vmcnt retires in issue order: a bias or LayerNorm parameter requested after a weight slab waits for the whole slab,
  // and one requested after output stores waits for their acknowledgements.  So every small operand is fetched here, first.
  {
    const int j = tid & (D - 1);
    const bool lo = tid < D;
    const float v0 = FFN ? (lo ? a.bf1 : a.bf2)[j] : 0.f;
    const float v1 = MODE == 1 ? a.bC[j] : (lo ? a.bA : a.bC)[j];
    const float v2 = MODE == 2 ? a.bD[j] : 0.f;
    const float v3 = FFN ? (lo ? a.lnr_g : a.lnr_b)[j] : 0.f;
    sBias[tid] = v0;                              // bf1 | bf2
    sBias[2 * D + tid] = v1;                      // bA | bC   (mode 1: bC in both halves)
    if (lo) sBias[4 * D + j] = v2;                // bD
    sLN[tid] = v3;
  }
  float4 wA[16];
  if (FFN) load_w(a.Wf1p, wave, lane, wA);
  else load_w(firstW, wave, lane, wA);

  // stage x rows (zero-fill the ragged tail so the MFMAs see defined data).  The four row loads of a thread are issued
  // together from clamped rows and masked afterwards: a load under a per-thread guard costs a full memory round trip each.
  {
    const int c4 = tid & 31, r0 = tid >> 5;  // rows r0, r0 + 8, r0 + 16, r0 + 24
    int src[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rc = min(r0 + 8 * k, nrows - 1);
      src[k] = a.x_index ? a.x_index[row0 + rc] : (row0 + rc);
    }
    float4 xv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = reinterpret_cast<const float4*>(a.x)[(size_t)src[k] * 32 + c4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + 8 * k;
      float4 v = r < nrows ? xv[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < nrows) {
        if (!FFN && a.drop_p > 0.f) {  // training: Dropout after dense_embed (scann_model.py:374)
          const size_t e = (size_t)(row0 + r) * D + 4 * c4;
          v.x *= drop_scale(a.drop_seed, a.drop_tag, e, a.drop_p);
          v.y *= drop_scale(a.drop_seed, a.drop_tag, e + 1, a.drop_p);
          v.z *= drop_scale(a.drop_seed, a.drop_tag, e + 2, a.drop_p);
          v.w *= drop_scale(a.drop_seed, a.drop_tag, e + 3, a.drop_p);
        }
        if (!FFN) reinterpret_cast<float4*>(a.c)[(size_t)(row0 + r) * 32 + c4] = v;  // centres = staged rows (layer 0 / no ResidualNorm)
      }
      *reinterpret_cast<float4*>(&sX[r * LDS_STRIDE + 4 * c4]) = v;
    }
  }
  __syncthreads();
  STAMP(a.stamps, 1);

  f32x16 acc;
  if (FFN) {
    // ResidualNorm (attention.py:37-40): h = swish(x W1 + b1)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    mma128T(sX, wA, lane, acc);
    STAMP(a.stamps, 2);
    __builtin_amdgcn_sched_barrier(0);
    load_w(a.Wf2p, wave, lane, wA);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tcol(wave, lane, j);
      const float4 bv = *reinterpret_cast<const float4*>(&sBias[c]);
      const float4 pre = make_float4(acc[4 * j] + bv.x, acc[4 * j + 1] + bv.y, acc[4 * j + 2] + bv.z, acc[4 * j + 3] + bv.w);
      const float4 hh = make_float4(swishf(pre.x), swishf(pre.y), swishf(pre.z), swishf(pre.w));
      *reinterpret_cast<float4*>(&sH[trow * LDS_STRIDE + c]) = hh;
      if (a.keep_pre1 && trow < nrows) {  // training forward: kept for the backward
        *reinterpret_cast<float4*>(a.keep_pre1 + (size_t)(row0 + trow) * D + c) = pre;
        *reinterpret_cast<float4*>(a.keep_H1 + (size_t)(row0 + trow) * D + c) = hh;
      }
    }
    __syncthreads();
    STAMP(a.stamps, 3);
    // y = h W2 + b2 ; t = x + y
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    mma128T(sH, wA, lane, acc);
    STAMP(a.stamps, 4);
    __builtin_amdgcn_sched_barrier(0);
    load_w(firstW, wave, lane, wA);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // every wave is done reading sH
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tcol(wave, lane, j);
      const float4 bv = *reinterpret_cast<const float4*>(&sBias[D + c]);
      const float4 xv = *reinterpret_cast<const float4*>(&sX[trow * LDS_STRIDE + c]);
      float4 y = make_float4(acc[4 * j] + bv.x, acc[4 * j + 1] + bv.y, acc[4 * j + 2] + bv.z, acc[4 * j + 3] + bv.w);
      if (a.drop_p > 0.f) {  // attention.py:29 (training)
        const size_t e = (size_t)(row0 + trow) * D + c;
        y.x *= drop_scale(a.drop_seed, a.drop_tag, e, a.drop_p);
        y.y *= drop_scale(a.drop_seed, a.drop_tag, e + 1, a.drop_p);
        y.z *= drop_scale(a.drop_seed, a.drop_tag, e + 2, a.drop_p);
        y.w *= drop_scale(a.drop_seed, a.drop_tag, e + 3, a.drop_p);
      }
      const float4 t2 = f4add(xv, y);
      *reinterpret_cast<float4*>(&sH[trow * LDS_STRIDE + c]) = t2;
      if (a.keep_T2 && trow < nrows) *reinterpret_cast<float4*>(a.keep_T2 + (size_t)(row0 + trow) * D + c) = t2;
    }
    __syncthreads();
    STAMP(a.stamps, 5);
    // c = LayerNorm(t): 8 threads per row, 4 float4 each
    {
      const int r = tid >> 3, sub = tid & 7;
      float4 t[4];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t[i] = *reinterpret_cast<const float4*>(&sH[r * LDS_STRIDE + 4 * (sub + 8 * i)]);
        s += f4sum(t[i]);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      const float mean = s * (1.0f / D);
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dx = t[i].x - mean, dy = t[i].y - mean, dz = t[i].z - mean, dw = t[i].w - mean;
        v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c4 = sub + 8 * i;
        const float4 g = *reinterpret_cast<const float4*>(&sLN[4 * c4]);
        const float4 be = *reinterpret_cast<const float4*>(&sLN[D + 4 * c4]);
        float4 y;
        float inv;
        inv = rstd * g.x; y.x = t[i].x * inv + (be.x - mean * inv);
        inv = rstd * g.y; y.y = t[i].y * inv + (be.y - mean * inv);
        inv = rstd * g.z; y.z = t[i].z * inv + (be.z - mean * inv);
        inv = rstd * g.w; y.w = t[i].w * inv + (be.w - mean * inv);
        *reinterpret_cast<float4*>(&sX[r * LDS_STRIDE + 4 * c4]) = y;
        if (r < nrows) reinterpret_cast<float4*>(a.c)[(size_t)(row0 + r) * 32 + c4] = y;
      }
    }
    __syncthreads();
    STAMP(a.stamps, 6);
  }

  // The projections stay in registers until the last weight slab has been requested; their stores go out together at
  // the end (16-byte pieces) so that no load of this workgroup ever queues behind a store acknowledgement.
  f32x16 accP1, accP3;
  if (MODE == 0) {  // P1 = c W1 + bg ; P3 = c W3 ; q = c Wq + bq (attention.py:142-151 thirds, :160)
#pragma unroll
    for (int i = 0; i < 16; ++i) accP1[i] = 0.f;
    mma128T(sX, wA, lane, accP1);
    STAMP(a.stamps, 7);
    __builtin_amdgcn_sched_barrier(0);
    load_w(a.WBp, wave, lane, wA);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(a.stamps, 8);
#pragma unroll
    for (int i = 0; i < 16; ++i) accP3[i] = 0.f;
    mma128T(sX, wA, lane, accP3);
    STAMP(a.stamps, 9);
    __builtin_amdgcn_sched_barrier(0);
    load_w(a.WCp, wave, lane, wA);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (MODE == 0 || MODE == 1) {  // q = c Wq + bq (attention.py:160)
    STAMP(a.stamps, 10);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    mma128T(sX, wA, lane, acc);
    STAMP(a.stamps, 11);
    __builtin_amdgcn_sched_barrier(0);
    if (trow < nrows) {
      const size_t o = (size_t)(row0 + trow) * D;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = tcol(wave, lane, j);
        const float4 bq = *reinterpret_cast<const float4*>(&sBias[(MODE == 0 ? 3 : 2) * D + c]);
        if (MODE == 0) {
          const float4 bg = *reinterpret_cast<const float4*>(&sBias[2 * D + c]);
          *reinterpret_cast<float4*>(a.oA + o + c) = make_float4(accP1[4 * j] + bg.x, accP1[4 * j + 1] + bg.y, accP1[4 * j + 2] + bg.z, accP1[4 * j + 3] + bg.w);
          *reinterpret_cast<float4*>(a.oB + o + c) = make_float4(accP3[4 * j], accP3[4 * j + 1], accP3[4 * j + 2], accP3[4 * j + 3]);
        }
        *reinterpret_cast<float4*>(a.oC + o + c) = make_float4(acc[4 * j] + bq.x, acc[4 * j + 1] + bq.y, acc[4 * j + 2] + bq.z, acc[4 * j + 3] + bq.w);
      }
    }
    STAMP(a.stamps, 12);
  }
  if (MODE == 2) {  // z = swish(c Wa + ba) (scann_model.py:424); gq = z Wgq + b ; gk = z Wgk + b (attention.py:269-272)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    mma128T(sX, wA, lane, acc);
    __builtin_amdgcn_sched_barrier(0);
    load_w(a.WCp, wave, lane, wA);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tcol(wave, lane, j);
      const float4 bv = *reinterpret_cast<const float4*>(&sBias[2 * D + c]);
      *reinterpret_cast<float4*>(&sH[trow * LDS_STRIDE + c]) =
          make_float4(swishf(acc[4 * j] + bv.x), swishf(acc[4 * j + 1] + bv.y), swishf(acc[4 * j + 2] + bv.z), swishf(acc[4 * j + 3] + bv.w));
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) accP1[i] = 0.f;
    mma128T(sH, wA, lane, accP1);
    __builtin_amdgcn_sched_barrier(0);
    load_w(a.WDp, wave, lane, wA);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    mma128T(sH, wA, lane, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (trow < nrows) {
      const size_t o = (size_t)(row0 + trow) * D;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = tcol(wave, lane, j);
        const float4 bq = *reinterpret_cast<const float4*>(&sBias[3 * D + c]);
        const float4 bk = *reinterpret_cast<const float4*>(&sBias[4 * D + c]);
        *reinterpret_cast<float4*>(a.oC + o + c) = make_float4(accP1[4 * j] + bq.x, accP1[4 * j + 1] + bq.y, accP1[4 * j + 2] + bq.z, accP1[4 * j + 3] + bq.w);
        *reinterpret_cast<float4*>(a.oB + o + c) = make_float4(acc[4 * j] + bk.x, acc[4 * j + 1] + bk.y, acc[4 * j + 2] + bk.z, acc[4 * j + 3] + bk.w);
      }
    }
  }
}

void launch_atom(const AtomArgs& a, hipStream_t s) {
  if (a.n_atom <= 0) return;
  const dim3 grid((a.n_atom + TA - 1) / TA), block(256);
#define SCANN_ATOM_CASE(F, M) hipLaunchKernelGGL((atom_kernel<F, M>), grid, block, 0, s, a)
  if (a.ffn) {
    if (a.mode == 0) SCANN_ATOM_CASE(true, 0);
    else if (a.mode == 1) SCANN_ATOM_CASE(true, 1);
    else SCANN_ATOM_CASE(true, 2);
  } else {
    if (a.mode == 0) SCANN_ATOM_CASE(false, 0);
    else if (a.mode == 1) SCANN_ATOM_CASE(false, 1);
    else SCANN_ATOM_CASE(false, 2);
  }
#undef SCANN_ATOM_CASE
}

// ---- edge-tile kernel ------------------------------------------------------------------------------

#ifndef EDGE_OCC32
#define EDGE_OCC32 3  // waves per SIMD requested for the 32-row tile variant
#endif
// RT = 32-row MFMA row tiles per edge tile (tile holds <= 32*RT edges of whole atoms, <= TA atoms).
template <bool GUPD, int RT>
__global__ __launch_bounds__(256, RT == 1 ? EDGE_OCC32 : 2) void edge_kernel(EdgeArgs a) {
  constexpr int TEK = 32 * RT;     // edge rows per tile
  constexpr int TPR = 256 / TEK;   // threads per edge row in the row pass (4 or 8)
  constexpr int NCH = 32 / TPR;    // float4 chunks per thread
  constexpr int SA_ROWS = TEK > TA ? TEK : TA;
  __shared__ __attribute__((aligned(16))) float sA[SA_ROWS * LDS_STRIDE];  // G, then ang = c[j]*geom', then query rows
  __shared__ __attribute__((aligned(16))) float sB[TEK * LDS_STRIDE];      // U = G W2, then K
  __shared__ __attribute__((aligned(16))) float sE[TEK * NHEAD];           // logits
  __shared__ __attribute__((aligned(16))) float sPar[4 * D];               // layer_norm_g gamma/beta, layer_norm gamma/beta
  __shared__ int sCol[TEK], sCtr[TEK], sOff[TA + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const EdgeTile tile = a.tiles[blockIdx.x];
  const int eb = tile.edge_begin;
  const int ne = tile.edge_end - eb;
  const int natom = tile.atom_end - tile.atom_begin;
  const int col = 32 * wave + (lane & 31);
  float* const sQ = sA;  // [<=TA][LDS_STRIDE] query rows of the tile's atoms (attention phase)

  STAMP(a.stamps, 0);
  float4 w[16];
  if (GUPD) load_w(a.p.W2p, wave, lane, w);  // in flight while the geometry tile is staged
  else load_w(a.p.Wkp, wave, lane, w);
  if (tid < TEK) {
    sCol[tid] = tid < ne ? a.edge_col[eb + tid] : 0;
    sCtr[tid] = tid < ne ? a.edge_row[eb + tid] : 0;
  } else if (tid - TEK <= natom) {
    sOff[tid - TEK] = a.edge_offset[tile.atom_begin + (tid - TEK)] - eb;  // tile-local CSR row pointers
  }
  if (tid < D) {
    if (GUPD) {
      sPar[tid] = a.p.lng_g[tid];
      sPar[D + tid] = a.p.lng_b[tid];
    }
  } else {
    sPar[2 * D + (tid - D)] = a.p.ln_g[tid - D];
    sPar[3 * D + (tid - D)] = a.p.ln_b[tid - D];
  }
  f32x16 acc[RT];
  if (GUPD) {
    for (int i = tid; i < TEK * 32; i += 256) {
      const int r = i >> 5, c4 = i & 31;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < ne) v = reinterpret_cast<const float4*>(a.geom)[(size_t)(eb + r) * 32 + c4];
      *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * c4]) = v;
    }
    __syncthreads();
    STAMP(a.stamps, 1);
    // U = G . W2  (geometry third of the concat GEMM, attention.py:142-151)
    zero_acc(acc);
    mma128<RT>(sA, w, lane, acc);
    STAMP(a.stamps, 2);
    load_w(a.p.Wkp, wave, lane, w);  // key weights arrive during the row pass
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 16; ++i) sB[(32 * rt + acc_row(i, lane)) * LDS_STRIDE + col] = acc[rt][i];
  }
  __syncthreads();
  STAMP(a.stamps, 3);

  // Row pass, TPR threads per edge row: geometry update + LayerNorm_g, gate with the gathered neighbour row.
  {
    const int r = tid / TPR, sub = tid % TPR;
    if (r < ne) {
      const int ctr = sCtr[r], nb = sCol[r];
      const float4* crow = reinterpret_cast<const float4*>(a.c) + (size_t)nb * 32;
      if (GUPD) {
        const float4* p1 = reinterpret_cast<const float4*>(a.P1) + (size_t)ctr * 32;
        const float4* p3 = reinterpret_cast<const float4*>(a.P3) + (size_t)nb * 32;
        float4 t[NCH], cn[NCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) cn[i] = crow[sub + TPR * i];  // neighbour centre row (attention.py:136), used below
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c4 = sub + TPR * i;
          const float4 u = *reinterpret_cast<const float4*>(&sB[r * LDS_STRIDE + 4 * c4]);
          const float4 g = *reinterpret_cast<const float4*>(&sA[r * LDS_STRIDE + 4 * c4]);
          // concat order [centre, geometry, neighbour] (attention.py:143-149): (c_i W1 + b) + g W2 + c_j W3
          const float4 v = f4add(f4add(p1[c4], u), p3[c4]);
          t[i] = f4add(f4swish(v), g);  // geometry_update + neighbor_geometry (:153)
          s += f4sum(t[i]);
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / D);
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const float dx = t[i].x - mean, dy = t[i].y - mean, dz = t[i].z - mean, dw = t[i].w - mean;
          v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o);
        const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c4 = sub + TPR * i;
          const float4 g = *reinterpret_cast<const float4*>(&sPar[4 * c4]);
          const float4 be = *reinterpret_cast<const float4*>(&sPar[D + 4 * c4]);
          float4 y;
          float inv;
          inv = rstd * g.x; y.x = t[i].x * inv + (be.x - mean * inv);
          inv = rstd * g.y; y.y = t[i].y * inv + (be.y - mean * inv);
          inv = rstd * g.z; y.z = t[i].z * inv + (be.z - mean * inv);
          inv = rstd * g.w; y.w = t[i].w * inv + (be.w - mean * inv);
          reinterpret_cast<float4*>(a.geom)[(size_t)(eb + r) * 32 + c4] = y;  // threaded to the next layer (scann_model.py:415)
          *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * c4]) = f4mul(cn[i], y);  // attention.py:157
        }
      } else {
        // base SCANN: geomL = swish(gd Wf + bf) * weight (attention.py:155), gd = raw Gaussian basis
        float gd[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) gd[k] = a.gd[(size_t)(eb + r) * NG + k];
        const float wgt = a.edge_weight[eb + r];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c4 = sub + TPR * i;
          float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int k = 0; k < NG; ++k) {
            const float4 wv = reinterpret_cast<const float4*>(a.p.Wfg)[k * 32 + c4];
            acc4.x += gd[k] * wv.x; acc4.y += gd[k] * wv.y; acc4.z += gd[k] * wv.z; acc4.w += gd[k] * wv.w;
          }
          const float4 bv = reinterpret_cast<const float4*>(a.p.bfg)[c4];
          float4 y = f4swish(f4add(acc4, bv));
          y.x *= wgt; y.y *= wgt; y.z *= wgt; y.w *= wgt;
          *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * c4]) = f4mul(crow[c4], y);
        }
      }
    } else if (!GUPD && r < TEK) {
      // ragged tail rows must be defined for the MFMA (GUPD staged zeros already)
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * (sub + TPR * i)]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // query rows of this tile's atoms: fetched now, parked in registers across the key GEMM
  float4 qreg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, la = idx >> 5, c4 = idx & 31;
    qreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (la < natom) qreg[i] = reinterpret_cast<const float4*>(a.q)[(size_t)(tile.atom_begin + la) * 32 + c4];
  }
  __syncthreads();
  STAMP(a.stamps, 4);

  // K = ang . Wk + bk  (attention.py:163)
  zero_acc(acc);
  mma128<RT>(sA, w, lane, acc);
  STAMP(a.stamps, 5);
  {
    const float b = a.p.bk[col];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 16; ++i) sB[(32 * rt + acc_row(i, lane)) * LDS_STRIDE + col] = acc[rt][i] + b;
  }
  __syncthreads();  // K complete, sA no longer read by any MFMA
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, la = idx >> 5, c4 = idx & 31;
    *reinterpret_cast<float4*>(&sQ[la * LDS_STRIDE + 4 * c4]) = qreg[i];
  }
  __syncthreads();
  STAMP(a.stamps, 6);

  // Attention.  Packed edges are all unmasked, so the additive -1e9 mask and the multiplicative mask
  // (attention.py:186,206) are the identity; an atom without edges reduces to ctx = LN(q), exactly what the
  // reference's fully-masked row yields (uniform softmax zeroed by the mask).
  // (1) logits: thread = (edge row, 8/TPR heads); e[b,h,c,n] = sum_d (q*dk)[c,h,d] k[c,n,h,d]  (:180-183)
  {
    constexpr int HPT = NHEAD / TPR;  // heads per thread
    const int n = tid / TPR, hh = tid % TPR;
    if (n < ne) {
      const float* qrow = sQ + (sCtr[n] - tile.atom_begin) * LDS_STRIDE + HDIM * HPT * hh;
      const float* krow = sB + n * LDS_STRIDE + HDIM * HPT * hh;
#pragma unroll
      for (int hp = 0; hp < HPT; ++hp) {
        float e = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 q4 = *reinterpret_cast<const float4*>(qrow + HDIM * hp + 4 * j);
          const float4 k4 = *reinterpret_cast<const float4*>(krow + HDIM * hp + 4 * j);
          e += (q4.x * 0.25f) * k4.x; e += (q4.y * 0.25f) * k4.y; e += (q4.z * 0.25f) * k4.z; e += (q4.w * 0.25f) * k4.w;
        }
        sE[n * NHEAD + HPT * hh + hp] = e;
      }
    }
  }
  __syncthreads();
  STAMP(a.stamps, 8);
  // (2) softmax (tf.nn.softmax, :189) + context + residual: thread = (atom group, float4 chunk); the 4 lanes of
  // a head recompute that head's softmax rather than exchanging it.  exp via v_exp_f32, 1/sum via v_rcp_f32.
  {
    const int lgp = tid >> 5, c4 = tid & 31, h = c4 >> 2;
    for (int la = lgp; la < natom; la += 8) {
      const int e0 = sOff[la], e1 = sOff[la + 1];
      float m = -INFINITY;
      for (int n = e0; n < e1; ++n) m = fmaxf(m, sE[n * NHEAD + h]);
      float ssum = 0.f;
      for (int n = e0; n < e1; ++n) ssum += fast_exp(sE[n * NHEAD + h] - m);
      const float rs = __builtin_amdgcn_rcpf(ssum);
      float4 cx = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int n = e0; n < e1; ++n) {
        const float attn = fast_exp(sE[n * NHEAD + h] - m) * rs;
        const float4 k4 = *reinterpret_cast<const float4*>(&sB[n * LDS_STRIDE + 4 * c4]);
        cx.x += attn * k4.x; cx.y += attn * k4.y; cx.z += attn * k4.z; cx.w += attn * k4.w;  // v = key (:198-206)
      }
      float4* qp = reinterpret_cast<float4*>(&sQ[la * LDS_STRIDE + 4 * c4]);
      *qp = f4add(cx, *qp);  // residual is the unscaled query (:212)
    }
  }
  __syncthreads();
  STAMP(a.stamps, 9);
  // (3) LayerNorm of the context rows (:214): 8 threads per atom row
  {
    const int r = tid >> 3, sub = tid & 7;
    if (r < natom) {
      float4 t[4];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t[i] = *reinterpret_cast<const float4*>(&sQ[r * LDS_STRIDE + 4 * (sub + 8 * i)]);
        s += f4sum(t[i]);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      const float mean = s * (1.0f / D);
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dx = t[i].x - mean, dy = t[i].y - mean, dz = t[i].z - mean, dw = t[i].w - mean;
        v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c4 = sub + 8 * i;
        const float4 g = *reinterpret_cast<const float4*>(&sPar[2 * D + 4 * c4]);
        const float4 be = *reinterpret_cast<const float4*>(&sPar[3 * D + 4 * c4]);
        float4 y;
        float inv;
        inv = rstd * g.x; y.x = t[i].x * inv + (be.x - mean * inv);
        inv = rstd * g.y; y.y = t[i].y * inv + (be.y - mean * inv);
        inv = rstd * g.z; y.z = t[i].z * inv + (be.z - mean * inv);
        inv = rstd * g.w; y.w = t[i].w * inv + (be.w - mean * inv);
        reinterpret_cast<float4*>(a.ctx)[(size_t)(tile.atom_begin + r) * 32 + c4] = y;
      }
    }
  }
  STAMP(a.stamps, 7);
}

// ---- 8-wave edge kernel: same 64-row tile, one 32x32 MFMA tile per wave ----------------------------------------
//
// Occupancy variant of edge_kernel<true, 2>: 512 threads, wave (rt, cb) = (wave >> 2, wave & 3) owns rows [32 rt, 32 rt+32)
// x columns [32 cb, 32 cb+32).  Every VALU/LDS phase has twice the threads per tile (8 threads per edge row, one head per
// thread in the logits, 16 atom groups), and the weight slab is streamed in two halves so the kernel fits 128 VGPRs:
// 4 waves per SIMD, two workgroups per CU.
__device__ __forceinline__ void load_w_half(const float* __restrict__ Wp, int cb, int lane, int half, float4 (&w)[8]) {
  const float4* __restrict__ wsrc = reinterpret_cast<const float4*>(Wp) + cb * (16 * 64) + half * (8 * 64) + lane;
#pragma unroll
  for (int t = 0; t < 8; ++t) w[t] = wsrc[t * 64];
}
__device__ __forceinline__ void mma_half(const float* __restrict__ sXrt, const float4 (&w)[8], int lane, int half, f32x16& acc) {
  const float* xrow = sXrt + (lane & 31) * LDS_STRIDE + 4 * (lane >> 5) + 64 * half;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(xrow + 8 * t);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w[t].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w[t].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w[t].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w[t].w, acc, 0, 0, 0);
  }
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Consecutive tiles hold neighbouring
// atoms of the same structures and gather the same centre rows, so give every XCD a CONTIGUOUS run of tiles
// (bijective for any grid size).  Speed only: any placement is correct.
__device__ __forceinline__ int xcd_tile(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
  return x * q + min(x, r) + i;
}

// RTW = 32-row MFMA row tiles per workgroup: 2 -> 64-edge tiles, 512 threads (default); 1 -> 32-edge tiles, 256 threads.
template <int RTW>
__global__ __launch_bounds__(256 * RTW, 4) void edge_kernel_w8(EdgeArgs a) {
  constexpr int TEK = 32 * RTW;
  constexpr int NT = 256 * RTW;  // threads
  __shared__ __attribute__((aligned(16))) float sA[(TEK > TA ? TEK : TA) * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float sB[TEK * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float sE[TEK * NHEAD];
  __shared__ __attribute__((aligned(16))) float sPar[4 * D];
  __shared__ int sCol[TEK], sCtr[TEK], sOff[TA + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rt = wave >> 2, cb = wave & 3;
  const EdgeTile tile = a.tiles[a.xcd_remap ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x];
  const int eb = tile.edge_begin, ne = tile.edge_end - eb, natom = tile.atom_end - tile.atom_begin;
  const int col = 32 * cb + (lane & 31);
  float* const sQ = sA;

  STAMP(a.stamps, 0);
  float4 w[8];
  load_w_half(a.p.W2p, cb, lane, 0, w);
  if (tid < TEK) {
    sCol[tid] = tid < ne ? a.edge_col[eb + tid] : 0;
    sCtr[tid] = tid < ne ? a.edge_row[eb + tid] : 0;
  } else if (tid - TEK <= natom) {
    sOff[tid - TEK] = a.edge_offset[tile.atom_begin + (tid - TEK)] - eb;
  }
  for (int i = tid; i < 4 * D; i += NT) {
    const float* src = i < D ? a.p.lng_g : i < 2 * D ? a.p.lng_b : i < 3 * D ? a.p.ln_g : a.p.ln_b;
    sPar[i] = src[i & (D - 1)];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + NT * i, r = idx >> 5, c4 = idx & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < ne) v = reinterpret_cast<const float4*>(a.geom)[(size_t)(eb + r) * 32 + c4];
    *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * c4]) = v;
  }
  __syncthreads();
  STAMP(a.stamps, 1);
  // U = G . W2
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  mma_half(sA + rt * 32 * LDS_STRIDE, w, lane, 0, acc);
  load_w_half(a.p.W2p, cb, lane, 1, w);
  mma_half(sA + rt * 32 * LDS_STRIDE, w, lane, 1, acc);
  STAMP(a.stamps, 2);
  load_w_half(a.p.Wkp, cb, lane, 0, w);  // first half of the key weights arrives during the row pass
#pragma unroll
  for (int i = 0; i < 16; ++i) sB[(32 * rt + acc_row(i, lane)) * LDS_STRIDE + col] = acc[i];
  __syncthreads();
  STAMP(a.stamps, 3);

  // row pass, 8 threads per edge row (attention.py:141-157)
  {
    const int r = tid >> 3, sub = tid & 7;
    if (r < ne) {
      const int ctr = sCtr[r], nb = sCol[r];
      const float4* crow = reinterpret_cast<const float4*>(a.c) + (size_t)nb * 32;
      const float4* p1 = reinterpret_cast<const float4*>(a.P1) + (size_t)ctr * 32;
      const float4* p3 = reinterpret_cast<const float4*>(a.P3) + (size_t)nb * 32;
      float4 t[4];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c4 = sub + 8 * i;
        const float4 u = *reinterpret_cast<const float4*>(&sB[r * LDS_STRIDE + 4 * c4]);
        const float4 g = *reinterpret_cast<const float4*>(&sA[r * LDS_STRIDE + 4 * c4]);
        const float4 v = f4add(f4add(p1[c4], u), p3[c4]);
        t[i] = f4add(f4swish(v), g);
        s += f4sum(t[i]);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      const float mean = s * (1.0f / D);
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dx = t[i].x - mean, dy = t[i].y - mean, dz = t[i].z - mean, dw = t[i].w - mean;
        v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c4 = sub + 8 * i;
        const float4 g = *reinterpret_cast<const float4*>(&sPar[4 * c4]);
        const float4 be = *reinterpret_cast<const float4*>(&sPar[D + 4 * c4]);
        float4 y;
        float inv;
        inv = rstd * g.x; y.x = t[i].x * inv + (be.x - mean * inv);
        inv = rstd * g.y; y.y = t[i].y * inv + (be.y - mean * inv);
        inv = rstd * g.z; y.z = t[i].z * inv + (be.z - mean * inv);
        inv = rstd * g.w; y.w = t[i].w * inv + (be.w - mean * inv);
        reinterpret_cast<float4*>(a.geom)[(size_t)(eb + r) * 32 + c4] = y;
        *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * c4]) = f4mul(crow[c4], y);
      }
    }
  }
  constexpr int NQ = 1024 / NT;
  float4 qreg[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int idx = tid + NT * i, la = idx >> 5, c4 = idx & 31;
    qreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (la < natom) qreg[i] = reinterpret_cast<const float4*>(a.q)[(size_t)(tile.atom_begin + la) * 32 + c4];
  }
  __syncthreads();
  STAMP(a.stamps, 4);
  // K = ang . Wk + bk
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  mma_half(sA + rt * 32 * LDS_STRIDE, w, lane, 0, acc);
  load_w_half(a.p.Wkp, cb, lane, 1, w);
  mma_half(sA + rt * 32 * LDS_STRIDE, w, lane, 1, acc);
  STAMP(a.stamps, 5);
  {
    const float b = a.p.bk[col];
#pragma unroll
    for (int i = 0; i < 16; ++i) sB[(32 * rt + acc_row(i, lane)) * LDS_STRIDE + col] = acc[i] + b;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int idx = tid + NT * i, la = idx >> 5, c4 = idx & 31;
    *reinterpret_cast<float4*>(&sQ[la * LDS_STRIDE + 4 * c4]) = qreg[i];
  }
  __syncthreads();
  STAMP(a.stamps, 6);
  // logits: thread = (edge row, head)
  {
    const int n = tid >> 3, hh = tid & 7;
    if (n < ne) {
      const float* qrow = sQ + (sCtr[n] - tile.atom_begin) * LDS_STRIDE + HDIM * hh;
      const float* krow = sB + n * LDS_STRIDE + HDIM * hh;
      float e = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 q4 = *reinterpret_cast<const float4*>(qrow + 4 * j);
        const float4 k4 = *reinterpret_cast<const float4*>(krow + 4 * j);
        e += (q4.x * 0.25f) * k4.x; e += (q4.y * 0.25f) * k4.y; e += (q4.z * 0.25f) * k4.z; e += (q4.w * 0.25f) * k4.w;
      }
      sE[n * NHEAD + hh] = e;
    }
  }
  __syncthreads();
  STAMP(a.stamps, 8);
  // softmax + context + residual: thread = (atom group of 16, float4 chunk).  One pass over the atom's edges with a
  // running maximum (online softmax: exp(e - max) / sum, the max-subtracted form of tf.nn.softmax, attention.py:189,
  // evaluated with rescaling instead of three passes); two edges per iteration so their LDS reads overlap.
  {
    const int lgp = tid >> 5, c4 = tid & 31, h = c4 >> 2;
    for (int la = lgp; la < natom; la += NT / 32) {
      const int e0 = sOff[la], e1 = sOff[la + 1];
      float m = -INFINITY, ssum = 0.f;
      float4 cx = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int n = e0; n < e1; n += 2) {
        const bool two = n + 1 < e1;
        const int n1 = two ? n + 1 : n;
        const float ea = sE[n * NHEAD + h];
        const float eb2 = two ? sE[n1 * NHEAD + h] : -INFINITY;
        const float4 ka = *reinterpret_cast<const float4*>(&sB[n * LDS_STRIDE + 4 * c4]);
        const float4 kb = *reinterpret_cast<const float4*>(&sB[n1 * LDS_STRIDE + 4 * c4]);
        const float mn = fmaxf(m, fmaxf(ea, eb2));
        const float resc = fast_exp(m - mn);  // exp2(-inf) = 0 on the first iteration
        float pa = fast_exp(ea - mn), pb = fast_exp(eb2 - mn);
        ssum = ssum * resc + (pa + pb);
        if (a.attn_drop_p > 0.f) {  // training only: dropout on the (normalised) attention weights, not on the sum
          pa *= drop_scale(a.attn_drop_seed, a.attn_drop_tag, (size_t)(eb + n) * NHEAD + h, a.attn_drop_p);
          pb *= drop_scale(a.attn_drop_seed, a.attn_drop_tag, (size_t)(eb + n1) * NHEAD + h, a.attn_drop_p);
        }
        cx.x = cx.x * resc + (pa * ka.x + pb * kb.x);
        cx.y = cx.y * resc + (pa * ka.y + pb * kb.y);
        cx.z = cx.z * resc + (pa * ka.z + pb * kb.z);
        cx.w = cx.w * resc + (pa * ka.w + pb * kb.w);
        m = mn;
      }
      const float rs = e1 > e0 ? __builtin_amdgcn_rcpf(ssum) : 0.f;
      float4* qp = reinterpret_cast<float4*>(&sQ[la * LDS_STRIDE + 4 * c4]);
      const float4 q4 = *qp;
      *qp = make_float4(cx.x * rs + q4.x, cx.y * rs + q4.y, cx.z * rs + q4.z, cx.w * rs + q4.w);  // + unscaled query (:212)
    }
  }
  __syncthreads();
  STAMP(a.stamps, 9);
  // LayerNorm of the context rows: 16 threads per atom row
  for (int r = tid >> 4; r < TA; r += NT / 16) {
    const int sub = tid & 15;
    if (r < natom) {
      float4 t[2];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        t[i] = *reinterpret_cast<const float4*>(&sQ[r * LDS_STRIDE + 4 * (sub + 16 * i)]);
        s += f4sum(t[i]);
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
      const float mean = s * (1.0f / D);
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float dx = t[i].x - mean, dy = t[i].y - mean, dz = t[i].z - mean, dw = t[i].w - mean;
        v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
      const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = sub + 16 * i;
        const float4 g = *reinterpret_cast<const float4*>(&sPar[2 * D + 4 * c4]);
        const float4 be = *reinterpret_cast<const float4*>(&sPar[3 * D + 4 * c4]);
        float4 y;
        float inv;
        inv = rstd * g.x; y.x = t[i].x * inv + (be.x - mean * inv);
        inv = rstd * g.y; y.y = t[i].y * inv + (be.y - mean * inv);
        inv = rstd * g.z; y.z = t[i].z * inv + (be.z - mean * inv);
        inv = rstd * g.w; y.w = t[i].w * inv + (be.w - mean * inv);
        reinterpret_cast<float4*>(a.ctx)[(size_t)(tile.atom_begin + r) * 32 + c4] = y;
      }
    }
  }
  STAMP(a.stamps, 7);
}

// ---- lean-LDS edge kernel: three workgroups per CU ---------------------------------------------------------------
//
// Phase clocks of edge_kernel_w8 show the two GEMM phases saturating the MFMA pipe (8.2 k cycles each per tile) and the
// other phases being latency chains (35 k cycles per tile), so MFMA utilisation is set by how many tiles a CU keeps in
// flight: 2 x 16.4 k / 51.6 k = 62 %.  This variant keeps ONE 64 x 128 LDS buffer per workgroup -- G, U, ang and K
// take turns in it; the thread that stages a piece of G keeps it in registers for the residual -- plus the query
// rows of <= TQ atoms: 51 KB, so three workgroups (of 4 waves, <= 168 VGPRs) fit a CU.
__device__ __forceinline__ void mma_half2(const float* __restrict__ sX, const float4 (&w)[8], int lane, int half,
                                          f32x16 (&acc)[2]) {
  const float* xrow = sX + (lane & 31) * LDS_STRIDE + 4 * (lane >> 5) + 64 * half;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const float4 a = *reinterpret_cast<const float4*>(xrow + rt * 32 * LDS_STRIDE + 8 * t);
      // operands swapped (weights as A, rows as B): the product comes out transposed, i.e. lane l holds ROW l & 31 and the
      // sixteen columns (i & 3) + 8 (i >> 2) + 4 (l >> 5) of the wave's 32 -- four runs of four consecutive columns, so the
      // tile is written back with ds_write_b128 instead of sixteen ds_write_b32
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].x, a.x, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].y, a.y, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].z, a.z, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].w, a.w, acc[rt], 0, 0, 0);
    }
  }
}
// write the transposed accumulators of mma_half2 (+ a per-column bias row, or null) to a [64][LDS_STRIDE] tile
__device__ __forceinline__ void dump_t2(float* __restrict__ sT, const f32x16 (&acc)[2], int wave, int lane, const float* __restrict__ sBias) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 32 * wave + 8 * j + 4 * (lane >> 5);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sBias) bv = *reinterpret_cast<const float4*>(sBias + c);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      *reinterpret_cast<float4*>(&sT[(32 * rt + (lane & 31)) * LDS_STRIDE + c]) =
          make_float4(acc[rt][4 * j] + bv.x, acc[rt][4 * j + 1] + bv.y, acc[rt][4 * j + 2] + bv.z, acc[rt][4 * j + 3] + bv.w);
  }
}

__global__ __launch_bounds__(256, 3) void edge_kernel_lean(EdgeArgs a) {
  constexpr int TEK = 64;
  __shared__ __attribute__((aligned(16))) float sA[TEK * LDS_STRIDE];  // G -> U -> ang = c[j]*geom' -> K
  __shared__ __attribute__((aligned(16))) float sQ[TQ * LDS_STRIDE];   // P1 rows, then query rows of the tile's atoms, then context
  __shared__ __attribute__((aligned(16))) float sE[TEK * NHEAD];
  __shared__ __attribute__((aligned(16))) float sPar[5 * D];  // layer_norm_g gamma/beta, layer_norm gamma/beta, key bias
  __shared__ int sCol[TEK], sCtr[TEK], sOff[TQ + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tix = a.xcd_remap ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
  const EdgeTile tile = a.tiles[tix];
  const int part = a.tile_part ? a.tile_part[tix] : -1;  // >= 0: one <= 64-edge chunk of an atom with more than 64 neighbours
  const int eb = tile.edge_begin, ne = tile.edge_end - eb, natom = tile.atom_end - tile.atom_begin;
  const int r = tid >> 2, sub = tid & 3;  // row-pass mapping: 4 threads per edge row, float4 chunks sub, sub+4, ...

  STAMP(a.stamps, 0);
  STAMP_REAL(a.stamps, 12);  // 100 MHz reference clock at entry and exit: shader clock = cycles / ticks * 100 MHz
  // Every load of the prologue is issued UNCONDITIONALLY (rows clamped into the tile, values selected afterwards): a load
  // under a per-thread guard is compiled as branch + load + s_waitcnt vmcnt(0) + store, i.e. one full memory round trip
  // per guard (16 in a row here before: 22 k cycles per tile).
  const int nem1 = ne > 0 ? ne - 1 : 0;
  const int rs = r < ne ? r : nem1;
  float4 wA[8], wB[8];
  load_w_half(a.p.W2p, wave, lane, 0, wA);
  load_w_half(a.p.W2p, wave, lane, 1, wB);
  const int32_t* pa = tid < TEK ? (ne > 0 ? a.edge_col + eb + min(tid, nem1) : a.edge_offset)
                                : a.edge_offset + tile.atom_begin + min(tid - TEK, natom);
  const int32_t* pb = ne > 0 ? a.edge_row + eb + min(tid & (TEK - 1), nem1) : a.edge_offset;
  const int va = *pa, vb = *pb;
  const float bkc = a.p.bk[tid & (D - 1)];  // key bias row (-> sPar): fetched with the prologue so that its wait never queues behind the geometry stores
  const float par0 = (tid < D ? a.p.lng_g : a.p.lng_b)[tid & (D - 1)];
  const float par1 = (tid < D ? a.p.ln_g : a.p.ln_b)[tid & (D - 1)];
  float4 p1reg[3];  // centre thirds P1 = c_i W1 + bg of the tile's atoms (an atom's edges share the row)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 256 * i, la = min(idx >> 5, natom - 1), c4 = idx & 31;
    p1reg[i] = reinterpret_cast<const float4*>(a.P1)[(size_t)(tile.atom_begin + la) * 32 + c4];
  }
  float4 greg[8];  // this thread's pieces of G stay in registers for the residual (attention.py:153)
  {
    const float4* grow = reinterpret_cast<const float4*>(ne > 0 ? a.geom + (size_t)(eb + rs) * D : a.P1);
#pragma unroll
    for (int i = 0; i < 8; ++i) greg[i] = grow[sub + 4 * i];
  }
  if (tid < TEK) {
    sCol[tid] = tid < ne ? va : 0;
    sCtr[tid] = tid < ne ? vb : 0;
  } else if (tid - TEK <= natom) {
    sOff[tid - TEK] = part >= 0 ? (tid == TEK ? 0 : ne) : va - eb;  // a chunk tile holds edges [0, ne) of its single atom
  }
  sPar[tid] = par0;
  sPar[2 * D + tid] = par1;
  if (tid < D) sPar[4 * D + tid] = bkc;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 256 * i;
    *reinterpret_cast<float4*>(&sQ[(idx >> 5) * LDS_STRIDE + 4 * (idx & 31)]) = p1reg[i];  // rows >= natom: unused copies
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (r >= ne) greg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * (sub + 4 * i)]) = greg[i];
  }
  __syncthreads();
  STAMP(a.stamps, 1);
  // U = G . W2
  f32x16 acc[2];
  zero_acc(acc);
  mma_half2(sA, wA, lane, 0, acc);
  mma_half2(sA, wB, lane, 1, acc);
  STAMP(a.stamps, 2);
  __syncthreads();  // every wave is done reading G
  dump_t2(sA, acc, wave, lane, nullptr);
  __syncthreads();
  STAMP(a.stamps, 3);

#ifdef SCANN_DIAG_GEMMONLY
  // diagnostic (wrong results): prologue + the two GEMMs + their write-backs only -- the rate the matrix work alone reaches
  load_w_half(a.p.Wkp, wave, lane, 0, wA);
  load_w_half(a.p.Wkp, wave, lane, 1, wB);
  __syncthreads();
  zero_acc(acc);
  mma_half2(sA, wA, lane, 0, acc);
  mma_half2(sA, wB, lane, 1, acc);
  __syncthreads();
  dump_t2(sA, acc, wave, lane, sPar + 4 * D);
  __syncthreads();
  if (r < ne) reinterpret_cast<float4*>(a.geom_out ? a.geom_out : a.geom)[(size_t)(eb + r) * 32 + sub] = *reinterpret_cast<const float4*>(&sA[r * LDS_STRIDE + 4 * sub]);
  return;
#endif
  // row pass (attention.py:141-157): u read and ang written at the thread's own positions of sA
  if (r < ne) {
    const int ctr = sCtr[r], nb = sCol[r];
    const float4* crow = reinterpret_cast<const float4*>(a.c) + (size_t)nb * 32;
    const float* p1 = sQ + (ctr - tile.atom_begin) * LDS_STRIDE;
    const float4* p3 = reinterpret_cast<const float4*>(a.P3) + (size_t)nb * 32;
    // all sixteen gathered pieces (P3 and centre row of the neighbour, attention.py:136) are requested at once: one memory
    // round trip for the row pass (the key weights are fetched after it, so the registers are free here)
    float4 p3r[8], cn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p3r[i] = p3[sub + 4 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) cn[i] = crow[sub + 4 * i];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = sub + 4 * i;
      const float4 u = *reinterpret_cast<const float4*>(&sA[r * LDS_STRIDE + 4 * c4]);
      const float4 v = f4add(f4add(*reinterpret_cast<const float4*>(p1 + 4 * c4), u), p3r[i]);
      greg[i] = f4add(f4swish(v), greg[i]);
      s += f4sum(greg[i]);
      if (a.keep_V) {  // training forward: the backward reads these instead of recomputing them
        reinterpret_cast<float4*>(a.keep_V)[(size_t)(eb + r) * 32 + c4] = v;
        reinterpret_cast<float4*>(a.keep_T)[(size_t)(eb + r) * 32 + c4] = greg[i];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    load_w_half(a.p.Wkp, wave, lane, 0, wA);  // the P3 registers are free: first half of the key weights lands over the statistics
    __builtin_amdgcn_sched_barrier(0);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    const float mean = s * (1.0f / D);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float dx = greg[i].x - mean, dy = greg[i].y - mean, dz = greg[i].z - mean, dw = greg[i].w - mean;
      v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = sub + 4 * i;
      const float4 g = *reinterpret_cast<const float4*>(&sPar[4 * c4]);
      const float4 be = *reinterpret_cast<const float4*>(&sPar[D + 4 * c4]);
      float4 y;
      float inv;
      inv = rstd * g.x; y.x = greg[i].x * inv + (be.x - mean * inv);
      inv = rstd * g.y; y.y = greg[i].y * inv + (be.y - mean * inv);
      inv = rstd * g.z; y.z = greg[i].z * inv + (be.z - mean * inv);
      inv = rstd * g.w; y.w = greg[i].w * inv + (be.w - mean * inv);
      reinterpret_cast<float4*>(a.geom_out ? a.geom_out : a.geom)[(size_t)(eb + r) * 32 + c4] = y;
      const float4 ang = f4mul(cn[i], y);
      *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * c4]) = ang;
      if (a.keep_ang) reinterpret_cast<float4*>(a.keep_ang)[(size_t)(eb + r) * 32 + c4] = ang;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)  // ragged tail rows: U of the zero rows is 0 already, keep ang = 0 explicit
      *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * (sub + 4 * i)]) = make_float4(0.f, 0.f, 0.f, 0.f);
    load_w_half(a.p.Wkp, wave, lane, 0, wA);
  }
  load_w_half(a.p.Wkp, wave, lane, 1, wB);
  // query rows: fetched here, parked in sQ as soon as the P1 rows are dead (rows clamped, never guarded: see the prologue)
  const float4* const q4p = reinterpret_cast<const float4*>(a.q) + (size_t)tile.atom_begin * 32 + (tid & 31);
  const int qa = tid >> 5;  // atom rows qa, qa + 8, qa + 16
  const float4 q0 = q4p[(size_t)min(qa, natom - 1) * 32];
  const float4 q1 = q4p[(size_t)min(qa + 8, natom - 1) * 32];
  const float4 q2 = q4p[(size_t)min(qa + 16, natom - 1) * 32];
  __syncthreads();  // ang complete; nobody reads the P1 rows any more
  *reinterpret_cast<float4*>(&sQ[qa * LDS_STRIDE + 4 * (tid & 31)]) = q0;
  *reinterpret_cast<float4*>(&sQ[(qa + 8) * LDS_STRIDE + 4 * (tid & 31)]) = q1;
  *reinterpret_cast<float4*>(&sQ[(qa + 16) * LDS_STRIDE + 4 * (tid & 31)]) = q2;
  __builtin_amdgcn_sched_barrier(0);  // keep the stores here: sunk below the GEMM the rows get parked in scratch
  STAMP(a.stamps, 4);
  // K = ang . Wk + bk
  zero_acc(acc);
  mma_half2(sA, wA, lane, 0, acc);
  mma_half2(sA, wB, lane, 1, acc);
  STAMP(a.stamps, 5);
  __syncthreads();  // every wave is done reading ang
  dump_t2(sA, acc, wave, lane, sPar + 4 * D);
  __syncthreads();
  if (a.keep_K && r < ne) {  // training forward: K rows of the tile (coalesced copy of the finished LDS tile)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      reinterpret_cast<float4*>(a.keep_K)[(size_t)(eb + r) * 32 + sub + 4 * i] = *reinterpret_cast<const float4*>(&sA[r * LDS_STRIDE + 4 * (sub + 4 * i)]);
  }
  STAMP(a.stamps, 6);
  // logits: thread = (edge row, pair of heads)
  {
    const int n = tid >> 2, hh = tid & 3;
    if (n < ne) {
      const float* qrow = sQ + (sCtr[n] - tile.atom_begin) * LDS_STRIDE + 2 * HDIM * hh;
      const float* krow = sA + n * LDS_STRIDE + 2 * HDIM * hh;
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        float e = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 q4 = *reinterpret_cast<const float4*>(qrow + HDIM * hp + 4 * j);
          const float4 k4 = *reinterpret_cast<const float4*>(krow + HDIM * hp + 4 * j);
          e += (q4.x * 0.25f) * k4.x; e += (q4.y * 0.25f) * k4.y; e += (q4.z * 0.25f) * k4.z; e += (q4.w * 0.25f) * k4.w;
        }
        sE[n * NHEAD + 2 * hh + hp] = e;
      }
    }
  }
  __syncthreads();
  STAMP(a.stamps, 8);
  // softmax + context + residual, one pass with a running maximum (see edge_kernel_w8)
  {
    const int lgp = tid >> 5, c4 = tid & 31, h = c4 >> 2;
    for (int la = lgp; la < natom; la += 8) {
      const int e0 = sOff[la], e1 = sOff[la + 1];
      float m = -INFINITY, ssum = 0.f;
      float4 cx = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int n = e0; n < e1; n += 2) {
        const bool two = n + 1 < e1;
        const int n1 = two ? n + 1 : n;
        const float ea = sE[n * NHEAD + h];
        const float eb2 = two ? sE[n1 * NHEAD + h] : -INFINITY;
        const float4 ka = *reinterpret_cast<const float4*>(&sA[n * LDS_STRIDE + 4 * c4]);
        const float4 kb = *reinterpret_cast<const float4*>(&sA[n1 * LDS_STRIDE + 4 * c4]);
        const float mn = fmaxf(m, fmaxf(ea, eb2));
        const float resc = fast_exp(m - mn);
        float pa = fast_exp(ea - mn), pb = fast_exp(eb2 - mn);
        ssum = ssum * resc + (pa + pb);
        if (a.attn_drop_p > 0.f) {
          pa *= drop_scale(a.attn_drop_seed, a.attn_drop_tag, (size_t)(eb + n) * NHEAD + h, a.attn_drop_p);
          pb *= drop_scale(a.attn_drop_seed, a.attn_drop_tag, (size_t)(eb + n1) * NHEAD + h, a.attn_drop_p);
        }
        cx.x = cx.x * resc + (pa * ka.x + pb * kb.x);
        cx.y = cx.y * resc + (pa * ka.y + pb * kb.y);
        cx.z = cx.z * resc + (pa * ka.z + pb * kb.z);
        cx.w = cx.w * resc + (pa * ka.w + pb * kb.w);
        m = mn;
      }
      if (part >= 0) {  // chunk tile: leave the softmax state of this chunk for edge_merge_kernel
        float* pb = a.part_buf + (size_t)part * 3 * D + 4 * c4;
        *reinterpret_cast<float4*>(pb) = make_float4(m, m, m, m);
        *reinterpret_cast<float4*>(pb + D) = make_float4(ssum, ssum, ssum, ssum);
        *reinterpret_cast<float4*>(pb + 2 * D) = cx;
      } else {
        const float rs = e1 > e0 ? __builtin_amdgcn_rcpf(ssum) : 0.f;
        float4* qp = reinterpret_cast<float4*>(&sQ[la * LDS_STRIDE + 4 * c4]);
        const float4 q4 = *qp;
        *qp = make_float4(cx.x * rs + q4.x, cx.y * rs + q4.y, cx.z * rs + q4.z, cx.w * rs + q4.w);
      }
    }
  }
  __syncthreads();
  STAMP(a.stamps, 9);
  // LayerNorm of the context rows: 8 threads per atom row
  {
    const int rr = tid >> 3, sb = tid & 7;
    if (rr < natom && part < 0) {
      float4 t[4];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t[i] = *reinterpret_cast<const float4*>(&sQ[rr * LDS_STRIDE + 4 * (sb + 8 * i)]);
        s += f4sum(t[i]);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      const float mean = s * (1.0f / D);
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dx = t[i].x - mean, dy = t[i].y - mean, dz = t[i].z - mean, dw = t[i].w - mean;
        v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c4 = sb + 8 * i;
        const float4 g = *reinterpret_cast<const float4*>(&sPar[2 * D + 4 * c4]);
        const float4 be = *reinterpret_cast<const float4*>(&sPar[3 * D + 4 * c4]);
        float4 y;
        float inv;
        inv = rstd * g.x; y.x = t[i].x * inv + (be.x - mean * inv);
        inv = rstd * g.y; y.y = t[i].y * inv + (be.y - mean * inv);
        inv = rstd * g.z; y.z = t[i].z * inv + (be.z - mean * inv);
        inv = rstd * g.w; y.w = t[i].w * inv + (be.w - mean * inv);
        reinterpret_cast<float4*>(a.ctx)[(size_t)(tile.atom_begin + rr) * 32 + c4] = y;
      }
    }
  }
  STAMP(a.stamps, 7);
  STAMP_REAL(a.stamps, 13);
}

// Atoms with more than 64 neighbours: combine the per-chunk softmax states (running max m, sum s, unnormalised context x per
// column) exactly as the online softmax combines edges -- M = max m_i, s = sum s_i e^(m_i - M), x = sum x_i e^(m_i - M) -- then
// add the unscaled query and apply the LayerNorm (attention.py:189-214).  One workgroup of 128 threads per such atom.
__global__ __launch_bounds__(128) void edge_merge_kernel(const int32_t* __restrict__ big_tab, const float* __restrict__ part_buf,
                                                         const float* __restrict__ q, const float* __restrict__ ln_g,
                                                         const float* __restrict__ ln_b, float* __restrict__ ctx) {
  __shared__ float sRed[2][2];
  const int c = threadIdx.x, atom = big_tab[3 * blockIdx.x], s0 = big_tab[3 * blockIdx.x + 1], ns = big_tab[3 * blockIdx.x + 2];
  const float* pb = part_buf + (size_t)s0 * 3 * D + c;
  float M = -INFINITY;
  for (int i = 0; i < ns; ++i) M = fmaxf(M, pb[(size_t)i * 3 * D]);
  float S = 0.f, X = 0.f;
  for (int i = 0; i < ns; ++i) {
    const float wgt = fast_exp(pb[(size_t)i * 3 * D] - M);
    S += pb[(size_t)i * 3 * D + D] * wgt;
    X += pb[(size_t)i * 3 * D + 2 * D] * wgt;
  }
  const float t = X * __builtin_amdgcn_rcpf(S) + q[(size_t)atom * D + c];
  float s = t;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
  if ((c & 63) == 0) sRed[0][c >> 6] = s;
  __syncthreads();
  const float mean = (sRed[0][0] + sRed[0][1]) * (1.0f / D);
  const float d = t - mean;
  float v = d * d;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  if ((c & 63) == 0) sRed[1][c >> 6] = v;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((sRed[1][0] + sRed[1][1]) * (1.0f / D) + 1e-6f);
  const float inv = rstd * ln_g[c];
  ctx[(size_t)atom * D + c] = t * inv + (ln_b[c] - mean * inv);
}

void launch_edge_merge(const int32_t* big_tab, int n_big, const float* part_buf, const float* q, const float* ln_g,
                       const float* ln_b, float* ctx, hipStream_t s) {
  if (n_big > 0) hipLaunchKernelGGL(edge_merge_kernel, dim3(n_big), dim3(128), 0, s, big_tab, part_buf, q, ln_g, ln_b, ctx);
}

// ---- persistent form of edge_kernel_lean: the next tile's inputs are requested before the attention phases ------------
//
// Occupancy bound of the tile structure: N T_mfma / (T_other + T_mfma) with N = 3 tiles per CU; the prologue (13.5 k of the
// 52 k cycles of T_other) is pure waiting for the tile's geometry rows, P1 rows, indices and first weight slab.  Here a
// workgroup walks tiles blockIdx.x, + gridDim.x, ... and issues tile k+1's prologue loads into registers right after tile
// k's logits, so they land under the softmax / LayerNorm phases; LayerNorm parameters and the key bias are staged once.
// One tile's inputs, requested into plain locals of the kernel (a struct passed by reference stays a stack object here:
// every fetched value went through scratch).  VT = virtual block index of the tile.
#define LEAN_FETCH(VT)                                                                                                  \
  do {                                                                                                                  \
    const int tix_ = xr ? xcd_tile((VT), n_tile) : (VT);                                                                \
    const EdgeTile tile_ = g_tiles[tix_];                                                                               \
    t_part = g_part ? g_part[tix_] : -1;                                                                                \
    t_eb = tile_.edge_begin; t_ne = tile_.edge_end - tile_.edge_begin; t_natom = tile_.atom_end - tile_.atom_begin;      \
    t_abeg = tile_.atom_begin;                                                                                          \
    const int nem1_ = t_ne > 0 ? t_ne - 1 : 0;                                                                          \
    const int rs_ = r < t_ne ? r : nem1_;                                                                               \
    const int32_t* pa_ = tid < 64 ? (t_ne > 0 ? g_col + t_eb + min(tid, nem1_) : g_eoff)                                \
                                  : g_eoff + t_abeg + min(tid - 64, t_natom);                                           \
    const int32_t* pb_ = t_ne > 0 ? g_row + t_eb + min(tid & 63, nem1_) : g_eoff;                                       \
    t_va = *pa_; t_vb = *pb_;                                                                                           \
    {                                                                                                                   \
      const float4* p14_ = reinterpret_cast<const float4*>(g_P1) + (size_t)t_abeg * 32 + (tid & 31);                    \
      const int qa_ = tid >> 5;                                                                                         \
      t_p1a = p14_[(size_t)min(qa_, t_natom - 1) * 32];                                                                 \
      t_p1b = p14_[(size_t)min(qa_ + 8, t_natom - 1) * 32];                                                             \
      t_p1c = p14_[(size_t)min(qa_ + 16, t_natom - 1) * 32];                                                            \
    }                                                                                                                   \
    const float4* grow_ = reinterpret_cast<const float4*>(t_ne > 0 ? g_geom + (size_t)(t_eb + rs_) * D : g_P1);          \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) t_g[i_] = grow_[sub + 4 * i_];                                     \
  } while (0)

__global__ __launch_bounds__(256, 3) void edge_kernel_leanp(EdgeArgs a) {
  constexpr int TEK = 64;
  __shared__ __attribute__((aligned(16))) float sA[TEK * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float sQ[TQ * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float sE[TEK * NHEAD];
  __shared__ __attribute__((aligned(16))) float sPar[5 * D];
  __shared__ int sCol[TEK], sCtr[TEK], sOff[TQ + 1];
  const int tid0 = threadIdx.x;
  // kernel-argument fields used inside the loop, hoisted (the argument struct itself is never address-taken)
  const EdgeTile* const g_tiles = a.tiles;
  const int32_t* const g_part = a.tile_part;
  const int32_t* const g_col = a.edge_col;
  const int32_t* const g_row = a.edge_row;
  const int32_t* const g_eoff = a.edge_offset;
  float* const g_geom = a.geom;
  float* const g_geom_out = a.geom_out ? a.geom_out : a.geom;
  const float* const g_c = a.c;
  const float* const g_P1 = a.P1;
  const float* const g_P3 = a.P3;
  const float* const g_q = a.q;
  float* const g_ctx = a.ctx;
  float* const g_pbuf = a.part_buf;
  const float* const g_W2p = a.p.W2p;
  const float* const g_Wkp = a.p.Wkp;
  const int n_tile = a.n_tile, xr = a.xcd_remap, nwg = gridDim.x;
  const float drop_p = a.attn_drop_p;
  const unsigned drop_tag = a.attn_drop_tag;
  const unsigned long long drop_seed = a.attn_drop_seed;

  sPar[tid0] = (tid0 < D ? a.p.lng_g : a.p.lng_b)[tid0 & (D - 1)];
  sPar[2 * D + tid0] = (tid0 < D ? a.p.ln_g : a.p.ln_b)[tid0 & (D - 1)];
  if (tid0 < D) sPar[4 * D + tid0] = a.p.bk[tid0];

  float4 t_g[8], t_p1a, t_p1b, t_p1c;
  int t_va, t_vb, t_eb, t_ne, t_natom, t_abeg, t_part;
  {
    const int tid = tid0, r = tid >> 2, sub = tid & 3;
    LEAN_FETCH((int)blockIdx.x);
  }
  for (int vt = blockIdx.x;;) {
    // the thread index is re-materialised per iteration behind an opaque barrier: otherwise every per-thread address of the
    // body (LDS rows, weight fragments, staging slots) is hoisted out of the loop and kept in registers across it (156 spills)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6, r = tid >> 2, sub = tid & 3;
    const int eb = t_eb, ne = t_ne, natom = t_natom, abeg = t_abeg, part = t_part;
    float4 wA[8], wB[8];  // the weight slabs are L2-resident: requested here, they land while the tile is staged
    load_w_half(g_W2p, wave, lane, 0, wA);
    load_w_half(g_W2p, wave, lane, 1, wB);
    if (tid < TEK) {
      sCol[tid] = tid < ne ? t_va : 0;
      sCtr[tid] = tid < ne ? t_vb : 0;
    } else if (tid - TEK <= natom) {
      sOff[tid - TEK] = part >= 0 ? (tid == TEK ? 0 : ne) : t_va - eb;
    }
    *reinterpret_cast<float4*>(&sQ[(tid >> 5) * LDS_STRIDE + 4 * (tid & 31)]) = t_p1a;
    *reinterpret_cast<float4*>(&sQ[((tid >> 5) + 8) * LDS_STRIDE + 4 * (tid & 31)]) = t_p1b;
    *reinterpret_cast<float4*>(&sQ[((tid >> 5) + 16) * LDS_STRIDE + 4 * (tid & 31)]) = t_p1c;
    float4 greg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      greg[i] = r < ne ? t_g[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * (sub + 4 * i)]) = greg[i];
    }
    __syncthreads();
    // U = G . W2
    f32x16 acc[2];
    zero_acc(acc);
    mma_half2(sA, wA, lane, 0, acc);
    mma_half2(sA, wB, lane, 1, acc);
    __syncthreads();  // every wave is done reading G
    dump_t2(sA, acc, wave, lane, nullptr);
    __syncthreads();
    // row pass (attention.py:141-157)
    if (r < ne) {
      const int ctr = sCtr[r], nb = sCol[r];
      const float4* crow = reinterpret_cast<const float4*>(g_c) + (size_t)nb * 32;
      const float* p1 = sQ + (ctr - abeg) * LDS_STRIDE;
      const float4* p3 = reinterpret_cast<const float4*>(g_P3) + (size_t)nb * 32;
      float4 cn[8];
      float s = 0.f;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {  // P3 pieces in two groups of four (register budget of the persistent form)
        float4 p3r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p3r[i] = p3[sub + 4 * (4 * hf + i)];
        if (hf == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) cn[i] = crow[sub + 4 * i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c4 = sub + 4 * (4 * hf + i);
          const float4 u = *reinterpret_cast<const float4*>(&sA[r * LDS_STRIDE + 4 * c4]);
          const float4 v = f4add(f4add(*reinterpret_cast<const float4*>(p1 + 4 * c4), u), p3r[i]);
          greg[4 * hf + i] = f4add(f4swish(v), greg[4 * hf + i]);
          s += f4sum(greg[4 * hf + i]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      load_w_half(g_Wkp, wave, lane, 0, wA);
      __builtin_amdgcn_sched_barrier(0);
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      const float mean = s * (1.0f / D);
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float dx = greg[i].x - mean, dy = greg[i].y - mean, dz = greg[i].z - mean, dw = greg[i].w - mean;
        v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c4 = sub + 4 * i;
        const float4 g = *reinterpret_cast<const float4*>(&sPar[4 * c4]);
        const float4 be = *reinterpret_cast<const float4*>(&sPar[D + 4 * c4]);
        float4 y;
        float inv;
        inv = rstd * g.x; y.x = greg[i].x * inv + (be.x - mean * inv);
        inv = rstd * g.y; y.y = greg[i].y * inv + (be.y - mean * inv);
        inv = rstd * g.z; y.z = greg[i].z * inv + (be.z - mean * inv);
        inv = rstd * g.w; y.w = greg[i].w * inv + (be.w - mean * inv);
        reinterpret_cast<float4*>(g_geom_out)[(size_t)(eb + r) * 32 + c4] = y;
        *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * c4]) = f4mul(cn[i], y);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * (sub + 4 * i)]) = make_float4(0.f, 0.f, 0.f, 0.f);
      load_w_half(g_Wkp, wave, lane, 0, wA);
    }
    load_w_half(g_Wkp, wave, lane, 1, wB);
    const float4* const q4p = reinterpret_cast<const float4*>(g_q) + (size_t)abeg * 32 + (tid & 31);
    const int qa = tid >> 5;
    const float4 q0 = q4p[(size_t)min(qa, natom - 1) * 32];
    const float4 q1 = q4p[(size_t)min(qa + 8, natom - 1) * 32];
    const float4 q2 = q4p[(size_t)min(qa + 16, natom - 1) * 32];
    __syncthreads();  // ang complete; nobody reads the P1 rows any more
    *reinterpret_cast<float4*>(&sQ[qa * LDS_STRIDE + 4 * (tid & 31)]) = q0;
    *reinterpret_cast<float4*>(&sQ[(qa + 8) * LDS_STRIDE + 4 * (tid & 31)]) = q1;
    *reinterpret_cast<float4*>(&sQ[(qa + 16) * LDS_STRIDE + 4 * (tid & 31)]) = q2;
    __builtin_amdgcn_sched_barrier(0);
    // K = ang . Wk + bk
    zero_acc(acc);
    mma_half2(sA, wA, lane, 0, acc);
    mma_half2(sA, wB, lane, 1, acc);
    __syncthreads();  // every wave is done reading ang
    dump_t2(sA, acc, wave, lane, sPar + 4 * D);
    __syncthreads();
    // logits: thread = (edge row, pair of heads)
    {
      const int n = tid >> 2, hh = tid & 3;
      if (n < ne) {
        const float* qrow = sQ + (sCtr[n] - abeg) * LDS_STRIDE + 2 * HDIM * hh;
        const float* krow = sA + n * LDS_STRIDE + 2 * HDIM * hh;
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          float e = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 q4 = *reinterpret_cast<const float4*>(qrow + HDIM * hp + 4 * j);
            const float4 k4 = *reinterpret_cast<const float4*>(krow + HDIM * hp + 4 * j);
            e += (q4.x * 0.25f) * k4.x; e += (q4.y * 0.25f) * k4.y; e += (q4.z * 0.25f) * k4.z; e += (q4.w * 0.25f) * k4.w;
          }
          sE[n * NHEAD + 2 * hh + hp] = e;
        }
      }
    }
    // the next tile's inputs are requested here: they land under the softmax / LayerNorm phases
    const int vn = vt + nwg;
    const bool more = vn < n_tile;
    // (every field of `t` is dead here: the tile's scalars were copied at the top of the iteration)
    // unconditional (the last iteration re-requests its own tile): under `if (more)` the old values would stay live through
    // the whole iteration as the other input of the merge, i.e. 44 more registers under the GEMMs and the row pass
    LEAN_FETCH(more ? vn : vt);
    __syncthreads();
    // softmax + context + residual (online form)
    {
      const int lgp = tid >> 5, c4 = tid & 31, h = c4 >> 2;
      for (int la = lgp; la < natom; la += 8) {
        const int e0 = sOff[la], e1 = sOff[la + 1];
        float m = -INFINITY, ssum = 0.f;
        float4 cx = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = e0; n < e1; n += 2) {
          const bool two = n + 1 < e1;
          const int n1 = two ? n + 1 : n;
          const float ea = sE[n * NHEAD + h];
          const float eb2 = two ? sE[n1 * NHEAD + h] : -INFINITY;
          const float4 ka = *reinterpret_cast<const float4*>(&sA[n * LDS_STRIDE + 4 * c4]);
          const float4 kb = *reinterpret_cast<const float4*>(&sA[n1 * LDS_STRIDE + 4 * c4]);
          const float mn = fmaxf(m, fmaxf(ea, eb2));
          const float resc = fast_exp(m - mn);
          float pa2 = fast_exp(ea - mn), pb2 = fast_exp(eb2 - mn);
          ssum = ssum * resc + (pa2 + pb2);
          if (drop_p > 0.f) {
            pa2 *= drop_scale(drop_seed, drop_tag, (size_t)(eb + n) * NHEAD + h, drop_p);
            pb2 *= drop_scale(drop_seed, drop_tag, (size_t)(eb + n1) * NHEAD + h, drop_p);
          }
          cx.x = cx.x * resc + (pa2 * ka.x + pb2 * kb.x);
          cx.y = cx.y * resc + (pa2 * ka.y + pb2 * kb.y);
          cx.z = cx.z * resc + (pa2 * ka.z + pb2 * kb.z);
          cx.w = cx.w * resc + (pa2 * ka.w + pb2 * kb.w);
          m = mn;
        }
        if (part >= 0) {
          float* pbuf = g_pbuf + (size_t)part * 3 * D + 4 * c4;
          *reinterpret_cast<float4*>(pbuf) = make_float4(m, m, m, m);
          *reinterpret_cast<float4*>(pbuf + D) = make_float4(ssum, ssum, ssum, ssum);
          *reinterpret_cast<float4*>(pbuf + 2 * D) = cx;
        } else {
          const float rs2 = e1 > e0 ? __builtin_amdgcn_rcpf(ssum) : 0.f;
          float4* qp = reinterpret_cast<float4*>(&sQ[la * LDS_STRIDE + 4 * c4]);
          const float4 q4 = *qp;
          *qp = make_float4(cx.x * rs2 + q4.x, cx.y * rs2 + q4.y, cx.z * rs2 + q4.z, cx.w * rs2 + q4.w);
        }
      }
    }
    __syncthreads();
    // LayerNorm of the context rows: 8 threads per atom row
    {
      const int rr = tid >> 3, sb = tid & 7;
      if (rr < natom && part < 0) {
        float4 tt[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          tt[i] = *reinterpret_cast<const float4*>(&sQ[rr * LDS_STRIDE + 4 * (sb + 8 * i)]);
          s += f4sum(tt[i]);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        const float mean = s * (1.0f / D);
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dx = tt[i].x - mean, dy = tt[i].y - mean, dz = tt[i].z - mean, dw = tt[i].w - mean;
          v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c4 = sb + 8 * i;
          const float4 g = *reinterpret_cast<const float4*>(&sPar[2 * D + 4 * c4]);
          const float4 be = *reinterpret_cast<const float4*>(&sPar[3 * D + 4 * c4]);
          float4 y;
          float inv;
          inv = rstd * g.x; y.x = tt[i].x * inv + (be.x - mean * inv);
          inv = rstd * g.y; y.y = tt[i].y * inv + (be.y - mean * inv);
          inv = rstd * g.z; y.z = tt[i].z * inv + (be.z - mean * inv);
          inv = rstd * g.w; y.w = tt[i].w * inv + (be.w - mean * inv);
          reinterpret_cast<float4*>(g_ctx)[(size_t)(abeg + rr) * 32 + c4] = y;
        }
      }
    }
    if (!more) break;
    __syncthreads();  // every wave is done with sQ / sE / sOff before the next tile is staged
    vt = vn;
  }
}

// ---- lean-LDS edge kernel on 32-edge tiles: five workgroups per CU ----------------------------------------------------
//
// Same phases as edge_kernel_lean on one 32 x 128 buffer (29 KB of LDS, <= 96 VGPRs): the launch time of the edge path is
// (rounds of tiles) x (tile latency) and the latency is a dependency chain, not matrix work, so more, smaller tiles in
// flight per CU trade weight traffic (the kernels are re-read per tile) for matrix-pipe occupancy.  Weights stream in
// quarter slabs through two 4 x float4 buffers.
constexpr int TQ32 = 16;  // atoms per 32-edge tile
__device__ __forceinline__ void load_w_quarter(const float* __restrict__ Wp, int cb, int lane, int qt, float4 (&w)[4]) {
  const float4* __restrict__ wsrc = reinterpret_cast<const float4*>(Wp) + cb * (16 * 64) + qt * (4 * 64) + lane;
#pragma unroll
  for (int t = 0; t < 4; ++t) w[t] = wsrc[t * 64];
}
__device__ __forceinline__ void mma_quarter1(const float* __restrict__ sX, const float4 (&w)[4], int lane, int qt, f32x16& acc) {
  const float* xrow = sX + (lane & 31) * LDS_STRIDE + 4 * (lane >> 5) + 32 * qt;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(xrow + 8 * t);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].x, a.x, acc, 0, 0, 0);  // transposed product (see mma_half2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].y, a.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].z, a.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].w, a.w, acc, 0, 0, 0);
  }
}
// acc (+)= X[32 x 128] . W[:, 32 wave .. +32]; wa / wb hold quarters 0 / 1 on entry; `next` (or null): the kernel whose
// quarters 0 / 1 are requested as soon as the buffers free up
__device__ __forceinline__ void gemm32(const float* __restrict__ sX, const float* __restrict__ Wp, const float* __restrict__ next,
                                       int wave, int lane, float4 (&wa)[4], float4 (&wb)[4], f32x16& acc) {
  mma_quarter1(sX, wa, lane, 0, acc);
  __builtin_amdgcn_sched_barrier(0);
  load_w_quarter(Wp, wave, lane, 2, wa);
  __builtin_amdgcn_sched_barrier(0);
  mma_quarter1(sX, wb, lane, 1, acc);
  __builtin_amdgcn_sched_barrier(0);
  load_w_quarter(Wp, wave, lane, 3, wb);
  __builtin_amdgcn_sched_barrier(0);
  mma_quarter1(sX, wa, lane, 2, acc);
  __builtin_amdgcn_sched_barrier(0);
  if (next) load_w_quarter(next, wave, lane, 0, wa);
  __builtin_amdgcn_sched_barrier(0);
  mma_quarter1(sX, wb, lane, 3, acc);
  __builtin_amdgcn_sched_barrier(0);
  if (next) load_w_quarter(next, wave, lane, 1, wb);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void dump_t1(float* __restrict__ sT, const f32x16& acc, int wave, int lane, const float* __restrict__ sBias) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 32 * wave + 8 * j + 4 * (lane >> 5);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sBias) bv = *reinterpret_cast<const float4*>(sBias + c);
    *reinterpret_cast<float4*>(&sT[(lane & 31) * LDS_STRIDE + c]) =
        make_float4(acc[4 * j] + bv.x, acc[4 * j + 1] + bv.y, acc[4 * j + 2] + bv.z, acc[4 * j + 3] + bv.w);
  }
}

__global__ __launch_bounds__(256, 5) void edge_kernel_lean32(EdgeArgs a) {
  constexpr int TEK = 32;
  __shared__ __attribute__((aligned(16))) float sA[TEK * LDS_STRIDE];   // G -> U -> ang = c[j]*geom' -> K
  __shared__ __attribute__((aligned(16))) float sQ[TQ32 * LDS_STRIDE];  // P1 rows, then query rows of the tile's atoms, then context
  __shared__ __attribute__((aligned(16))) float sE[TEK * NHEAD];
  __shared__ __attribute__((aligned(16))) float sPar[5 * D];
  __shared__ int sCol[TEK], sCtr[TEK], sOff[TQ32 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tix = a.xcd_remap ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
  const EdgeTile tile = a.tiles[tix];
  const int part = a.tile_part ? a.tile_part[tix] : -1;
  const int eb = tile.edge_begin, ne = tile.edge_end - eb, natom = tile.atom_end - tile.atom_begin;
  const int r = tid >> 3, sub = tid & 7;  // row-pass mapping: 8 threads per edge row, float4 chunks sub, sub+8, sub+16, sub+24

  // prologue: every load unguarded from clamped rows (see edge_kernel_lean)
  const int nem1 = ne > 0 ? ne - 1 : 0;
  const int rs = r < ne ? r : nem1;
  float4 wa[4], wb[4];
  load_w_quarter(a.p.W2p, wave, lane, 0, wa);
  load_w_quarter(a.p.W2p, wave, lane, 1, wb);
  const int32_t* pa = tid < TEK ? (ne > 0 ? a.edge_col + eb + min(tid, nem1) : a.edge_offset)
                                : a.edge_offset + tile.atom_begin + min(tid - TEK, natom);
  const int32_t* pb = ne > 0 ? a.edge_row + eb + min(tid & (TEK - 1), nem1) : a.edge_offset;
  const int va = *pa, vb = *pb;
  const float bkc = a.p.bk[tid & (D - 1)];
  const float par0 = (tid < D ? a.p.lng_g : a.p.lng_b)[tid & (D - 1)];
  const float par1 = (tid < D ? a.p.ln_g : a.p.ln_b)[tid & (D - 1)];
  const int qa = tid >> 5, qc = tid & 31;  // atom rows qa, qa + 8 of the P1 / query staging
  const float4* const p14 = reinterpret_cast<const float4*>(a.P1) + (size_t)tile.atom_begin * 32 + qc;
  const float4 p1a = p14[(size_t)min(qa, natom - 1) * 32], p1b = p14[(size_t)min(qa + 8, natom - 1) * 32];
  float4 greg[4];
  {
    const float4* grow = reinterpret_cast<const float4*>(ne > 0 ? a.geom + (size_t)(eb + rs) * D : a.P1);
#pragma unroll
    for (int i = 0; i < 4; ++i) greg[i] = grow[sub + 8 * i];
  }
  if (tid < TEK) {
    sCol[tid] = tid < ne ? va : 0;
    sCtr[tid] = tid < ne ? vb : 0;
  } else if (tid - TEK <= natom) {
    sOff[tid - TEK] = part >= 0 ? (tid == TEK ? 0 : ne) : va - eb;
  }
  sPar[tid] = par0;
  sPar[2 * D + tid] = par1;
  if (tid < D) sPar[4 * D + tid] = bkc;
  *reinterpret_cast<float4*>(&sQ[qa * LDS_STRIDE + 4 * qc]) = p1a;
  *reinterpret_cast<float4*>(&sQ[(qa + 8) * LDS_STRIDE + 4 * qc]) = p1b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (r >= ne) greg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * (sub + 8 * i)]) = greg[i];
  }
  __syncthreads();
  // U = G . W2
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  gemm32(sA, a.p.W2p, nullptr, wave, lane, wa, wb, acc);
  __syncthreads();  // every wave is done reading G
  dump_t1(sA, acc, wave, lane, nullptr);
  __syncthreads();

  // row pass (attention.py:141-157)
  if (r < ne) {
    const int ctr = sCtr[r], nb = sCol[r];
    const float4* crow = reinterpret_cast<const float4*>(a.c) + (size_t)nb * 32;
    const float* p1 = sQ + (ctr - tile.atom_begin) * LDS_STRIDE;
    const float4* p3 = reinterpret_cast<const float4*>(a.P3) + (size_t)nb * 32;
    float4 p3r[4], cn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p3r[i] = p3[sub + 8 * i];
#pragma unroll
    for (int i = 0; i < 4; ++i) cn[i] = crow[sub + 8 * i];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c4 = sub + 8 * i;
      const float4 u = *reinterpret_cast<const float4*>(&sA[r * LDS_STRIDE + 4 * c4]);
      const float4 v = f4add(f4add(*reinterpret_cast<const float4*>(p1 + 4 * c4), u), p3r[i]);
      greg[i] = f4add(f4swish(v), greg[i]);
      s += f4sum(greg[i]);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    const float mean = s * (1.0f / D);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dx = greg[i].x - mean, dy = greg[i].y - mean, dz = greg[i].z - mean, dw = greg[i].w - mean;
      v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c4 = sub + 8 * i;
      const float4 g = *reinterpret_cast<const float4*>(&sPar[4 * c4]);
      const float4 be = *reinterpret_cast<const float4*>(&sPar[D + 4 * c4]);
      float4 y;
      float inv;
      inv = rstd * g.x; y.x = greg[i].x * inv + (be.x - mean * inv);
      inv = rstd * g.y; y.y = greg[i].y * inv + (be.y - mean * inv);
      inv = rstd * g.z; y.z = greg[i].z * inv + (be.z - mean * inv);
      inv = rstd * g.w; y.w = greg[i].w * inv + (be.w - mean * inv);
      reinterpret_cast<float4*>(a.geom_out ? a.geom_out : a.geom)[(size_t)(eb + r) * 32 + c4] = y;
      *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * c4]) = f4mul(cn[i], y);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(&sA[r * LDS_STRIDE + 4 * (sub + 8 * i)]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  load_w_quarter(a.p.Wkp, wave, lane, 0, wa);
  load_w_quarter(a.p.Wkp, wave, lane, 1, wb);
  const float4* const q4p = reinterpret_cast<const float4*>(a.q) + (size_t)tile.atom_begin * 32 + qc;
  const float4 q0 = q4p[(size_t)min(qa, natom - 1) * 32], q1 = q4p[(size_t)min(qa + 8, natom - 1) * 32];
  __syncthreads();  // ang complete; nobody reads the P1 rows any more
  *reinterpret_cast<float4*>(&sQ[qa * LDS_STRIDE + 4 * qc]) = q0;
  *reinterpret_cast<float4*>(&sQ[(qa + 8) * LDS_STRIDE + 4 * qc]) = q1;
  __builtin_amdgcn_sched_barrier(0);
  // K = ang . Wk + bk
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  gemm32(sA, a.p.Wkp, nullptr, wave, lane, wa, wb, acc);
  __syncthreads();  // every wave is done reading ang
  dump_t1(sA, acc, wave, lane, sPar + 4 * D);
  __syncthreads();
  // logits: thread = (edge row, head)
  {
    const int n = tid >> 3, hh = tid & 7;
    if (n < ne) {
      const float* qrow = sQ + (sCtr[n] - tile.atom_begin) * LDS_STRIDE + HDIM * hh;
      const float* krow = sA + n * LDS_STRIDE + HDIM * hh;
      float e = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 q4 = *reinterpret_cast<const float4*>(qrow + 4 * j);
        const float4 k4 = *reinterpret_cast<const float4*>(krow + 4 * j);
        e += (q4.x * 0.25f) * k4.x; e += (q4.y * 0.25f) * k4.y; e += (q4.z * 0.25f) * k4.z; e += (q4.w * 0.25f) * k4.w;
      }
      sE[n * NHEAD + hh] = e;
    }
  }
  __syncthreads();
  // softmax + context + residual (online form, see edge_kernel_w8)
  {
    const int lgp = tid >> 5, c4 = tid & 31, h = c4 >> 2;
    for (int la = lgp; la < natom; la += 8) {
      const int e0 = sOff[la], e1 = sOff[la + 1];
      float m = -INFINITY, ssum = 0.f;
      float4 cx = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int n = e0; n < e1; n += 2) {
        const bool two = n + 1 < e1;
        const int n1 = two ? n + 1 : n;
        const float ea = sE[n * NHEAD + h];
        const float eb2 = two ? sE[n1 * NHEAD + h] : -INFINITY;
        const float4 ka = *reinterpret_cast<const float4*>(&sA[n * LDS_STRIDE + 4 * c4]);
        const float4 kb = *reinterpret_cast<const float4*>(&sA[n1 * LDS_STRIDE + 4 * c4]);
        const float mn = fmaxf(m, fmaxf(ea, eb2));
        const float resc = fast_exp(m - mn);
        float pa2 = fast_exp(ea - mn), pb2 = fast_exp(eb2 - mn);
        ssum = ssum * resc + (pa2 + pb2);
        if (a.attn_drop_p > 0.f) {
          pa2 *= drop_scale(a.attn_drop_seed, a.attn_drop_tag, (size_t)(eb + n) * NHEAD + h, a.attn_drop_p);
          pb2 *= drop_scale(a.attn_drop_seed, a.attn_drop_tag, (size_t)(eb + n1) * NHEAD + h, a.attn_drop_p);
        }
        cx.x = cx.x * resc + (pa2 * ka.x + pb2 * kb.x);
        cx.y = cx.y * resc + (pa2 * ka.y + pb2 * kb.y);
        cx.z = cx.z * resc + (pa2 * ka.z + pb2 * kb.z);
        cx.w = cx.w * resc + (pa2 * ka.w + pb2 * kb.w);
        m = mn;
      }
      if (part >= 0) {
        float* pbuf = a.part_buf + (size_t)part * 3 * D + 4 * c4;
        *reinterpret_cast<float4*>(pbuf) = make_float4(m, m, m, m);
        *reinterpret_cast<float4*>(pbuf + D) = make_float4(ssum, ssum, ssum, ssum);
        *reinterpret_cast<float4*>(pbuf + 2 * D) = cx;
      } else {
        const float rs2 = e1 > e0 ? __builtin_amdgcn_rcpf(ssum) : 0.f;
        float4* qp = reinterpret_cast<float4*>(&sQ[la * LDS_STRIDE + 4 * c4]);
        const float4 q4 = *qp;
        *qp = make_float4(cx.x * rs2 + q4.x, cx.y * rs2 + q4.y, cx.z * rs2 + q4.z, cx.w * rs2 + q4.w);
      }
    }
  }
  __syncthreads();
  // LayerNorm of the context rows: 8 threads per atom row
  {
    const int rr = tid >> 3, sb = tid & 7;
    if (rr < natom && part < 0) {
      float4 t[4];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t[i] = *reinterpret_cast<const float4*>(&sQ[rr * LDS_STRIDE + 4 * (sb + 8 * i)]);
        s += f4sum(t[i]);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      const float mean = s * (1.0f / D);
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dx = t[i].x - mean, dy = t[i].y - mean, dz = t[i].z - mean, dw = t[i].w - mean;
        v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c4 = sb + 8 * i;
        const float4 g = *reinterpret_cast<const float4*>(&sPar[2 * D + 4 * c4]);
        const float4 be = *reinterpret_cast<const float4*>(&sPar[3 * D + 4 * c4]);
        float4 y;
        float inv;
        inv = rstd * g.x; y.x = t[i].x * inv + (be.x - mean * inv);
        inv = rstd * g.y; y.y = t[i].y * inv + (be.y - mean * inv);
        inv = rstd * g.z; y.z = t[i].z * inv + (be.z - mean * inv);
        inv = rstd * g.w; y.w = t[i].w * inv + (be.w - mean * inv);
        reinterpret_cast<float4*>(a.ctx)[(size_t)(tile.atom_begin + rr) * 32 + c4] = y;
      }
    }
  }
}

void launch_edge(const EdgeArgs& a, hipStream_t s) {
  if (a.n_tile <= 0) return;
  if (a.lean && a.lean_wgs > 0 && a.g_update && a.tile_rows == 64) {
    const int nwg = a.n_tile < a.lean_wgs ? a.n_tile : a.lean_wgs;
    hipLaunchKernelGGL(edge_kernel_leanp, dim3(nwg), dim3(256), 0, s, a);
    return;
  }
  if (a.lean && a.g_update && a.tile_rows == 64) {
    hipLaunchKernelGGL(edge_kernel_lean, dim3(a.n_tile), dim3(256), 0, s, a);
    return;
  }
  if (a.lean && a.g_update && a.tile_rows == 32) {
    hipLaunchKernelGGL(edge_kernel_lean32, dim3(a.n_tile), dim3(256), 0, s, a);
    return;
  }
  if (a.waves8 && a.g_update && a.tile_rows == 32) {
    hipLaunchKernelGGL(edge_kernel_w8<1>, dim3(a.n_tile), dim3(256), 0, s, a);
    return;
  }
  if (a.waves8 && a.g_update && a.tile_rows == 64) {
    hipLaunchKernelGGL(edge_kernel_w8<2>, dim3(a.n_tile), dim3(512), 0, s, a);
    return;
  }
  const dim3 grid(a.n_tile), block(256);
  if (a.tile_rows == 32) {
    if (a.g_update) hipLaunchKernelGGL((edge_kernel<true, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((edge_kernel<false, 1>), grid, block, 0, s, a);
  } else {
    if (a.g_update) hipLaunchKernelGGL((edge_kernel<true, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((edge_kernel<false, 2>), grid, block, 0, s, a);
  }
}

// ---- basis kernel ----------------------------------------------------------------------------------

// exp(-(x - c)^2 / 0.25)  (custom_layers.py:63-65, width 0.5 squared at :51)
__device__ __forceinline__ float gauss(float x, float c) {
  const float d = x - c;
  return expf(-(d * d) / 0.25f);
}
// v_exp_f32 form for the fused basis MLP: |abs error| <= ~2e-8 (value * |arg| * 6e-8 peaks at arg = -1)
__device__ __forceinline__ float gauss_fast(float x, float c) {
  const float d = x - c;
  return fast_exp(-(d * d) * 4.0f);
}

__global__ __launch_bounds__(256) void basis_kernel(BasisParams p, const float* __restrict__ dist,
                                                    const float* __restrict__ weight, int n_edge,
                                                    float* __restrict__ geom) {
  __shared__ float sG[TB][2 * NG];
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * TB;
  const int ne = min(TB, n_edge - e0);
  const int col = tid & (D - 1), half = tid >> 7;
  float wd[NG], ww[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) {  // this thread's column of both [20,128] kernels, in flight during the basis phase
    wd[k] = p.Wd[k * D + col];
    ww[k] = p.Ww[k * D + col];
  }
  const float bd = p.bd[col], bw = p.bw[col];
  {  // Gaussian expansions of the tile's edges; operands requested together from clamped rows, values masked afterwards
    constexpr int NV = (TB * 2 * NG + 255) / 256;
    float x[NV], cc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = min(tid + 256 * j, TB * 2 * NG - 1), e = min(i / (2 * NG), ne - 1), k = i % (2 * NG);
      x[j] = (k < NG ? dist : weight)[e0 + e];
      cc[j] = k < NG ? p.cd[k] : p.cw[k - NG];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + 256 * j;
      if (i < TB * 2 * NG) sG[i / (2 * NG)][i % (2 * NG)] = i / (2 * NG) < ne ? gauss_fast(x[j], cc[j]) : 0.f;
    }
  }
  __syncthreads();
#pragma unroll 2
  for (int e = half; e < ne; e += 2) {
    float ad = 0.f, aw = 0.f;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      ad += sG[e][k] * wd[k];
      aw += sG[e][NG + k] * ww[k];
    }
    // neighbor_d * neighbor_w (scann_model.py:381-389)
    geom[(size_t)(e0 + e) * D + col] = swishf(ad + bd) * swishf(aw + bw);
  }
}

__global__ void basis_raw_kernel(const float* __restrict__ cd, const float* __restrict__ dist, int n_edge,
                                 float* __restrict__ gd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_edge * NG) gd[i] = gauss(dist[i / NG], cd[i % NG]);
}

void launch_basis(const BasisParams& p, const float* dist, const float* weight, int n_edge, float* geom,
                  hipStream_t s) {
  if (n_edge <= 0) return;
  hipLaunchKernelGGL(basis_kernel, dim3((n_edge + TB - 1) / TB), dim3(256), 0, s, p, dist, weight, n_edge, geom);
}

void launch_basis_raw(const float* cd, const float* dist, int n_edge, float* gd, hipStream_t s) {
  if (n_edge <= 0) return;
  const int n = n_edge * NG;
  hipLaunchKernelGGL(basis_raw_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cd, dist, n_edge, gd);
}

// ---- embedding LUT (weight-load time) ----------------------------------------------------------------

// lut[s,:] = swish(E[s,:] . W + b): Embedding (scann_model.py:362) followed by dense_embed (:373) has only
// n_atoms distinct results, so it is folded into a table when weights are loaded.
__global__ void embed_lut_kernel(const float* __restrict__ emb, const float* __restrict__ W,
                                 const float* __restrict__ b, int emb_dim, float* __restrict__ lut) {
  const int sp = blockIdx.x, col = threadIdx.x;
  float acc = 0.f;
  for (int k = 0; k < emb_dim; ++k) acc += emb[sp * emb_dim + k] * W[k * D + col];
  lut[sp * D + col] = swishf(acc + b[col]);
}

void launch_embed_lut(const float* emb, const float* W, const float* b, int n_species, int emb_dim, float* lut,
                      hipStream_t s) {
  hipLaunchKernelGGL(embed_lut_kernel, dim3(n_species), dim3(D), 0, s, emb, W, b, emb_dim, lut);
}

// ---- general embedding kernel (use_ring / feature="cgcnn") -----------------------------------------------

// c0 = swish(dense_embed(concat[embed, extra_embed(ring)]))  (scann_model.py:361-374): `embed` is the Embedding row
// (feature="atomic", :362) or Dense(92 -> emb) of the CGCNN features without activation (:365); ring/aromatic flags go
// through Dense(2 -> 10) (:368) and are concatenated (:371).  8 atoms per 128-thread workgroup.
__global__ __launch_bounds__(128) void embed_kernel(EmbedArgs a) {
  __shared__ float sV[8][160];
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
  for (int la = 0; la < na; ++la) {
    float acc = 0.f;
    for (int k = 0; k < cin; ++k) acc += sV[la][k] * a.Wde[k * D + tid];
    a.c0[(size_t)(a0 + la) * D + tid] = swishf(acc + a.bde[tid]);
  }
}

void launch_embed(const EmbedArgs& a, hipStream_t s) {
  if (a.n_atom <= 0) return;
  hipLaunchKernelGGL(embed_kernel, dim3((a.n_atom + 7) / 8), dim3(128), 0, s, a);
}

// ---- readout kernel ----------------------------------------------------------------------------------

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// One workgroup per structure.  GlobalAttention.call (attention.py:267-318) on the real atoms only (the
// multiplicative atom mask zeroes every padded term).  The pair energies E[i][j] = k_i . q_j (:279) are 32x32 MFMA
// tiles; wave w contracts features [32w, 32w+32) of every tile, masks the diagonal (:282-285) and rows/columns past
// the structure, keeps per-lane partial row sums over all column tiles, reduces them across the 32 columns once
// per row tile, and the four waves' partial sums are added in fixed order (deterministic).
__global__ __launch_bounds__(256) void readout_kernel(ReadoutArgs a) {
  extern __shared__ float sDyn[];  // [4][npad] partial row sums, then [npad] scores / attention
  __shared__ float sRep[D];
  __shared__ float sRed[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a0 = a.mol_offset[blockIdx.x];
  const int n = a.mol_offset[blockIdx.x + 1] - a0;
  const int T = (n + 31) >> 5, npad = T << 5;
  float* sPart = sDyn;            // [4][npad]
  float* sAgg = sDyn + 4 * npad;  // [npad]
  const float4* gq4 = reinterpret_cast<const float4*>(a.gq) + (size_t)a0 * 32;
  const float4* gk4 = reinterpret_cast<const float4*>(a.gk) + (size_t)a0 * 32;
  const int r = lane & 31, h = lane >> 5;

  for (int it = 0; it < T; ++it) {
    const int irow = it * 32 + r;
    float4 ka[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)  // features 32*wave + 8t + 4h .. +3 of key row irow
      ka[t] = irow < n ? gk4[(size_t)irow * 32 + 8 * wave + 2 * t + h] : make_float4(0.f, 0.f, 0.f, 0.f);
    float part[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) part[i] = 0.f;
    for (int jt = 0; jt < T; ++jt) {
      const int jrow = jt * 32 + r;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 qb = jrow < n ? gq4[(size_t)jrow * 32 + 8 * wave + 2 * t + h] : make_float4(0.f, 0.f, 0.f, 0.f);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t].x, qb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t].y, qb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t].z, qb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t].w, qb.w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int gi = it * 32 + acc_row(i, lane);  // key index (row), this lane's column is query index jrow
        part[i] += (gi != jrow) ? acc[i] : 0.f;     // mask_center (:282-285); padded rows/cols contribute exact zeros
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = part[i];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      if (r == 0) sPart[wave * npad + it * 32 + acc_row(i, lane)] = v;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256)  // agg_i (:289-292)
    sAgg[i] = ((sPart[i] + sPart[npad + i]) + sPart[2 * npad + i]) + sPart[3 * npad + i];
  __syncthreads();
  // normalise + softmax over atoms by wave 0 (:295-302)
  if (wave == 0) {
    float nrm = 1.0f;
    if (a.use_ga_norm) {
      float ss = 0.f;
      for (int i = lane; i < n; i += 64) ss += sAgg[i] * sAgg[i];
      nrm = sqrtf(wave_sum(ss));  // tf.linalg.normalize: no epsilon (n == 1 -> 0/0 = NaN as in the reference)
    }
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) {
      const float v = a.use_ga_norm ? sAgg[i] / nrm : sAgg[i];
      sAgg[i] = v;
      m = fmaxf(m, v);
    }
    m = wave_max(m);
    float ss = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float e = expf(sAgg[i] - m);
      sAgg[i] = e;
      ss += e;
    }
    ss = wave_sum(ss);
    for (int i = lane; i < n; i += 64) {
      const float at = sAgg[i] / ss;
      sAgg[i] = at;
      a.ga_attn[a0 + i] = at;
    }
  }
  __syncthreads();
  // rep = sum_i attn_i k_i  (:314-316): two half-sums over interleaved atoms per feature, fixed order
  {
    const int f = tid & (D - 1), half = tid >> 7;
    float rsum = 0.f;
    for (int i = half; i < n; i += 2) rsum += sAgg[i] * a.gk[(size_t)(a0 + i) * D + f];
    if (half == 1) sRep[f] = rsum;
    __syncthreads();
    if (half == 0) sRep[f] = rsum + sRep[f];
  }
  __syncthreads();
  // bf_property + predict_property (scann_model.py:437-447)
  float part = 0.f;
  if (tid < D) {
    float hsum = 0.f;
#pragma unroll 8
    for (int k = 0; k < D; ++k) hsum += sRep[k] * a.p.Wb[k * D + tid];
    part = swish_exact(hsum + a.p.bb[tid]) * a.p.wo[tid];
  }
  part = wave_sum(part);
  if (lane == 0) sRed[wave] = part;
  __syncthreads();
  if (tid == 0) {
    float y = (sRed[0] + sRed[1]) + a.p.bo[0];
    if (a.relu_out) y = fmaxf(y, 0.f);  // mrelu forward (custom_layers.py:15)
    a.y[blockIdx.x] = y;
  }
}

void launch_readout(const ReadoutArgs& a, hipStream_t s) {
  if (a.n_struct <= 0) return;
  const size_t npad = (size_t)((a.max_atoms + 31) / 32) * 32;
  const size_t lds = 5 * npad * sizeof(float);  // 4 partial-sum rows + the score row
  hipLaunchKernelGGL(readout_kernel, dim3(a.n_struct), dim3(256), lds, s, a);
}

}  // namespace scann
