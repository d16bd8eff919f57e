// Hand-written CDNA4 (gfx950) kernels for the SCANN / SCANN+ forward hot path.
//
// Layout: structures are packed -- atoms [n_atom,128] and CSR edges [n_edge,128] -- so
// the reference's padded [B,M,N,d] tensors (attention.py:136-212) are never materialised.  Every
// dense projection is a split-fp16 MFMA product with fp32 accumulation (v_mfma_f32_32x32x16_f16, three
// products per operand pair: fp32 accuracy at 3/16 of the fp32-MFMA time; scheme and evidence in the
// edge-tile section) of a row tile staged in LDS as hi / lo planes against a 128x128 weight that each
// wave streams straight from L2 into registers in fragment order (no LDS staging of weights: a wave
// owns 32 output columns, so every weight element is read by exactly one wave of the workgroup).
//
// Kernels (one launch each per LocalAttention iteration, see scann_runtime.cpp: run_forward):
//   atom_kernel   : [ResidualNorm of previous layer] -> centres c; P1 = c W1 + bg, P3 = c W3, q = c Wq + bq
//                   (attention.py:37-40, :160, and the centre/neighbour thirds of filter_geo :142-151)
//   edge_kernel   : per tile of <=64 edges (whole atoms): U = G W2; geom' = LN_g(swish(U+P1[i]+P3[j])+G)
//                   (:141-153); ang = c[j]*geom' (:136,:157); K = ang Wk + bk (:163);
//                   per (atom, head) softmax over its edges and ctx = LN(sum attn K + q) (:180-214)
//   readout_kernel: GlobalAttention + bf_property + predict_property (attention.py:267-318,
//                   scann_model.py:437-447), one workgroup per structure (pair energies: exact-fp32 MFMA)
//   basis_kernel  : Gaussian expansion + neighbor_d/neighbor_w MLP (custom_layers.py:63-65,
//                   scann_model.py:378-389)
#include "scann_internal.h"
#include "scann_mma.h"

namespace scann {

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

// ---- atom-tile kernel ------------------------------------------------------------------------------
//
// One workgroup (4 waves) per tile of 64 (or, for small launches, 32: launch_atom) atom rows, three (four) workgroups per CU; every projection is a split-fp16 MFMA GEMM
// of the tile (hi / lo planes in ONE LDS buffer in which x, the ResidualNorm hidden row, the centres and -- readout --
// swish(after_Lc) take turns) against a 128x128 weight streamed from L2 in two halves of 32 VGPRs, the next half always
// requested while the current one multiplies.  Everything between the GEMMs happens in the accumulator layout (lane = row,
// 16 columns): bias, swish, dropout, the residual (x stays in registers, exact fp32), LayerNorm statistics (lane pair ->
// LDS -> the row's four waves), and the output stores (16-byte pieces).
//   FFN  : c = LN(x + drop(W2 swish(W1 x + b1) + b2))                       ResidualNorm of the previous layer, attention.py:37-40
//   MODE 0: P1 = c W1 + bg, P3 = c W3, q = c Wq + bq                        centre / neighbour thirds of filter_geo :142-151, query :160
//   MODE 1: q only (base branch)           MODE 2: z = swish(c Wa + ba); gq = z Wgq + b, gk = z Wgk + b   (scann_model.py:424, attention.py:269-272)

// EX: exact-fp32 projections (scann_mma.h: the fallback run_forward takes when the split-fp16 range guard fired)
template <bool FFN, int MODE, int RT, bool EX = false>
__global__ __launch_bounds__(256, RT == 2 ? 3 : 4) void atom_kernel(AtomArgs a) {
#pragma clang fp contract(off)  // fusions are written out: both row-tile copies of a formula round alike (see edge_kernel)
  constexpr int TAR = 32 * RT;  // atom rows per tile
  __shared__ __attribute__((aligned(16))) unsigned char sTile[2 * TAR * PLANE_STRIDE * 2];  // hi / lo planes of the current GEMM input
  __shared__ __attribute__((aligned(16))) float sRed[TAR * 8];  // LayerNorm partial statistics [row][wave][mean, m2]
  __shared__ __attribute__((aligned(16))) float sPar[7 * D];   // bf1 | bf2 | lnr_g | lnr_b | bA | bC | bD
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5, cbase = 32 * wave + 4 * lh;
  // (row_tab: the atom tiles of the structures that stay on this path in a batch shared with the structure-resident kernel)
  const int row0 = a.row_tab ? a.row_tab[2 * blockIdx.x] : blockIdx.x * TAR;
  const int nrows = a.row_tab ? a.row_tab[2 * blockIdx.x + 1] : min(TAR, a.n_atom - row0);
  constexpr float WINV = EX ? 1.0f : 1.0f / WSCALE;  // (exact images are unscaled)
  // first projection after the (optional) ResidualNorm: W1 (mode 0), Wq (mode 1), after_Lc (mode 2)
  const _Float16* const firstW = MODE == 1 ? a.WCh : a.WAh;

  STAMP(a.stamps, 0);
  WRegs<EX> wr;
  load_whalf<EX>(wr, FFN ? a.Wf1h : firstW, wave, lane, 0);
  load_whalf<EX>(wr, FFN ? a.Wf1h : firstW, wave, lane, 1);
  {  // bias / LayerNorm rows -> sPar (absent ones read a valid dummy row and are never used)
    const float* const tab[7] = {FFN ? a.bf1 : a.bC, FFN ? a.bf2 : a.bC, FFN ? a.lnr_g : a.bC, FFN ? a.lnr_b : a.bC,
                                 MODE != 1 ? a.bA : a.bC, a.bC, MODE == 2 ? a.bD : a.bC};
    float pv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pv[i] = (tid < D ? tab[2 * i] : tab[2 * i + 1 < 7 ? 2 * i + 1 : 6])[tid & (D - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (256 * i + tid < 7 * D) sPar[256 * i + tid] = pv[i];
  }
  // x rows of the tile in the accumulator layout, straight into registers (they stay there for the residual); rows clamped,
  // never guarded, and zero-filled afterwards so that the MFMAs see defined data
  float4 xr[RT][4];
  unsigned ooff[RT];  // byte offset of (output row, this lane's first column) in an [n_atom,128] tensor
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int rc = row0 + min(lrow + 32 * rt, nrows - 1);
    const int src = a.x_index ? a.x_index[rc] : rc;
    ooff[rt] = ((unsigned)(row0 + lrow + 32 * rt) * D + cbase) * 4;
    const unsigned soff = ((unsigned)src * D + cbase) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) xr[rt][j] = ld4(a.x, soff + 32 * j);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = lrow + 32 * rt;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 v = row < nrows ? xr[rt][j] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!FFN && row < nrows) {
        if (a.drop_p > 0.f) {  // training: Dropout after dense_embed (scann_model.py:374)
          const size_t e = (size_t)(row0 + row) * D + cbase + 8 * j;
          v.x *= drop_scale(a.drop_seed, a.drop_tag, e, a.drop_p);
          v.y *= drop_scale(a.drop_seed, a.drop_tag, e + 1, a.drop_p);
          v.z *= drop_scale(a.drop_seed, a.drop_tag, e + 2, a.drop_p);
          v.w *= drop_scale(a.drop_seed, a.drop_tag, e + 3, a.drop_p);
        }
        st4(a.c, ooff[rt] + 32 * j, v);  // centres = staged rows (layer 0 / no ResidualNorm)
      }
      xr[rt][j] = v;
      tile_store<EX, TAR>(sTile, row, cbase + 8 * j, v);
    }
  }
  __syncthreads();
  STAMP(a.stamps, 1);

  f32x16 acc[RT];
  if (FFN) {
    // ResidualNorm (attention.py:37-40): h = swish(x W1 + b1)
    gemm_tile_x<EX, true, TAR, RT>(sTile, wr, a.Wf2h, wave, lane, acc);
    STAMP(a.stamps, 2);
    __syncthreads();  // every wave is done reading the x planes
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = lrow + 32 * rt;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bv = *reinterpret_cast<const float4*>(&sPar[cbase + 8 * j]);
        const float4 pre = make_float4(fmaf(acc[rt][4 * j], WINV, bv.x), fmaf(acc[rt][4 * j + 1], WINV, bv.y),
                                       fmaf(acc[rt][4 * j + 2], WINV, bv.z), fmaf(acc[rt][4 * j + 3], WINV, bv.w));
        const float4 hh = f4swish(pre);
        if (a.keep_pre1 && row < nrows) {  // training forward: kept for the backward
          st4(a.keep_pre1, ooff[rt] + 32 * j, pre);
          st4(a.keep_H1, ooff[rt] + 32 * j, hh);
        }
        tile_store<EX, TAR>(sTile, row, cbase + 8 * j, hh);
      }
    }
    __syncthreads();
    STAMP(a.stamps, 3);
    // y = h W2 + b2 ; t = x + drop(y)
    gemm_tile_x<EX, true, TAR, RT>(sTile, wr, firstW, wave, lane, acc);
    STAMP(a.stamps, 4);
    float mean32[RT], m2[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = lrow + 32 * rt;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bv = *reinterpret_cast<const float4*>(&sPar[D + cbase + 8 * j]);
        float4 y = make_float4(fmaf(acc[rt][4 * j], WINV, bv.x), fmaf(acc[rt][4 * j + 1], WINV, bv.y),
                               fmaf(acc[rt][4 * j + 2], WINV, bv.z), fmaf(acc[rt][4 * j + 3], WINV, bv.w));
        if (a.drop_p > 0.f) {  // attention.py:29 (training)
          const size_t e = (size_t)(row0 + row) * D + cbase + 8 * j;
          y.x *= drop_scale(a.drop_seed, a.drop_tag, e, a.drop_p);
          y.y *= drop_scale(a.drop_seed, a.drop_tag, e + 1, a.drop_p);
          y.z *= drop_scale(a.drop_seed, a.drop_tag, e + 2, a.drop_p);
          y.w *= drop_scale(a.drop_seed, a.drop_tag, e + 3, a.drop_p);
        }
        const float4 t2 = f4add(xr[rt][j], y);
        acc[rt][4 * j] = t2.x; acc[rt][4 * j + 1] = t2.y; acc[rt][4 * j + 2] = t2.z; acc[rt][4 * j + 3] = t2.w;
        s += f4sum(t2);
        if (a.keep_T2 && row < nrows) st4(a.keep_T2, ooff[rt] + 32 * j, t2);
      }
      // LayerNorm statistics: pairwise combination of the eight 16-column pieces of the row (see edge_kernel)
      mean32[rt] = xor32(s) * (1.0f / 32.0f);
      float v2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = acc[rt][i] - mean32[rt];
        v2 = fmaf(d, d, v2);
      }
      m2[rt] = xor32(v2);
    }
    {
      // (the lane id is re-derived HERE, from the hardware (v_mbcnt): the address and the predicate of this store were otherwise
      // computed at the top of the kernel and carried -- at 168 VGPRs, spilled -- across both products)
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      if ((ln >> 5) == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<float2*>(&sRed[(((ln & 31) + 32 * rt) * 4 + wave) * 2]) = make_float2(mean32[rt], m2[rt]);
      }
    }
    __syncthreads();  // statistics complete; every wave is done reading the hidden planes
    STAMP(a.stamps, 5);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = lrow + 32 * rt;
      const float4 sa = *reinterpret_cast<const float4*>(&sRed[row * 8]), sb = *reinterpret_cast<const float4*>(&sRed[row * 8 + 4]);
      const float mean = ((sa.x + sa.z) + (sb.x + sb.z)) * 0.25f;
      const float d0 = sa.x - mean, d1 = sa.z - mean, d2 = sb.x - mean, d3 = sb.z - mean;
      const float var = (((sa.y + sa.w) + (sb.y + sb.w)) + 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3))) * (1.0f / D);
      const float rstd = 1.0f / sqrtf(var + 1e-6f);
      if (!EX && !(var < RANGE_FINITE) && row < nrows) flag_range(a.range_flag, 3, a.layer - 1);  // an operand of the ResidualNorm overflowed fp16
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 g = *reinterpret_cast<const float4*>(&sPar[2 * D + cbase + 8 * j]);
        const float4 be = *reinterpret_cast<const float4*>(&sPar[3 * D + cbase + 8 * j]);
        float4 y;
        float inv;
        inv = rstd * g.x; y.x = fmaf(acc[rt][4 * j], inv, be.x - mean * inv);
        inv = rstd * g.y; y.y = fmaf(acc[rt][4 * j + 1], inv, be.y - mean * inv);
        inv = rstd * g.z; y.z = fmaf(acc[rt][4 * j + 2], inv, be.z - mean * inv);
        inv = rstd * g.w; y.w = fmaf(acc[rt][4 * j + 3], inv, be.w - mean * inv);
        if (row < nrows) st4(a.c, ooff[rt] + 32 * j, y);
        tile_store<EX, TAR>(sTile, row, cbase + 8 * j, y);
      }
    }
    __syncthreads();
    STAMP(a.stamps, 6);
  }

  if (MODE == 0) {  // P1 = c W1 + bg ; P3 = c W3 ; q = c Wq + bq (attention.py:142-151 thirds, :160)
    gemm_tile_x<EX, true, TAR, RT>(sTile, wr, a.WBh, wave, lane, acc);
    STAMP(a.stamps, 7);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bg = *reinterpret_cast<const float4*>(&sPar[4 * D + cbase + 8 * j]);
        if (lrow + 32 * rt < nrows)
          st4(a.oA, ooff[rt] + 32 * j, make_float4(fmaf(acc[rt][4 * j], WINV, bg.x), fmaf(acc[rt][4 * j + 1], WINV, bg.y),
                                                   fmaf(acc[rt][4 * j + 2], WINV, bg.z), fmaf(acc[rt][4 * j + 3], WINV, bg.w)));
      }
    gemm_tile_x<EX, true, TAR, RT>(sTile, wr, a.WCh, wave, lane, acc);
    STAMP(a.stamps, 9);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (lrow + 32 * rt < nrows)
          st4(a.oB, ooff[rt] + 32 * j, make_float4(acc[rt][4 * j] * WINV, acc[rt][4 * j + 1] * WINV, acc[rt][4 * j + 2] * WINV, acc[rt][4 * j + 3] * WINV));
  }
  if (MODE == 0 || MODE == 1) {  // q = c Wq + bq (attention.py:160)
    STAMP(a.stamps, 10);
    gemm_tile_x<EX, false, TAR, RT>(sTile, wr, nullptr, wave, lane, acc);
    STAMP(a.stamps, 11);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bq = *reinterpret_cast<const float4*>(&sPar[5 * D + cbase + 8 * j]);
        if (lrow + 32 * rt < nrows)
          st4(a.oC, ooff[rt] + 32 * j, make_float4(fmaf(acc[rt][4 * j], WINV, bq.x), fmaf(acc[rt][4 * j + 1], WINV, bq.y),
                                                   fmaf(acc[rt][4 * j + 2], WINV, bq.z), fmaf(acc[rt][4 * j + 3], WINV, bq.w)));
      }
    STAMP(a.stamps, 12);
  }
  if (MODE == 2) {  // z = swish(c Wa + ba) (scann_model.py:424); gq = z Wgq + b ; gk = z Wgk + b (attention.py:269-272)
    gemm_tile_x<EX, true, TAR, RT>(sTile, wr, a.WCh, wave, lane, acc);
    __syncthreads();  // every wave is done reading the centre planes
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = lrow + 32 * rt;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bv = *reinterpret_cast<const float4*>(&sPar[4 * D + cbase + 8 * j]);
        const float4 pre = make_float4(fmaf(acc[rt][4 * j], WINV, bv.x), fmaf(acc[rt][4 * j + 1], WINV, bv.y),
                                       fmaf(acc[rt][4 * j + 2], WINV, bv.z), fmaf(acc[rt][4 * j + 3], WINV, bv.w));
        const float4 z = f4swish(pre);
        // no LayerNorm follows this activation: test it directly (its hi part feeds the GlobalAttention projections)
        if (!EX && !(fmaxf(fmaxf(fabsf(z.x), fabsf(z.y)), fmaxf(fabsf(z.z), fabsf(z.w))) < 65504.f) && row < nrows) flag_range(a.range_flag, 4, a.layer);
        if (a.keep_preA && row < nrows) {  // training forward: after_Lc pre-activation and output, kept for the backward
          st4(a.keep_preA, ooff[rt] + 32 * j, pre);
          st4(a.keep_z, ooff[rt] + 32 * j, z);
        }
        tile_store<EX, TAR>(sTile, row, cbase + 8 * j, z);
      }
    }
    __syncthreads();
    gemm_tile_x<EX, true, TAR, RT>(sTile, wr, a.WDh, wave, lane, acc);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bq = *reinterpret_cast<const float4*>(&sPar[5 * D + cbase + 8 * j]);
        if (lrow + 32 * rt < nrows)
          st4(a.oC, ooff[rt] + 32 * j, make_float4(fmaf(acc[rt][4 * j], WINV, bq.x), fmaf(acc[rt][4 * j + 1], WINV, bq.y),
                                                   fmaf(acc[rt][4 * j + 2], WINV, bq.z), fmaf(acc[rt][4 * j + 3], WINV, bq.w)));
      }
    gemm_tile_x<EX, false, TAR, RT>(sTile, wr, nullptr, wave, lane, acc);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bk = *reinterpret_cast<const float4*>(&sPar[6 * D + cbase + 8 * j]);
        if (lrow + 32 * rt < nrows)
          st4(a.oB, ooff[rt] + 32 * j, make_float4(fmaf(acc[rt][4 * j], WINV, bk.x), fmaf(acc[rt][4 * j + 1], WINV, bk.y),
                                                   fmaf(acc[rt][4 * j + 2], WINV, bk.z), fmaf(acc[rt][4 * j + 3], WINV, bk.w)));
      }
  }
}

void launch_atom(const AtomArgs& a, hipStream_t s) {
  if (a.n_atom <= 0) return;
  // 32-row tiles (<= 128 VGPRs, 22 KB of LDS: four workgroups per CU = 1,024 slots) while they all fit ONE round of workgroups: the
  // launch is then the latency chain of a tile, and a 32-row tile's chain is shorter (one batch per launch: 490 k -> 568 k
  // molecules/s, training step 1.14 -> 1.09 ms).  Beyond that 64-row tiles (half the weight traffic per row).
  const int rows = a.row_tab ? TA : a.n_atom <= 32 * 1024 ? 32 : 64;
  if (a.row_tab && a.n_row_tab <= 0) return;
  const dim3 grid(a.row_tab ? a.n_row_tab : (a.n_atom + rows - 1) / rows), block(256);
#define SCANN_ATOM_CASE(F, M)                                                                  \
  do {                                                                                         \
    if (a.exact) {                                                                             \
      if (rows == 32) hipLaunchKernelGGL((atom_kernel<F, M, 1, true>), grid, block, 0, s, a);  \
      else hipLaunchKernelGGL((atom_kernel<F, M, 2, true>), grid, block, 0, s, a);             \
    } else if (rows == 32) hipLaunchKernelGGL((atom_kernel<F, M, 1>), grid, block, 0, s, a);   \
    else hipLaunchKernelGGL((atom_kernel<F, M, 2>), grid, block, 0, s, a);                     \
  } while (0)
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
//
// One workgroup (4 waves) per tile of <= 64 edges of <= TQ whole atoms; three workgroups per CU (<= 53.3 KB of LDS,
// <= 168 VGPRs).  LocalAttention.call for those edges (attention.py:136-216), fused:
//
//   GEMM 1   U = G . W2 (g_update; base branch: gd . Wf, K = 20 padded to 32)                    split-fp16 MFMA
//   epilogue IN THE ACCUMULATOR LAYOUT (lane = edge row, 16 columns): V = U + P1[i] + P3[j], T = swish(V) + G,
//            LayerNorm_g statistics (lane pair -> LDS -> the row's four waves), geom' -> HBM, ang = c[j] * geom' -> LDS
//   GEMM 2   K = ang . Wk + bk                                                                    split-fp16 MFMA
//   epilogue logits q[i].K per (edge, head) straight from the accumulators, K -> LDS
//   softmax over each atom's edges (online), context + unscaled-query residual, LayerNorm -> HBM
//
// Split-fp16 projections: every fp32 operand x is carried as two fp16 numbers, hi = fp16(x) and lo = fp16(x - hi)
// (22 significant bits; fp16 subnormals are honoured by the matrix pipe, tools/mfma_f16_probe.hip), and a product of two
// such operands is three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator: lo.hi + hi.lo + hi.hi (the lo.lo term is below
// fp32 resolution).  Measured against fp64 the result is as close as the exact-fp32 MFMA chain (1.5e-7 of sum |a b| for
// K = 128; profiles/r02_notes.md) at 3/16 of its matrix-pipe time: f16 MFMA runs at 16x the f32 rate.  Weights are split
// once at load time (pack_weight_f16: pre-scaled by 2^8 so that their lo parts stay normal numbers; the exact inverse is
// applied to the accumulators), activations when a tile is written to LDS -- the tile buffer holds a hi plane and a lo plane
// of 64 x 128 fp16 instead of 64 x 128 fp32: same bytes, same b128 fragment reads.
// Range: |activation| and |weight * 256| must stay below 65504 (scann_load_weights refuses larger weights; activations here
// are LayerNorm / swish outputs of O(1..10)).

//
// FB (g_update, 64-row tiles, first layer of an inference forward): the geometry rows come out of basis_kernel's arithmetic, done
// here on the tile's rows (Gaussian expansions -> planes -> two K = 20 products -> bias, swish, product), not out of memory.
// EX: exact-fp32 projections for the 128x128 kernels (scann_mma.h; the K = 20 filters keep the split form: their inputs are Gaussians
// in [0, 1]) -- the fallback run_forward takes when the split-fp16 range guard fired.  Never with FB.
template <bool GUPD, int RT, bool FB = false, bool EX = false>
__global__ __launch_bounds__(256, RT == 2 ? 3 : 4) void edge_kernel(EdgeArgs a) {
#pragma clang fp contract(off)  // fusions are written out (fmaf): both unrolled row-tile copies of a formula must round alike,
                                // so that a row's result does not depend on where in a tile it lands (batch-composition invariance)
  static_assert(!FB || GUPD, "the fused basis exists for the g_update kernel only");
  static_assert(!(FB && EX), "the exact fallback runs the plain first layer");
  constexpr int TEK = 32 * RT;  // edge rows per tile: 64 (three workgroups per CU) or, for launches of one round, 32 (four)
  // hi / lo planes of the A operand (G or the basis, then ang): 2 x 64 x 272 B; afterwards K as fp32 [64][LDS_STRIDE]
  __shared__ __attribute__((aligned(16))) unsigned char sTile[2 * TEK * PLANE_STRIDE * 2];
  __shared__ __attribute__((aligned(16))) float sQ[TQ * LDS_STRIDE];  // P1 rows, then query rows of the tile's atoms, then context
  __shared__ __attribute__((aligned(16))) float sE[TEK * NHEAD];      // LayerNorm_g partial statistics [row][wave][2], then logits
  __shared__ __attribute__((aligned(16))) float sPar[5 * D];          // layer_norm_g gamma/beta (base: filter bias), layer_norm gamma/beta, key bias
  __shared__ int sOff[TQ + 1];
#ifdef SCANN_DIAG_OCC2  // diagnostic: two workgroups per CU instead of three (how much of the time is latency hidden by occupancy?)
  __shared__ float sDummy[7 * 1024];
  if (a.n_tile < 0) sDummy[threadIdx.x] = 1.f;
#endif
  static_assert(sizeof(sTile) >= TEK * LDS_STRIDE * sizeof(float), "K tile must fit the plane buffer");
  _Float16* const sH = reinterpret_cast<_Float16*>(sTile);
  _Float16* const sL = sH + TEK * PLANE_STRIDE;
  float* const sK = reinterpret_cast<float*>(sTile);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tix = a.xcd_remap ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
  const EdgeTile tile = a.tiles[tix];
  const int part = a.tile_part ? a.tile_part[tix] : -1;  // >= 0: one <= 64-edge chunk of an atom with more than 64 neighbours
  const int eb = tile.edge_begin, ne = tile.edge_end - eb, natom = tile.atom_end - tile.atom_begin;
  const int nem1 = ne > 0 ? ne - 1 : 0;
  const int lrow = lane & 31, lh = lane >> 5;  // accumulator layout: this lane's rows lrow, lrow + 32; column runs 32 wave + 8 j + 4 lh
  const int cbase = 32 * wave + 4 * lh;

  STAMP(a.stamps, 0);
  STAMP_REAL(a.stamps, 12);  // 100 MHz reference clock at entry and exit: shader clock = cycles / ticks * 100 MHz
  // ---- prologue: every load is issued UNCONDITIONALLY from clamped rows and selected afterwards (a load under a per-thread
  // guard compiles to branch + load + s_waitcnt vmcnt(0): one full memory round trip per guard) -------------------------------
  // one weight slab at a time, in two halves of 4 k-steps (A: k < 64, B: k >= 64): W2 (base branch: Wf, 2 k-steps in A), later Wk
  WRegs<EX> wr;
  f16x8 fth[2], ftl[2];  // base branch: the K = 20 filter (split form in every instantiation)
  if (GUPD) {
    if (!FB) {  // (FB: requested after the basis products, whose operands and accumulators need the registers first)
      load_whalf<EX>(wr, a.p.W2h, wave, lane, 0);
      load_whalf<EX>(wr, a.p.W2h, wave, lane, 1);
    }
  } else {
    load_wsplit<2>(a.p.Wfh, wave, lane, fth, ftl);
  }
  unsigned nboff[RT];  // byte offset of (neighbour atom row, this lane's first column) in an [n_atom,128] tensor
  int ctr[RT];         // tile-local centre atom of this lane's two edge rows
  float ewgt[RT] = {};
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int e = ne > 0 ? eb + min(lrow + 32 * rt, nem1) : 0;
    int nb = ne > 0 ? a.edge_col[e] : 0;
    if (FB && a.species) nb = a.species[nb];  // row of the per-species tables
    nboff[rt] = ((unsigned)nb * D + cbase) * 4;
    ctr[rt] = ne > 0 ? a.edge_row[e] - tile.atom_begin : 0;
    if (!GUPD) ewgt[rt] = ne > 0 ? a.edge_weight[e] : 0.f;
  }
  const int voff = a.edge_offset[tile.atom_begin + min(tid, natom)];
  const float bkc = a.p.bk[tid & (D - 1)];
  const float par0 = GUPD ? (tid < D ? a.p.lng_g : a.p.lng_b)[tid & (D - 1)] : a.p.bfg[tid & (D - 1)];
  const float par1 = (tid < D ? a.p.ln_g : a.p.ln_b)[tid & (D - 1)];
  float4 greg[RT][4];
  {
    const int r = tid >> 2, sub = tid & 3;  // staging map of the base branch: 4 threads per edge row
    const int rs = r < ne ? r : nem1;
    if (GUPD) {
      float4 p1reg[3];  // centre thirds P1 = c_i W1 + bg of the tile's atoms
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int idx = tid + 256 * i, la = min(idx >> 5, natom - 1), c4 = idx & 31;
        const int arow = FB && a.species ? a.species[tile.atom_begin + la] : tile.atom_begin + la;
        p1reg[i] = ld4(a.P1, ((unsigned)arow * 32 + c4) * 16);
      }
      // geometry rows G of the tile's edges in the ACCUMULATOR layout, straight into registers: they stay there, exact fp32,
      // for the residual (attention.py:153); a tile without edges reads (and ignores) a valid row
      if (FB) {
        // geom0 rows of the tile = swish(Gauss(dist) Wd + bd) * swish(Gauss(weight) Ww + bw) (scann_model.py:378-389): basis_kernel's
        // instruction sequence on this tile's rows (a row's result does not depend on where in a tile it sits), the basis planes
        // staged in the plane buffer the geometry rows take over afterwards
        _Float16* const bH = reinterpret_cast<_Float16*>(sTile);
        _Float16* const bL = bH + TEK * BASIS_STRIDE;
        f16x8 bdh[2], bdl[2], bwh[2], bwl[2];
        load_wsplit<2>(a.basis.Wdh, wave, lane, bdh, bdl);
        load_wsplit<2>(a.basis.Wwh, wave, lane, bwh, bwl);
        {
          const float xd = ne > 0 ? a.dist[eb + rs] : 0.f, xw = ne > 0 ? a.edge_weight[eb + rs] : 0.f;
          // the 20 + 20 Gaussian centres: ONE coalesced load per wave and table, handed to the lanes by ds_bpermute (the sixteen
          // per-lane table loads of basis_kernel were a fifth of this kernel's vector-memory instructions); the same values
          const float cd_l = a.basis.cd[min(lane, NG - 1)], cw_l = a.basis.cw[min(lane, NG - 1)];
          f16x8 gh[2], gl[2];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = 8 * sub + i;
            const float cdk = __shfl(cd_l, k), cwk = __shfl(cw_l, k);
            const float vd = (k < NG && r < ne) ? gauss_fast(xd, cdk) : 0.f, vw = (k < NG && r < ne) ? gauss_fast(xw, cwk) : 0.f;
            gh[0][i] = (_Float16)vd; gl[0][i] = (_Float16)(vd - (float)gh[0][i]);
            gh[1][i] = (_Float16)vw; gl[1][i] = (_Float16)(vw - (float)gh[1][i]);
          }
          if (RT == 2 || r < TEK) {  // (32-row tiles: the upper half of the staging threads has no row)
            *reinterpret_cast<f16x8*>(bH + r * BASIS_STRIDE + 8 * sub) = gh[0];
            *reinterpret_cast<f16x8*>(bL + r * BASIS_STRIDE + 8 * sub) = gl[0];
            *reinterpret_cast<f16x8*>(bH + r * BASIS_STRIDE + 32 + 8 * sub) = gh[1];
            *reinterpret_cast<f16x8*>(bL + r * BASIS_STRIDE + 32 + 8 * sub) = gl[1];
          }
        }
        // the P1 rows (requested before the Gaussians) go to their LDS rows NOW: twelve registers less across the basis products
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int idx = tid + 256 * i;
          *reinterpret_cast<float4*>(&sQ[(idx >> 5) * LDS_STRIDE + 4 * (idx & 31)]) = p1reg[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        f32x16 accd[RT], accw[RT];
        mma_split<2, true, BASIS_STRIDE, RT>(bH, bL, bdh, bdl, lane, accd);
        mma_split<2, true, BASIS_STRIDE, RT>(bH + 32, bL + 32, bwh, bwl, lane, accw);
        __builtin_amdgcn_sched_barrier(0);
        // W2 in halves, each requested once a row tile's accumulators are free (all of it at once: 28 B of scratch per lane)
        constexpr float WINV0 = 1.0f / WSCALE;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 bd = *reinterpret_cast<const float4*>(a.basis.bd + cbase + 8 * j);
            const float4 bw = *reinterpret_cast<const float4*>(a.basis.bw + cbase + 8 * j);
            const float4 sd = f4swish(make_float4(fmaf(accd[rt][4 * j], WINV0, bd.x), fmaf(accd[rt][4 * j + 1], WINV0, bd.y),
                                                  fmaf(accd[rt][4 * j + 2], WINV0, bd.z), fmaf(accd[rt][4 * j + 3], WINV0, bd.w)));
            const float4 sw = f4swish(make_float4(fmaf(accw[rt][4 * j], WINV0, bw.x), fmaf(accw[rt][4 * j + 1], WINV0, bw.y),
                                                  fmaf(accw[rt][4 * j + 2], WINV0, bw.z), fmaf(accw[rt][4 * j + 3], WINV0, bw.w)));
            greg[rt][j] = f4mul(sd, sw);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (rt == 0) load_whalf<EX>(wr, a.p.W2h, wave, lane, 0);
          if (rt == RT - 1) load_whalf<EX>(wr, a.p.W2h, wave, lane, 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // every wave is done reading the basis planes: the geometry planes may overwrite them
      } else {
        const float* gsrc = ne > 0 ? a.geom : a.P1;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const unsigned goff = ((ne > 0 ? (unsigned)(eb + min(lrow + 32 * rt, nem1)) : (unsigned)tile.atom_begin) * D + cbase) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) greg[rt][j] = ld4(gsrc, goff + 32 * j);
        }
      }
      if (!FB) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int idx = tid + 256 * i;
          *reinterpret_cast<float4*>(&sQ[(idx >> 5) * LDS_STRIDE + 4 * (idx & 31)]) = p1reg[i];  // rows >= natom: unused copies
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = lrow + 32 * rt;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (row >= ne) greg[rt][j] = make_float4(0.f, 0.f, 0.f, 0.f);
          tile_store<EX, TEK>(sTile, row, cbase + 8 * j, greg[rt][j]);
        }
      }
    } else {
      // base SCANN (attention.py:155): the raw distance basis gd[e][0..20) of the tile's edges, zero-padded to K = 32
      float4 gv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = sub + 4 * i;  // float4 piece 0..7 of the 32-wide row; pieces 5..7 are padding
        gv[i] = ne > 0 ? ld4(a.gd, (unsigned)(eb + rs) * (NG * 4) + min(c4, 4) * 16) : ld4(a.q, (unsigned)tile.atom_begin * (D * 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = sub + 4 * i;
        if (r >= ne || c4 > 4) gv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        f16x4 h, l;
        split4(gv[i], h, l);
        if (r < TEK) {  // (32-row tiles: the upper half of the staging threads has no row)
          *reinterpret_cast<f16x4*>(sH + r * PLANE_STRIDE + 4 * c4) = h;
          *reinterpret_cast<f16x4*>(sL + r * PLANE_STRIDE + 4 * c4) = l;
        }
      }
    }
  }
  if (tid <= natom) sOff[tid] = part >= 0 ? (tid == 0 ? 0 : ne) : voff - eb;  // a chunk tile holds edges [0, ne) of its single atom
  sPar[tid] = par0;            // g_update: gamma | beta of layer_norm_g; base: filter bias (both halves)
  sPar[2 * D + tid] = par1;    // gamma | beta of layer_norm
  if (tid < D) sPar[4 * D + tid] = bkc;
  __syncthreads();
  STAMP(a.stamps, 1);

  float4 p3r[RT][4];
  f32x16 acc[RT];
  if (GUPD) {
    mma_half<EX, TEK, RT>(sTile, wr, lane, acc, 0);
    __builtin_amdgcn_sched_barrier(0);
    // gathered neighbour thirds P3[j] = c_j W3 (into the registers the first weight half leaves): in flight over the second half
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) p3r[rt][j] = ld4(a.P3, nboff[rt] + 32 * j);
    __builtin_amdgcn_sched_barrier(0);
    mma_half<EX, TEK, RT>(sTile, wr, lane, acc, 1);
  } else {
    mma_split<2, true, PLANE_STRIDE, RT>(sH, sL, fth, ftl, lane, acc);
  }
  STAMP(a.stamps, 2);
  __builtin_amdgcn_sched_barrier(0);

  constexpr float WINV = EX ? 1.0f : 1.0f / WSCALE;     // the 128x128 kernels (exact images are unscaled)
  constexpr float WINVF = 1.0f / WSCALE;                // the base branch's K = 20 filter: always the split form
  const int qa = tid >> 5;  // query rows qa, qa + 8, qa + 16 of the tile go through this thread
  const unsigned qoff = (tid & 31) * 16;
  // byte offset of the row of tile-local atom la (clamped by the caller) in an [n_atom,128] tensor / the per-species table
  const auto arow_off = [&](int la) __attribute__((always_inline)) {
    return qoff + (unsigned)(FB && a.species ? a.species[tile.atom_begin + la] : tile.atom_begin + la) * (D * 4);
  };
  float4 q0, q1, q2;
  float4 cn[RT][4];
  unsigned eoff[RT];  // (edge row, first column) bytes
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) eoff[rt] = ((unsigned)(eb + lrow + 32 * rt) * D + cbase) * 4;
  float* const gout = a.geom_out ? a.geom_out : a.geom;
  if (GUPD) {
    // geometry update (attention.py:141-153) on the accumulators: T = swish(U + P1[i] + P3[j]) + G
    float mean32[RT], m2[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = lrow + 32 * rt;
      const float* p1 = sQ + ctr[rt] * LDS_STRIDE + cbase;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 p1v = *reinterpret_cast<const float4*>(p1 + 8 * j);
        const float4 g = greg[rt][j];
        float4 v;
        v.x = fmaf(acc[rt][4 * j], WINV, p1v.x) + p3r[rt][j].x;
        v.y = fmaf(acc[rt][4 * j + 1], WINV, p1v.y) + p3r[rt][j].y;
        v.z = fmaf(acc[rt][4 * j + 2], WINV, p1v.z) + p3r[rt][j].z;
        v.w = fmaf(acc[rt][4 * j + 3], WINV, p1v.w) + p3r[rt][j].w;
        const float4 t = f4swish_plus(v, g);
        acc[rt][4 * j] = t.x; acc[rt][4 * j + 1] = t.y; acc[rt][4 * j + 2] = t.z; acc[rt][4 * j + 3] = t.w;
        s += f4sum(t);
        if (a.keep_V && row < ne) {  // training forward: the backward reads these instead of recomputing them
          st4(a.keep_V, eoff[rt] + 32 * j, v);
          st4(a.keep_T, eoff[rt] + 32 * j, t);
        }
      }
      // LayerNorm_g statistics of the row: its 128 columns sit in 2 lanes x 4 waves.  Pairwise (Chan) combination of
      // (mean, sum of squared deviations) of the eight 16-column pieces: as accurate as the two-pass form.
      mean32[rt] = xor32(s) * (1.0f / 32.0f);
      float v2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = acc[rt][i] - mean32[rt];
        v2 = fmaf(d, d, v2);
      }
      m2[rt] = xor32(v2);
      __builtin_amdgcn_sched_barrier(0);  // one row tile at a time: hoisting both tiles' LDS reads costs 40 VGPRs (spills)
    }
    if (lh == 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<float2*>(&sE[((lrow + 32 * rt) * 4 + wave) * 2]) = make_float2(mean32[rt], m2[rt]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the P3 registers are free: neighbour centre rows c[j] (attention.py:136) and the key weights land over the barrier
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) cn[rt][j] = ld4(a.c, nboff[rt] + 32 * j);
    load_whalf<EX>(wr, a.p.Wkh, wave, lane, 0);  // first half of the key weights; the second half is requested at the GEMM
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // statistics complete; every wave is done with the G planes and the P1 rows
    STAMP(a.stamps, 3);
    // query rows of the tile's atoms: requested now, parked in sQ (the P1 rows are dead) before the key GEMM
    q0 = ld4(a.q, arow_off(min(qa, natom - 1)));
    q1 = ld4(a.q, arow_off(min(qa + 8, natom - 1)));
    q2 = ld4(a.q, arow_off(min(qa + 16, natom - 1)));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = lrow + 32 * rt;
      const float4 sa = *reinterpret_cast<const float4*>(&sE[row * 8]), sb = *reinterpret_cast<const float4*>(&sE[row * 8 + 4]);
      const float mean = ((sa.x + sa.z) + (sb.x + sb.z)) * 0.25f;
      const float d0 = sa.x - mean, d1 = sa.z - mean, d2 = sb.x - mean, d3 = sb.z - mean;
      const float var = (((sa.y + sa.w) + (sb.y + sb.w)) + 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3))) * (1.0f / D);
      const float rstd = 1.0f / sqrtf(var + 1e-6f);
      if (!EX && !(var < RANGE_FINITE) && row < ne) flag_range(a.range_flag, 1, a.layer);  // an operand of the geometry update overflowed fp16
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 g = *reinterpret_cast<const float4*>(&sPar[cbase + 8 * j]);
        const float4 be = *reinterpret_cast<const float4*>(&sPar[D + cbase + 8 * j]);
        float4 y;
        float inv;
        inv = rstd * g.x; y.x = fmaf(acc[rt][4 * j], inv, be.x - mean * inv);
        inv = rstd * g.y; y.y = fmaf(acc[rt][4 * j + 1], inv, be.y - mean * inv);
        inv = rstd * g.z; y.z = fmaf(acc[rt][4 * j + 2], inv, be.z - mean * inv);
        inv = rstd * g.w; y.w = fmaf(acc[rt][4 * j + 3], inv, be.w - mean * inv);
        float4 ang = f4mul(cn[rt][j], y);  // attention.py:157
        if (row < ne) {
          if (!a.geom_dead) st4(gout, eoff[rt] + 32 * j, y);  // threaded to the next layer (scann_model.py:415)
          if (a.keep_ang) st4(a.keep_ang, eoff[rt] + 32 * j, ang);
        } else {
          ang = make_float4(0.f, 0.f, 0.f, 0.f);  // ragged tail rows stay defined (and zero) for the MFMA
        }
        tile_store<EX, TEK>(sTile, row, cbase + 8 * j, ang);
      }
      __builtin_amdgcn_sched_barrier(0);  // one row tile at a time (register pressure)
    }
  } else {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) cn[rt][j] = ld4(a.c, nboff[rt] + 32 * j);
    load_whalf<EX>(wr, a.p.Wkh, wave, lane, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // every wave is done reading the basis planes
    STAMP(a.stamps, 3);
    // query rows of the tile's atoms: requested now, parked in sQ (the P1 rows are dead) before the key GEMM
    q0 = ld4(a.q, arow_off(min(qa, natom - 1)));
    q1 = ld4(a.q, arow_off(min(qa + 8, natom - 1)));
    q2 = ld4(a.q, arow_off(min(qa + 16, natom - 1)));
    __builtin_amdgcn_sched_barrier(0);
    // base SCANN: geomL = swish(gd Wf + bf) * weight (attention.py:155); ang = c[j] * geomL
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = lrow + 32 * rt;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bf = *reinterpret_cast<const float4*>(&sPar[cbase + 8 * j]);
        float4 y = f4swish(make_float4(fmaf(acc[rt][4 * j], WINVF, bf.x), fmaf(acc[rt][4 * j + 1], WINVF, bf.y),
                                       fmaf(acc[rt][4 * j + 2], WINVF, bf.z), fmaf(acc[rt][4 * j + 3], WINVF, bf.w)));
        y.x *= ewgt[rt]; y.y *= ewgt[rt]; y.z *= ewgt[rt]; y.w *= ewgt[rt];
        float4 ang = f4mul(cn[rt][j], y);
        if (row >= ne) ang = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.keep_ang && row < ne) {  // training forward: geomL (in the V slot) and the gated rows, kept for the backward
          st4(a.keep_V, eoff[rt] + 32 * j, y);
          st4(a.keep_ang, eoff[rt] + 32 * j, ang);
        }
        tile_store<EX, TEK>(sTile, row, cbase + 8 * j, ang);
      }
    }
  }
  // query rows: the P1 rows are dead (barrier above)
  *reinterpret_cast<float4*>(&sQ[qa * LDS_STRIDE + 4 * (tid & 31)]) = q0;
  *reinterpret_cast<float4*>(&sQ[(qa + 8) * LDS_STRIDE + 4 * (tid & 31)]) = q1;
  *reinterpret_cast<float4*>(&sQ[(qa + 16) * LDS_STRIDE + 4 * (tid & 31)]) = q2;
  __syncthreads();  // ang planes and query rows complete
  STAMP(a.stamps, 4);

  // K = ang . Wk + bk (attention.py:163); the second half of Wk arrives under the first half's MFMAs
  load_whalf<EX>(wr, a.p.Wkh, wave, lane, 1);
  __builtin_amdgcn_sched_barrier(0);
  mma_half<EX, TEK, RT>(sTile, wr, lane, acc, 0);
  mma_half<EX, TEK, RT>(sTile, wr, lane, acc, 1);
  STAMP(a.stamps, 5);
  // logits e[n, h] = (q[i, h, :] * 16^-0.5) . K[n, h, :] (attention.py:180-183) from the accumulators: this lane holds 8 of the
  // 16 columns of heads 2 wave (j = 0, 1) and 2 wave + 1 (j = 2, 3) of its rows; its partner lane (^32) holds the other 8
  float lg[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const float* qrow = sQ + ctr[rt] * LDS_STRIDE + cbase;
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
      float e = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * hp + jj;
        const float4 bk = *reinterpret_cast<const float4*>(&sPar[4 * D + cbase + 8 * j]);
        const float4 q4 = *reinterpret_cast<const float4*>(qrow + 8 * j);
        const float k0 = fmaf(acc[rt][4 * j], WINV, bk.x), k1 = fmaf(acc[rt][4 * j + 1], WINV, bk.y);
        const float k2 = fmaf(acc[rt][4 * j + 2], WINV, bk.z), k3 = fmaf(acc[rt][4 * j + 3], WINV, bk.w);
        acc[rt][4 * j] = k0; acc[rt][4 * j + 1] = k1; acc[rt][4 * j + 2] = k2; acc[rt][4 * j + 3] = k3;
        e = fmaf(q4.x * 0.25f, k0, e); e = fmaf(q4.y * 0.25f, k1, e); e = fmaf(q4.z * 0.25f, k2, e); e = fmaf(q4.w * 0.25f, k3, e);
      }
      lg[rt][hp] = xor32(e);
    }
  }
  __syncthreads();  // every wave is done reading the ang planes: K may overwrite them
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = lrow + 32 * rt;
    // (two predicated stores, not `lh ? lg[rt][1] : lg[rt][0]`: hipcc makes a private array of lg for the select in the base kernel)
    if (lh == 0) sE[row * NHEAD + 2 * wave] = lg[rt][0];
    else sE[row * NHEAD + 2 * wave + 1] = lg[rt][1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 k4 = make_float4(acc[rt][4 * j], acc[rt][4 * j + 1], acc[rt][4 * j + 2], acc[rt][4 * j + 3]);
      *reinterpret_cast<float4*>(&sK[row * LDS_STRIDE + cbase + 8 * j]) = k4;
      if (a.keep_K && row < ne) st4(a.keep_K, eoff[rt] + 32 * j, k4);
    }
  }
  __syncthreads();
  STAMP(a.stamps, 8);
  // softmax over each atom's edges + context + unscaled-query residual (attention.py:186-212).
  // Packed edges are all unmasked: the additive -1e9 and the multiplicative mask are the identity; an atom without edges
  // yields q (then LayerNorm), which is what the reference's fully-masked row gives.
  {
    const int lgp = tid >> 5, c4 = tid & 31, h = c4 >> 2;
    for (int la = lgp; la < natom; la += 8) {
      const int e0 = sOff[la], e1 = sOff[la + 1];
      // row maximum first (logits only), then one exp per edge: no running-maximum rescaling in the accumulation loop
      float m = -INFINITY;
      for (int n = e0; n < e1; n += 4) {
        const float v0 = sE[n * NHEAD + h], v1 = sE[min(n + 1, e1 - 1) * NHEAD + h];
        const float v2 = sE[min(n + 2, e1 - 1) * NHEAD + h], v3 = sE[min(n + 3, e1 - 1) * NHEAD + h];
        m = fmaxf(fmaxf(m, fmaxf(v0, v1)), fmaxf(v2, v3));
      }
      float ssum = 0.f;
      float4 cx = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int n = e0; n < e1; n += 2) {
        const bool two = n + 1 < e1;
        const int n1 = two ? n + 1 : n;
        const float4 ka = *reinterpret_cast<const float4*>(&sK[n * LDS_STRIDE + 4 * c4]);
        const float4 kb = *reinterpret_cast<const float4*>(&sK[n1 * LDS_STRIDE + 4 * c4]);
        float pa = fast_exp(sE[n * NHEAD + h] - m), pb = two ? fast_exp(sE[n1 * NHEAD + h] - m) : 0.f;
        ssum += pa + pb;
        if (a.attn_drop_p > 0.f) {  // training with use_drop: Dropout(0.05) on the attention weights (attention.py:116,191)
          pa *= drop_scale(a.attn_drop_seed, a.attn_drop_tag, (size_t)(eb + n) * NHEAD + h, a.attn_drop_p);
          pb *= drop_scale(a.attn_drop_seed, a.attn_drop_tag, (size_t)(eb + n1) * NHEAD + h, a.attn_drop_p);
        }
        cx.x = fmaf(pa, ka.x, fmaf(pb, kb.x, cx.x));
        cx.y = fmaf(pa, ka.y, fmaf(pb, kb.y, cx.y));
        cx.z = fmaf(pa, ka.z, fmaf(pb, kb.z, cx.z));
        cx.w = fmaf(pa, ka.w, fmaf(pb, kb.w, cx.w));
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
        *qp = make_float4(fmaf(cx.x, rs, q4.x), fmaf(cx.y, rs, q4.y), fmaf(cx.z, rs, q4.z), fmaf(cx.w, rs, q4.w));
      }
    }
  }
  __syncthreads();
  STAMP(a.stamps, 9);
  // LayerNorm of the context rows (attention.py:214): 8 threads per atom row
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
      if (!EX && !(v < RANGE_FINITE)) flag_range(a.range_flag, 2, a.layer);  // the gated rows or the keys overflowed fp16
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
        st4(a.ctx, ((unsigned)(tile.atom_begin + rr) * 32 + c4) * 16, y);
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
                                                         const float* __restrict__ ln_b, float* __restrict__ ctx,
                                                         int32_t* __restrict__ range_flag, int layer) {
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
  if (!(sRed[1][0] + sRed[1][1] < RANGE_FINITE) && c == 0) flag_range(range_flag, 2, layer);
  const float rstd = 1.0f / sqrtf((sRed[1][0] + sRed[1][1]) * (1.0f / D) + 1e-6f);
  const float inv = rstd * ln_g[c];
  ctx[(size_t)atom * D + c] = t * inv + (ln_b[c] - mean * inv);
}

void launch_edge_merge(const int32_t* big_tab, int n_big, const float* part_buf, const float* q, const float* ln_g,
                       const float* ln_b, float* ctx, int32_t* range_flag, int layer, hipStream_t s) {
  if (n_big > 0)
    hipLaunchKernelGGL(edge_merge_kernel, dim3(n_big), dim3(128), 0, s, big_tab, part_buf, q, ln_g, ln_b, ctx, range_flag, layer);
}

// centre atom of every edge: one thread per atom writes its CSR row's entries (a batch has ~8 edges per atom)
__global__ void edge_row_kernel(const int32_t* __restrict__ edge_offset, int n_atom, int32_t* __restrict__ edge_row) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_atom) return;
  for (int e = edge_offset[a]; e < edge_offset[a + 1]; ++e) edge_row[e] = a;
}
void launch_edge_row(const int32_t* edge_offset, int n_atom, int32_t* edge_row, hipStream_t s) {
  if (n_atom > 0) hipLaunchKernelGGL(edge_row_kernel, dim3((n_atom + 255) / 256), dim3(256), 0, s, edge_offset, n_atom, edge_row);
}

void launch_edge(const EdgeArgs& a, hipStream_t s) {
  if (a.n_tile <= 0) return;
  const dim3 grid(a.n_tile), block(256);
  // tile_rows is the height the batch's tile plan was made for (scann_batch_upload: 32 for launches of one round of workgroups)
  if (a.exact) {  // exact-fp32 fallback: never with the fused first layer
    if (a.tile_rows == 32) {
      if (a.g_update) hipLaunchKernelGGL((edge_kernel<true, 1, false, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((edge_kernel<false, 1, false, true>), grid, block, 0, s, a);
    } else {
      if (a.g_update) hipLaunchKernelGGL((edge_kernel<true, 2, false, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((edge_kernel<false, 2, false, true>), grid, block, 0, s, a);
    }
  } else if (a.fuse_basis && a.g_update) {
    if (a.tile_rows == 32) hipLaunchKernelGGL((edge_kernel<true, 1, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((edge_kernel<true, 2, true>), grid, block, 0, s, a);
  } else if (a.tile_rows == 32) {
    if (a.g_update) hipLaunchKernelGGL((edge_kernel<true, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((edge_kernel<false, 1>), grid, block, 0, s, a);
  } else {
    if (a.g_update) hipLaunchKernelGGL((edge_kernel<true, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((edge_kernel<false, 2>), grid, block, 0, s, a);
  }
}

// ---- basis kernel ----------------------------------------------------------------------------------

// geom0 = swish(G(dist) Wd + bd) * swish(G(weight) Ww + bw) for a tile of 64 edges: the two K = 20 products run on the matrix
// pipe (split-fp16, K padded to 32; the VALU form of round 1 issued 2,600 instructions per wave and was VALU-bound at 92 us
// per 16-batch launch), bias / swish / product in the accumulator layout, rows stored in 16-byte pieces.
__global__ __launch_bounds__(256) void basis_kernel(BasisParams p, const float* __restrict__ dist,
                                                    const float* __restrict__ weight, int n_edge,
                                                    float* __restrict__ geom) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) _Float16 sH[64 * BASIS_STRIDE];  // per row: distance basis [0,32) | angle basis [32,64), hi parts
  __shared__ __attribute__((aligned(16))) _Float16 sL[64 * BASIS_STRIDE];  // lo parts
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5, cbase = 32 * wave + 4 * lh;
  const int e0 = blockIdx.x * 64;
  const int ne = min(64, n_edge - e0);
  f16x8 bdh[2], bdl[2], bwh[2], bwl[2];
  load_wsplit<2>(p.Wdh, wave, lane, bdh, bdl);
  load_wsplit<2>(p.Wwh, wave, lane, bwh, bwl);
  float4 bd[4], bw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bd[j] = *reinterpret_cast<const float4*>(p.bd + cbase + 8 * j);
    bw[j] = *reinterpret_cast<const float4*>(p.bw + cbase + 8 * j);
  }
  {
    const int r = tid >> 2, sub = tid & 3;  // 4 threads per edge row, 8 basis columns each
    const int rs = min(r, ne - 1);
    const float xd = dist[e0 + rs], xw = weight[e0 + rs];
    f16x8 gh[2], gl[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 8 * sub + i;
      const float cdk = p.cd[min(k, NG - 1)], cwk = p.cw[min(k, NG - 1)];
      const float vd = (k < NG && r < ne) ? gauss_fast(xd, cdk) : 0.f, vw = (k < NG && r < ne) ? gauss_fast(xw, cwk) : 0.f;
      gh[0][i] = (_Float16)vd; gl[0][i] = (_Float16)(vd - (float)gh[0][i]);
      gh[1][i] = (_Float16)vw; gl[1][i] = (_Float16)(vw - (float)gh[1][i]);
    }
    *reinterpret_cast<f16x8*>(sH + r * BASIS_STRIDE + 8 * sub) = gh[0];
    *reinterpret_cast<f16x8*>(sL + r * BASIS_STRIDE + 8 * sub) = gl[0];
    *reinterpret_cast<f16x8*>(sH + r * BASIS_STRIDE + 32 + 8 * sub) = gh[1];
    *reinterpret_cast<f16x8*>(sL + r * BASIS_STRIDE + 32 + 8 * sub) = gl[1];
  }
  __syncthreads();
  f32x16 accd[2], accw[2];
  mma_split<2, true, BASIS_STRIDE>(sH, sL, bdh, bdl, lane, accd);
  mma_split<2, true, BASIS_STRIDE>(sH + 32, sL + 32, bwh, bwl, lane, accw);
  constexpr float WINV = 1.0f / WSCALE;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int row = lrow + 32 * rt;
    if (row < ne) {
      const unsigned off = ((unsigned)(e0 + row) * D + cbase) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 sd = f4swish(make_float4(fmaf(accd[rt][4 * j], WINV, bd[j].x), fmaf(accd[rt][4 * j + 1], WINV, bd[j].y),
                                              fmaf(accd[rt][4 * j + 2], WINV, bd[j].z), fmaf(accd[rt][4 * j + 3], WINV, bd[j].w)));
        const float4 sw = f4swish(make_float4(fmaf(accw[rt][4 * j], WINV, bw[j].x), fmaf(accw[rt][4 * j + 1], WINV, bw[j].y),
                                              fmaf(accw[rt][4 * j + 2], WINV, bw[j].z), fmaf(accw[rt][4 * j + 3], WINV, bw[j].w)));
        st4(geom, off + 32 * j, f4mul(sd, sw));  // neighbor_d * neighbor_w (scann_model.py:381-389)
      }
    }
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
  hipLaunchKernelGGL(basis_kernel, dim3((n_edge + 63) / 64), dim3(256), 0, s, p, dist, weight, n_edge, geom);
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
  // the same sum in the same order, its operands requested sixteen at a time (the one-by-one loop was emb_dim dependent round trips:
  // 13 us at the head of every training step)
  for (int k0 = 0; k0 < emb_dim; k0 += 16) {
    float e[16], w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = min(k0 + u, emb_dim - 1);
      e[u] = emb[sp * emb_dim + k];
      w[u] = W[k * D + col];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (k0 + u < emb_dim) acc += e[u] * w[u];
  }
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
  __shared__ float sHead[D];
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
  // (both halves of the workgroup walk 64 of the 128 input features each, 16 weight loads in flight: the 128 dependent-in-order
  // loads of a one-thread-per-output loop were the longest wait of this kernel)
  float part = 0.f;
  {
    const int f = tid & (D - 1), half = tid >> 7;
    float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;  // four interleaved chains, fixed association
    const float* __restrict__ wcol = a.p.Wb + (size_t)(64 * half) * D + f;
    const float* __restrict__ rp = sRep + 64 * half;
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      h0 = fmaf(rp[k], wcol[(size_t)k * D], h0);
      h1 = fmaf(rp[k + 1], wcol[(size_t)(k + 1) * D], h1);
      h2 = fmaf(rp[k + 2], wcol[(size_t)(k + 2) * D], h2);
      h3 = fmaf(rp[k + 3], wcol[(size_t)(k + 3) * D], h3);
    }
    const float hs = (h0 + h1) + (h2 + h3);
    if (half == 1) sHead[f] = hs;
    __syncthreads();
    if (half == 0) part = swish_exact((hs + sHead[f]) + a.p.bb[f]) * a.p.wo[f];
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

