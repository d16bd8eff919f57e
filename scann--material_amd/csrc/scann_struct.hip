// Structure-resident forward (gfx950): ONE workgroup runs every LocalAttention / ResidualNorm iteration of a run of whole
// structures without leaving the CU between layers (scann_model.py:413-421 loop, attention.py:118-216, :37-40, scann_model.py:424,
// attention.py:269-272).
//
// Why: in the layer-streamed forward (scann_kernels.hip) every layer reads and writes the geometry tensor [n_edge,128] -- E x 1 KiB
// per layer, 80 % of the path's traffic and two thirds of edge_kernel's vector-memory instructions -- only because a layer's atom
// side needs the contexts of ALL atoms before any edge of the next layer can run.  That dependency is local to a structure: every
// neighbour of an atom is an atom of the same structure (custom_layers.py:18-28).  Here a workgroup owns a GROUP (a run of whole
// structures, <= NT edge tiles of <= 64 edges of whole atoms) and keeps
//   * the group's geometry rows IN REGISTERS from the basis MLP to the last layer, exact fp32, in the accumulator layout of the
//     32x32 MFMA (lane = edge row, 16 of the wave's 32 columns): they are the residual of the geometry update and the source of the
//     hi / lo operand planes, which are staged in LDS one tile at a time exactly as edge_kernel stages them;
//   * the group's atom rows in two LDS arrays: X = P3 (phase A) / centres (phase B), both gathered by NEIGHBOUR atom, and
//     Y = context in -> P1 (phase A) -> query, then context out (phase B), all read by CENTRE atom;
//   * the edge rows' (neighbour, centre) indices and the atoms' edge offsets in LDS.
// Only the centres and the query rows take a round trip through the batch's scratch arrays (written by the atom phase, copied into
// X / Y before phase B; L2 hits, ordered by workgroup barriers) -- nothing inside the per-tile chain waits for global memory.
// Per layer:
//   atom phase   c = ResidualNorm(ctx) (layer 0: the embedding row); P1 = c W1 + bg, P3 = c W3, q = c Wq + bq          32-row tiles
//   phase A      W2 resident in registers; per tile: G planes <- registers, U = G W2, T = swish(U + P1[i] + P3[j]) + G,
//                G' = LayerNorm_g(T) -> registers (never stored)
//   phase B      Wk resident; per tile: ang = c[j] * G' -> planes, K = ang Wk + bk, logits, softmax over each atom's edges,
//                context + unscaled query, LayerNorm -> Y
// and after the last layer the readout's atom side (after_Lc, GlobalAttention query / key).  The tile loops are RUNTIME loops (the
// whole kernel has to stay inside the 64 KB instruction cache: the first, fully unrolled version was 80 KB and 2x slower than the
// streamed path); the pieces that touch a tile's registers sit in a switch over the tile number.  Every formula is the
// instruction sequence of atom_kernel / edge_kernel on the same rows (a row's result there does not depend on its tile), so a
// structure gives the same BYTES on either path (tests/test_gpu_parity.py::test_resident_and_streamed_structures_mixed).
#include "scann_internal.h"
#include <cstdio>
#include <cstdlib>

#include "scann_mma.h"

namespace scann {

// A pointer read from a table in memory (SrArgs::layers) is a generic pointer to the compiler: flat loads, 64-bit address arithmetic
// per lane and waits on both memory counters.  It is a global pointer: say so.
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)p;
}

// Diagnostic build only (-DSCANN_STAMPS): shader-clock stamps of layer 2, 64 per group, written to a buffer nothing else reads.
#ifdef SCANN_STAMPS
#define SR_STAMP(slot)                                                                        \
  do {                                                                                        \
    if (a.stamps && l == 2 && threadIdx.x == 0) {                                             \
      unsigned long long t_;                                                                  \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
      a.stamps[(size_t)blockIdx.x * 64 + (slot)] = t_;                                        \
    }                                                                                         \
  } while (0)
#else
#define SR_STAMP(slot) do {} while (0)
#endif

typedef float4 GTile[2][4];  // one tile's geometry rows of a lane: row tile rt, columns cbase + 8 j .. + 3 of row lrow + 32 rt

// acc = X[32 rows][0 .. 128) . W[0 .. 128)[32 wave .. +32): one row tile against a whole resident weight slab; the MFMA order per
// accumulator is mma_split's (k-step ascending; lo.hi, hi.lo, hi.hi)
__device__ __forceinline__ void mma_slab(const _Float16* __restrict__ sH, const _Float16* __restrict__ sL, const f16x8 (&wh)[8],
                                         const f16x8 (&wl)[8], int lane, f32x16& acc) {
  const int off = (lane & 31) * PLANE_STRIDE + 8 * (lane >> 5);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const f16x8 xh = *reinterpret_cast<const f16x8*>(sH + off + 16 * s);
    const f16x8 xl = *reinterpret_cast<const f16x8*>(sL + off + 16 * s);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], xh, s == 0 ? zero : acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xh, acc, 0, 0, 0);
  }
}

__device__ __forceinline__ void plane_store(_Float16* sH, _Float16* sL, int at, const float4 v) {
  f16x4 h, lo;
  split4(v, h, lo);
  *reinterpret_cast<f16x4*>(sH + at) = h;
  *reinterpret_cast<f16x4*>(sL + at) = lo;
}

// ---- the pieces that touch one tile's registers (called from a switch over the tile number) ---------------------------------------

// phase A, start: zero the rows past the tile's edges, planes <- G
__device__ __forceinline__ void g_to_planes(GTile& G, int ne, _Float16* sH, _Float16* sL, int lrow, int cbase) {
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int row = lrow + 32 * rt;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (row >= ne) G[rt][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      plane_store(sH, sL, row * PLANE_STRIDE + cbase + 8 * j, G[rt][j]);
    }
  }
}

// phase A, after the product of row tile rt: T = swish(V) + G (V = U + P1[i] + P3[j]) takes G's registers; the row's partial
// LayerNorm_g statistics of this lane pair (mean over its 32 columns, sum of squared deviations) as edge_kernel forms them
__device__ __forceinline__ float2 g_residual_stats(GTile& G, int rt, const float4 (&v)[4]) {
#pragma clang fp contract(off)
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 t = f4swish_plus(v[j], G[rt][j]);
    G[rt][j] = t;
    s += f4sum(t);
  }
  const float mean32 = xor32(s) * (1.0f / 32.0f);
  float v2 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float d;
    d = G[rt][j].x - mean32; v2 = fmaf(d, d, v2);
    d = G[rt][j].y - mean32; v2 = fmaf(d, d, v2);
    d = G[rt][j].z - mean32; v2 = fmaf(d, d, v2);
    d = G[rt][j].w - mean32; v2 = fmaf(d, d, v2);
  }
  return make_float2(mean32, xor32(v2));
}

// phase A, end: G' = LayerNorm_g(T) (attention.py:153) -> registers
__device__ __forceinline__ void g_layer_norm(GTile& G, bool two, int ne, const float* sE, const float* sPar, int lrow, int cbase,
                                             int32_t* range_flag, int l) {
#pragma clang fp contract(off)
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
    if (rt == 0 || two) {
      const int row = lrow + 32 * rt;
      const float4 sa = *reinterpret_cast<const float4*>(&sE[row * 8]), sb = *reinterpret_cast<const float4*>(&sE[row * 8 + 4]);
      const float mean = ((sa.x + sa.z) + (sb.x + sb.z)) * 0.25f;
      const float d0 = sa.x - mean, d1 = sa.z - mean, d2 = sb.x - mean, d3 = sb.z - mean;
      const float var = (((sa.y + sa.w) + (sb.y + sb.w)) + 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3))) * (1.0f / D);
      const float rstd = 1.0f / sqrtf(var + 1e-6f);
      if (!(var < RANGE_FINITE) && row < ne) flag_range(range_flag, 1, l);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 gm = *reinterpret_cast<const float4*>(&sPar[cbase + 8 * j]);
        const float4 be = *reinterpret_cast<const float4*>(&sPar[D + cbase + 8 * j]);
        const float4 t = G[rt][j];
        float4 y;
        float inv;
        inv = rstd * gm.x; y.x = fmaf(t.x, inv, be.x - mean * inv);
        inv = rstd * gm.y; y.y = fmaf(t.y, inv, be.y - mean * inv);
        inv = rstd * gm.z; y.z = fmaf(t.z, inv, be.z - mean * inv);
        inv = rstd * gm.w; y.w = fmaf(t.w, inv, be.w - mean * inv);
        G[rt][j] = y;  // geom' (scann_model.py:415): the next layer's input, never stored
      }
    }
}

// phase B, start: ang = c[j] * G' (attention.py:136,157) -> planes; rows past the tile's edges are zeros
__device__ __forceinline__ void g_gate_to_planes(const GTile& G, bool two, int ne, const float* sX, const int (&nbx)[2], _Float16* sH,
                                                 _Float16* sL, int lrow, int cbase) {
#pragma clang fp contract(off)
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int row = lrow + 32 * rt;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 ang = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rt == 0 || two) ang = f4mul(*reinterpret_cast<const float4*>(&sX[nbx[rt] + 8 * j]), G[rt][j]);
      if (row >= ne) ang = make_float4(0.f, 0.f, 0.f, 0.f);
      plane_store(sH, sL, row * PLANE_STRIDE + cbase + 8 * j, ang);
    }
  }
}

#define SR_TILE_SWITCH(p, CALL)                                                                         \
  switch (p) {                                                                                          \
    default: { GTile& G = greg[0]; CALL; } break;                                                       \
    case 1: if (NT > 1) { GTile& G = greg[NT > 1 ? 1 : 0]; CALL; } break;                               \
    case 2: if (NT > 2) { GTile& G = greg[NT > 2 ? 2 : 0]; CALL; } break;                               \
    case 3: if (NT > 3) { GTile& G = greg[NT > 3 ? 3 : 0]; CALL; } break;                               \
    case 4: if (NT > 4) { GTile& G = greg[NT > 4 ? 4 : 0]; CALL; } break;                               \
    case 5: if (NT > 5) { GTile& G = greg[NT > 5 ? 5 : 0]; CALL; } break;                               \
  }

// ---- atom phase: atom_kernel<FFN, MODE, 1>'s arithmetic on the 32-row atom tiles of the group -----------------------------------
// MODE 0: c, P1, P3, q of layer l.  MODE 2: gq, gk (readout).  Input rows: layer 0 the embedding rows (global), afterwards the context
// rows in Y.  MODE 0 outputs: c, q -> scratch arrays; P1 -> Y (the input rows are in registers by then); P3 -> X.
template <int MODE>
__device__ __forceinline__ void sr_atom_phase(const SrArgs& a, const SrGroup& g, int l, _Float16* sH, _Float16* sL, float* sRed, float* sPar,
                                              float* sX, float* sY, int tid, int lane, int wave) {
#pragma clang fp contract(off)
  constexpr int RT = 1;
  constexpr float WINV = 1.0f / WSCALE;
  const int lrow = lane & 31, lh = lane >> 5, cbase = 32 * wave + 4 * lh;
  const bool ffn = l > 0 && a.use_attn_norm;
  const LayerParams* const pp = a.layers + (l > 0 ? l - 1 : 0);  // the ResidualNorm that follows LocalAttention l - 1
  const _Float16 *WA, *WB, *WC, *WD;
  const float *bA, *bC, *bD;
  if (MODE == 0) {
    const LayerParams* const p = a.layers + l;
    WA = as_global(p->W1h); WB = as_global(p->W3h); WC = as_global(p->Wqh); WD = nullptr;
    bA = as_global(p->bg); bC = as_global(p->bq); bD = bC;
  } else {
    WA = a.head.Wah; WB = nullptr; WC = a.head.Wgqh; WD = a.head.Wgkh; bA = a.head.ba; bC = a.head.bgq; bD = a.head.bgk;
  }
  const float* const bf1 = ffn ? as_global(pp->bf1) : bC;
  const float* const bf2 = ffn ? as_global(pp->bf2) : bC;
  const float* const lng = ffn ? as_global(pp->lnr_g) : bC;
  const float* const lnb = ffn ? as_global(pp->lnr_b) : bC;
  const _Float16* const Wf1 = as_global(pp->Wf1h);
  const _Float16* const Wf2 = as_global(pp->Wf2h);
#pragma nounroll
  for (int row0 = g.atom_begin; row0 < g.atom_end; row0 += 32) {
    const int nrows = min(32, g.atom_end - row0);
    const int r0 = row0 - g.atom_begin;  // first row of the tile in X / Y
    f16x8 whA[4], wlA[4], whB[4], wlB[4];
    load_wsplit<4, 8>(ffn ? Wf1 : WA, wave, lane, whA, wlA, 0);
    load_wsplit<4, 8>(ffn ? Wf1 : WA, wave, lane, whB, wlB, 4);
    {  // bf1 | bf2 | lnr_g | lnr_b | bA | bC | bD  (absent ones read a valid dummy row and are never used)
      const int k = tid & (D - 1);
      const float p0 = (tid < D ? bf1 : bf2)[k], p1 = (tid < D ? lng : lnb)[k], p2 = (tid < D ? bA : bC)[k], p3 = bD[k];
      sPar[tid] = p0;
      sPar[256 + tid] = p1;
      sPar[512 + tid] = p2;
      if (tid < D) sPar[768 + tid] = p3;
    }
    float4 xr[4];
    const unsigned ooff = ((unsigned)(row0 + lrow) * D + cbase) * 4;
    const int yat = (r0 + lrow) * LDS_STRIDE + cbase;  // (this lane's row, first column) in X / Y
    if (l == 0) {
      const int rc = row0 + min(lrow, nrows - 1);
      const int src = a.x0_index ? a.x0_index[rc] : rc;
      const unsigned soff = ((unsigned)src * D + cbase) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) xr[j] = ld4(a.x0, soff + 32 * j);
    } else {
      const float* xs = sY + (r0 + min(lrow, nrows - 1)) * LDS_STRIDE + cbase;
#pragma unroll
      for (int j = 0; j < 4; ++j) xr[j] = *reinterpret_cast<const float4*>(xs + 8 * j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = lrow < nrows ? xr[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!ffn && lrow < nrows && MODE == 0) st4(a.c, ooff + 32 * j, v);  // centres = staged rows (layer 0 / no ResidualNorm)
      if (!ffn && lrow < nrows && a.dbg_c) st4(a.dbg_c + (size_t)l * a.n_atom_total * D, ooff + 32 * j, v);
      xr[j] = v;
      plane_store(sH, sL, lrow * PLANE_STRIDE + cbase + 8 * j, v);
    }
    __syncthreads();
    f32x16 acc[RT];
    if (ffn) {
      // ResidualNorm (attention.py:37-40): h = swish(x W1 + b1)
      gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, Wf2, wave, lane, acc);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bv = *reinterpret_cast<const float4*>(&sPar[cbase + 8 * j]);
        const float4 pre = make_float4(fmaf(acc[0][4 * j], WINV, bv.x), fmaf(acc[0][4 * j + 1], WINV, bv.y),
                                       fmaf(acc[0][4 * j + 2], WINV, bv.z), fmaf(acc[0][4 * j + 3], WINV, bv.w));
        plane_store(sH, sL, lrow * PLANE_STRIDE + cbase + 8 * j, f4swish(pre));
      }
      __syncthreads();
      // y = h W2 + b2 ; t = x + y
      gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, WA, wave, lane, acc);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bv = *reinterpret_cast<const float4*>(&sPar[D + cbase + 8 * j]);
        const float4 y = make_float4(fmaf(acc[0][4 * j], WINV, bv.x), fmaf(acc[0][4 * j + 1], WINV, bv.y),
                                     fmaf(acc[0][4 * j + 2], WINV, bv.z), fmaf(acc[0][4 * j + 3], WINV, bv.w));
        const float4 t2 = f4add(xr[j], y);
        acc[0][4 * j] = t2.x; acc[0][4 * j + 1] = t2.y; acc[0][4 * j + 2] = t2.z; acc[0][4 * j + 3] = t2.w;
        s += f4sum(t2);
      }
      const float mean32 = xor32(s) * (1.0f / 32.0f);
      float v2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = acc[0][i] - mean32;
        v2 = fmaf(d, d, v2);
      }
      const float m2 = xor32(v2);
      if (lh == 0) *reinterpret_cast<float2*>(&sRed[(lrow * 4 + wave) * 2]) = make_float2(mean32, m2);
      __syncthreads();
      {
        const float4 sa = *reinterpret_cast<const float4*>(&sRed[lrow * 8]), sb = *reinterpret_cast<const float4*>(&sRed[lrow * 8 + 4]);
        const float mean = ((sa.x + sa.z) + (sb.x + sb.z)) * 0.25f;
        const float d0 = sa.x - mean, d1 = sa.z - mean, d2 = sb.x - mean, d3 = sb.z - mean;
        const float var = (((sa.y + sa.w) + (sb.y + sb.w)) + 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3))) * (1.0f / D);
        const float rstd = 1.0f / sqrtf(var + 1e-6f);
        if (!(var < RANGE_FINITE) && lrow < nrows) flag_range(a.range_flag, 3, l - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 gm = *reinterpret_cast<const float4*>(&sPar[2 * D + cbase + 8 * j]);
          const float4 be = *reinterpret_cast<const float4*>(&sPar[3 * D + cbase + 8 * j]);
          float4 y;
          float inv;
          inv = rstd * gm.x; y.x = fmaf(acc[0][4 * j], inv, be.x - mean * inv);
          inv = rstd * gm.y; y.y = fmaf(acc[0][4 * j + 1], inv, be.y - mean * inv);
          inv = rstd * gm.z; y.z = fmaf(acc[0][4 * j + 2], inv, be.z - mean * inv);
          inv = rstd * gm.w; y.w = fmaf(acc[0][4 * j + 3], inv, be.w - mean * inv);
          if (lrow < nrows && MODE == 0) st4(a.c, ooff + 32 * j, y);
          if (lrow < nrows && a.dbg_c) st4(a.dbg_c + (size_t)l * a.n_atom_total * D, ooff + 32 * j, y);
          plane_store(sH, sL, lrow * PLANE_STRIDE + cbase + 8 * j, y);
        }
      }
      __syncthreads();
    }
    if (MODE == 0) {  // P1 = c W1 + bg ; P3 = c W3 ; q = c Wq + bq (attention.py:142-151 thirds, :160)
      gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, WB, wave, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bg = *reinterpret_cast<const float4*>(&sPar[4 * D + cbase + 8 * j]);
        if (lrow < nrows)
          *reinterpret_cast<float4*>(&sY[yat + 8 * j]) = make_float4(fmaf(acc[0][4 * j], WINV, bg.x), fmaf(acc[0][4 * j + 1], WINV, bg.y),
                                                                    fmaf(acc[0][4 * j + 2], WINV, bg.z), fmaf(acc[0][4 * j + 3], WINV, bg.w));
      }
      gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, WC, wave, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (lrow < nrows)
          *reinterpret_cast<float4*>(&sX[yat + 8 * j]) =
              make_float4(acc[0][4 * j] * WINV, acc[0][4 * j + 1] * WINV, acc[0][4 * j + 2] * WINV, acc[0][4 * j + 3] * WINV);
      gemm_tile<false, RT>(sH, sL, whA, wlA, whB, wlB, nullptr, wave, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bq = *reinterpret_cast<const float4*>(&sPar[5 * D + cbase + 8 * j]);
        if (lrow < nrows)
          st4(a.q, ooff + 32 * j, make_float4(fmaf(acc[0][4 * j], WINV, bq.x), fmaf(acc[0][4 * j + 1], WINV, bq.y),
                                              fmaf(acc[0][4 * j + 2], WINV, bq.z), fmaf(acc[0][4 * j + 3], WINV, bq.w)));
      }
    } else {  // z = swish(c Wa + ba) (scann_model.py:424); gq = z Wgq + b ; gk = z Wgk + b (attention.py:269-272)
      gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, WC, wave, lane, acc);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bv = *reinterpret_cast<const float4*>(&sPar[4 * D + cbase + 8 * j]);
        const float4 pre = make_float4(fmaf(acc[0][4 * j], WINV, bv.x), fmaf(acc[0][4 * j + 1], WINV, bv.y),
                                       fmaf(acc[0][4 * j + 2], WINV, bv.z), fmaf(acc[0][4 * j + 3], WINV, bv.w));
        const float4 z = f4swish(pre);
        if (!(fmaxf(fmaxf(fabsf(z.x), fabsf(z.y)), fmaxf(fabsf(z.z), fabsf(z.w))) < 65504.f) && lrow < nrows) flag_range(a.range_flag, 4, l);
        plane_store(sH, sL, lrow * PLANE_STRIDE + cbase + 8 * j, z);
      }
      __syncthreads();
      gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, WD, wave, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bq = *reinterpret_cast<const float4*>(&sPar[5 * D + cbase + 8 * j]);
        if (lrow < nrows)
          st4(a.gq, ooff + 32 * j, make_float4(fmaf(acc[0][4 * j], WINV, bq.x), fmaf(acc[0][4 * j + 1], WINV, bq.y),
                                               fmaf(acc[0][4 * j + 2], WINV, bq.z), fmaf(acc[0][4 * j + 3], WINV, bq.w)));
      }
      gemm_tile<false, RT>(sH, sL, whA, wlA, whB, wlB, nullptr, wave, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bk = *reinterpret_cast<const float4*>(&sPar[6 * D + cbase + 8 * j]);
        if (lrow < nrows)
          st4(a.gk, ooff + 32 * j, make_float4(fmaf(acc[0][4 * j], WINV, bk.x), fmaf(acc[0][4 * j + 1], WINV, bk.y),
                                               fmaf(acc[0][4 * j + 2], WINV, bk.z), fmaf(acc[0][4 * j + 3], WINV, bk.w)));
      }
    }
    __syncthreads();  // rows stored (LDS and scratch: visible to the whole workgroup); planes / parameters may be overwritten
  }
}

template <int NT>
#ifdef SR_OCC1
__global__ __launch_bounds__(256, 1) void sr_kernel(SrArgs a) {
#else
__global__ __launch_bounds__(256, NT <= 3 ? 2 : 1) void sr_kernel(SrArgs a) {
#endif
#pragma clang fp contract(off)  // fusions are written out (fmaf), as in edge_kernel: a row's result must not depend on its place
  constexpr int NATOM = NT <= 3 ? SR_ATOMS_SMALL : SR_ATOMS_BIG;
  __shared__ __attribute__((aligned(16))) unsigned char sTile[2 * TE_MAX * PLANE_STRIDE * 2];  // hi / lo planes; afterwards K as fp32
  __shared__ __attribute__((aligned(16))) float sX[NATOM * LDS_STRIDE];  // atom rows read by NEIGHBOUR: P3 (phase A), centres (phase B)
  __shared__ __attribute__((aligned(16))) float sY[NATOM * LDS_STRIDE];  // atom rows read by CENTRE: context -> P1 -> query -> context
  __shared__ __attribute__((aligned(16))) float sE[TE_MAX * NHEAD];      // LayerNorm partial statistics, then logits
  __shared__ __attribute__((aligned(16))) float sPar[7 * D];
  __shared__ int sIdx[NT * TE_MAX];     // per edge row of tile p: neighbour atom | centre atom << 16, both relative to the group
  __shared__ int sOffs[NT][TQ + 1];     // per tile: edge offsets of its atoms relative to the tile's first edge
  static_assert(sizeof(sTile) >= TE_MAX * LDS_STRIDE * sizeof(float), "K tile must fit the plane buffer");
  _Float16* const sH = reinterpret_cast<_Float16*>(sTile);
  _Float16* const sL = sH + TE_MAX * PLANE_STRIDE;
  float* const sK = reinterpret_cast<float*>(sTile);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5, cbase = 32 * wave + 4 * lh;
  const SrGroup g = a.groups[blockIdx.x];
  const int nt = g.n_tile;
  constexpr float WINV = 1.0f / WSCALE;

  GTile greg[NT];  // the group's geometry rows: registers for the whole forward

  // ---- once per group: indices into LDS; geom0 = swish(Gauss(dist) Wd + bd) * swish(Gauss(weight) Ww + bw)
  //      (scann_model.py:378-389: basis_kernel's instruction sequence) -> registers ------------------------------------------------
  {
    _Float16* const bH = reinterpret_cast<_Float16*>(sTile);
    _Float16* const bL = bH + TE_MAX * BASIS_STRIDE;
    f16x8 bdh[2], bdl[2], bwh[2], bwl[2];
    load_wsplit<2>(a.basis.Wdh, wave, lane, bdh, bdl);
    load_wsplit<2>(a.basis.Wwh, wave, lane, bwh, bwl);
#pragma nounroll
    for (int p = 0; p < nt; ++p) {
      const EdgeTile tile = a.tiles[g.tile_begin + p];
      const int eb = tile.edge_begin, ne = tile.edge_end - eb, nem1 = ne > 0 ? ne - 1 : 0;
      const int natom = tile.atom_end - tile.atom_begin;
      if (tid < TE_MAX) {
        const int e = ne > 0 ? eb + min(tid, nem1) : 0;
        const int nb = ne > 0 ? a.edge_col[e] - g.atom_begin : 0, ct = ne > 0 ? a.edge_row[e] - g.atom_begin : 0;
        sIdx[p * TE_MAX + tid] = nb | (ct << 16);
      } else if (tid - TE_MAX <= natom) {
        sOffs[p][tid - TE_MAX] = a.edge_offset[tile.atom_begin + tid - TE_MAX] - eb;
      }
      const int r = tid >> 2, sub = tid & 3;
      const int rs = r < ne ? r : nem1;
      const float xd = ne > 0 ? a.dist[eb + rs] : 0.f, xw = ne > 0 ? a.edge_weight[eb + rs] : 0.f;
      f16x8 gh[2], gl[2];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 8 * sub + i;
        const float cdk = a.basis.cd[min(k, NG - 1)], cwk = a.basis.cw[min(k, NG - 1)];
        const float vd = (k < NG && r < ne) ? gauss_fast(xd, cdk) : 0.f, vw = (k < NG && r < ne) ? gauss_fast(xw, cwk) : 0.f;
        gh[0][i] = (_Float16)vd; gl[0][i] = (_Float16)(vd - (float)gh[0][i]);
        gh[1][i] = (_Float16)vw; gl[1][i] = (_Float16)(vw - (float)gh[1][i]);
      }
      *reinterpret_cast<f16x8*>(bH + r * BASIS_STRIDE + 8 * sub) = gh[0];
      *reinterpret_cast<f16x8*>(bL + r * BASIS_STRIDE + 8 * sub) = gl[0];
      *reinterpret_cast<f16x8*>(bH + r * BASIS_STRIDE + 32 + 8 * sub) = gh[1];
      *reinterpret_cast<f16x8*>(bL + r * BASIS_STRIDE + 32 + 8 * sub) = gl[1];
      __syncthreads();
      f32x16 accd[2], accw[2];
      mma_split<2, true, BASIS_STRIDE, 2>(bH, bL, bdh, bdl, lane, accd);
      mma_split<2, true, BASIS_STRIDE, 2>(bH + 32, bL + 32, bwh, bwl, lane, accw);
      GTile g0;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 bd = *reinterpret_cast<const float4*>(a.basis.bd + cbase + 8 * j);
          const float4 bw = *reinterpret_cast<const float4*>(a.basis.bw + cbase + 8 * j);
          const float4 sd = f4swish(make_float4(fmaf(accd[rt][4 * j], WINV, bd.x), fmaf(accd[rt][4 * j + 1], WINV, bd.y),
                                                fmaf(accd[rt][4 * j + 2], WINV, bd.z), fmaf(accd[rt][4 * j + 3], WINV, bd.w)));
          const float4 sw = f4swish(make_float4(fmaf(accw[rt][4 * j], WINV, bw.x), fmaf(accw[rt][4 * j + 1], WINV, bw.y),
                                                fmaf(accw[rt][4 * j + 2], WINV, bw.z), fmaf(accw[rt][4 * j + 3], WINV, bw.w)));
          g0[rt][j] = f4mul(sd, sw);
        }
      SR_TILE_SWITCH(p, {
        _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)
          _Pragma("unroll") for (int j = 0; j < 4; ++j) G[rt][j] = g0[rt][j];
      })
      __syncthreads();  // every wave is done reading the basis planes
    }
  }

  const int L = a.n_layer;
#pragma nounroll
  for (int l = 0; l < L; ++l) {
    SR_STAMP(0);
    sr_atom_phase<0>(a, g, l, sH, sL, sE, sPar, sX, sY, tid, lane, wave);
    SR_STAMP(1);
    const LayerParams* const P = a.layers + l;
    f16x8 wh[8], wl[8];

    // ---- phase A: geometry update (attention.py:141-153), W2 resident ------------------------------------------------------------
    load_wsplit<8>(as_global(P->W2h), wave, lane, wh, wl);
    sPar[tid] = (tid < D ? as_global(P->lng_g) : as_global(P->lng_b))[tid & (D - 1)];
    sPar[2 * D + tid] = (tid < D ? as_global(P->ln_g) : as_global(P->ln_b))[tid & (D - 1)];
    if (tid < D) sPar[4 * D + tid] = as_global(P->bk)[tid];
#pragma nounroll
    for (int p = 0; p < nt; ++p) {
      const EdgeTile tile = a.tiles[g.tile_begin + p];
      const int ne = tile.edge_end - tile.edge_begin;
      const bool two = ne > 32;  // the second row tile holds edges
      int nbx[2], ctx_at[2];     // (neighbour atom's row, this lane's first column) in X; (centre atom's row, ...) in Y
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int ix = sIdx[p * TE_MAX + lrow + 32 * rt];
        nbx[rt] = (ix & 0xffff) * LDS_STRIDE + cbase;
        ctx_at[rt] = (ix >> 16) * LDS_STRIDE + cbase;
      }
      SR_STAMP(3 + 4 * p);
      SR_TILE_SWITCH(p, g_to_planes(G, ne, sH, sL, lrow, cbase))
      __syncthreads();  // planes (and, first tile, the layer's parameters) complete
      SR_STAMP(4 + 4 * p);
      // T = swish(U + P1[i] + P3[j]) + G on the accumulators, one row tile at a time (T takes G's registers: G is dead once T
      // exists); LayerNorm_g partial statistics
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        if (rt == 0 || two) {
          f32x16 acc;
          mma_slab(sH + rt * 32 * PLANE_STRIDE, sL + rt * 32 * PLANE_STRIDE, wh, wl, lane, acc);
          float4 sw[4];  // V = U + P1[i] + P3[j]
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 p1v = *reinterpret_cast<const float4*>(&sY[ctx_at[rt] + 8 * j]);  // centre third P1[i] = c_i W1 + bg
            const float4 p3v = *reinterpret_cast<const float4*>(&sX[nbx[rt] + 8 * j]);     // neighbour third P3[j] = c_j W3
            sw[j].x = fmaf(acc[4 * j], WINV, p1v.x) + p3v.x;
            sw[j].y = fmaf(acc[4 * j + 1], WINV, p1v.y) + p3v.y;
            sw[j].z = fmaf(acc[4 * j + 2], WINV, p1v.z) + p3v.z;
            sw[j].w = fmaf(acc[4 * j + 3], WINV, p1v.w) + p3v.w;
          }
          float2 st;
          SR_TILE_SWITCH(p, st = g_residual_stats(G, rt, sw))
          if (lh == 0) *reinterpret_cast<float2*>(&sE[((lrow + 32 * rt) * 4 + wave) * 2]) = st;
        }
      SR_STAMP(5 + 4 * p);
      __syncthreads();  // statistics complete; every wave is done with the G planes
      SR_TILE_SWITCH(p, g_layer_norm(G, two, ne, sE, sPar, lrow, cbase, a.range_flag, l))
      if (a.dbg_g) {  // test hook (env SCANN_SR_DEBUG): geom' of this layer, for scann_debug_read
        SR_TILE_SWITCH(p, {
          _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)
            _Pragma("unroll") for (int j = 0; j < 4; ++j)
              if (lrow + 32 * rt < ne)
                st4(a.dbg_g + (size_t)(l + 1) * a.n_edge_total * D, ((unsigned)(tile.edge_begin + lrow + 32 * rt) * D + cbase + 8 * j) * 4, G[rt][j]);
        })
      }
      SR_STAMP(6 + 4 * p);
    }

    // ---- phase B: gate, keys, softmax, context (attention.py:136,157-214), Wk resident ---------------------------------------------
    SR_STAMP(15);
    load_wsplit<8>(as_global(P->Wkh), wave, lane, wh, wl);
    {  // centres -> X, query rows -> Y (every wave is past its last read of the P3 / P1 rows: the barrier above)
      const int nrow4 = (g.atom_end - g.atom_begin) * 32;
#pragma nounroll
      for (int i = tid; i < nrow4; i += 256) {
        const unsigned go = ((unsigned)g.atom_begin * 32 + i) * 16;
        const float4 cv = ld4(a.c, go), qv = ld4(a.q, go);
        const int at = (i >> 5) * LDS_STRIDE + 4 * (i & 31);
        *reinterpret_cast<float4*>(&sX[at]) = cv;
        *reinterpret_cast<float4*>(&sY[at]) = qv;
      }
    }
    __syncthreads();
    SR_STAMP(16);
#pragma nounroll
    for (int p = 0; p < nt; ++p) {
      const EdgeTile tile = a.tiles[g.tile_begin + p];
      const int ne = tile.edge_end - tile.edge_begin, natom = tile.atom_end - tile.atom_begin;
      const int arow0 = tile.atom_begin - g.atom_begin;  // the tile's first atom row in Y
      const bool two = ne > 32;
      int nbx[2], ctx_at[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int ix = sIdx[p * TE_MAX + lrow + 32 * rt];
        nbx[rt] = (ix & 0xffff) * LDS_STRIDE + cbase;
        ctx_at[rt] = (ix >> 16) * LDS_STRIDE + cbase;
      }
      SR_STAMP(17 + 6 * p);
      SR_TILE_SWITCH(p, g_gate_to_planes(G, two, ne, sX, nbx, sH, sL, lrow, cbase))
      __syncthreads();  // ang planes complete
      SR_STAMP(18 + 6 * p);
      f32x16 acc[2];
      mma_slab(sH, sL, wh, wl, lane, acc[0]);
      if (two) mma_slab(sH + 32 * PLANE_STRIDE, sL + 32 * PLANE_STRIDE, wh, wl, lane, acc[1]);
      // logits e[n, h] = (q[i, h, :] * 16^-0.5) . K[n, h, :] (attention.py:180-183) from the accumulators
      float lg[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        if (rt == 0 || two) {
          const float* qrow = sY + ctx_at[rt];
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
      SR_STAMP(19 + 6 * p);
      __syncthreads();  // every wave is done reading the ang planes (and the query rows): K may overwrite the planes
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        if (rt == 0 || two) {
          const int row = lrow + 32 * rt;
          sE[row * NHEAD + 2 * wave + lh] = lh ? lg[rt][1] : lg[rt][0];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(&sK[row * LDS_STRIDE + cbase + 8 * j]) =
                make_float4(acc[rt][4 * j], acc[rt][4 * j + 1], acc[rt][4 * j + 2], acc[rt][4 * j + 3]);
        }
      __syncthreads();
      SR_STAMP(20 + 6 * p);
      // softmax over each atom's edges + context + unscaled-query residual (attention.py:186-212): edge_kernel's loop, the context
      // row built in place of the query row in Y
      {
        const int lgp = tid >> 5, c4 = tid & 31, h = c4 >> 2;
        for (int la = lgp; la < natom; la += 8) {
          const int e0 = sOffs[p][la], e1 = sOffs[p][la + 1];
          float m = -INFINITY;
          for (int n = e0; n < e1; n += 4) {
            const float v0 = sE[n * NHEAD + h], v1 = sE[min(n + 1, e1 - 1) * NHEAD + h];
            const float v2 = sE[min(n + 2, e1 - 1) * NHEAD + h], v3 = sE[min(n + 3, e1 - 1) * NHEAD + h];
            m = fmaxf(fmaxf(m, fmaxf(v0, v1)), fmaxf(v2, v3));
          }
          float ssum = 0.f;
          float4 cx = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int n = e0; n < e1; n += 2) {
            const bool tw = n + 1 < e1;
            const int n1 = tw ? n + 1 : n;
            const float4 ka = *reinterpret_cast<const float4*>(&sK[n * LDS_STRIDE + 4 * c4]);
            const float4 kb = *reinterpret_cast<const float4*>(&sK[n1 * LDS_STRIDE + 4 * c4]);
            const float pa = fast_exp(sE[n * NHEAD + h] - m), pb = tw ? fast_exp(sE[n1 * NHEAD + h] - m) : 0.f;
            ssum += pa + pb;
            cx.x = fmaf(pa, ka.x, fmaf(pb, kb.x, cx.x));
            cx.y = fmaf(pa, ka.y, fmaf(pb, kb.y, cx.y));
            cx.z = fmaf(pa, ka.z, fmaf(pb, kb.z, cx.z));
            cx.w = fmaf(pa, ka.w, fmaf(pb, kb.w, cx.w));
          }
          const float rs = e1 > e0 ? __builtin_amdgcn_rcpf(ssum) : 0.f;
          float4* qp = reinterpret_cast<float4*>(&sY[(arow0 + la) * LDS_STRIDE + 4 * c4]);
          const float4 q4 = *qp;
          *qp = make_float4(fmaf(cx.x, rs, q4.x), fmaf(cx.y, rs, q4.y), fmaf(cx.z, rs, q4.z), fmaf(cx.w, rs, q4.w));
        }
      }
      // LayerNorm of the context rows (attention.py:214): 8 lanes per atom row, the atoms THIS wave has just summed (half-wave
      // lgp took atoms lgp, lgp + 8, lgp + 16), so no workgroup barrier in between; in place
      {
        const int gi = lane >> 3, sb = lane & 7;
        const int la = 2 * wave + (gi & 1) + 8 * (gi >> 1);
        if (la < natom) {
          float* row = sY + (arow0 + la) * LDS_STRIDE;
          float4 t[4];
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            t[i] = *reinterpret_cast<const float4*>(&row[4 * (sb + 8 * i)]);
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
          if (!(v < RANGE_FINITE)) flag_range(a.range_flag, 2, l);
          const float rstd = 1.0f / sqrtf(v * (1.0f / D) + 1e-6f);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c4 = sb + 8 * i;
            const float4 gm = *reinterpret_cast<const float4*>(&sPar[2 * D + 4 * c4]);
            const float4 be = *reinterpret_cast<const float4*>(&sPar[3 * D + 4 * c4]);
            float4 y;
            float inv;
            inv = rstd * gm.x; y.x = t[i].x * inv + (be.x - mean * inv);
            inv = rstd * gm.y; y.y = t[i].y * inv + (be.y - mean * inv);
            inv = rstd * gm.z; y.z = t[i].z * inv + (be.z - mean * inv);
            inv = rstd * gm.w; y.w = t[i].w * inv + (be.w - mean * inv);
            *reinterpret_cast<float4*>(&row[4 * c4]) = y;
            if (a.dbg_ctx) st4(a.dbg_ctx + (size_t)l * a.n_atom_total * D, ((unsigned)(tile.atom_begin + la) * 32 + c4) * 16, y);
          }
        }
      }
      SR_STAMP(21 + 6 * p);
      __syncthreads();  // K rows and logits are free again; the context rows are complete
      SR_STAMP(22 + 6 * p);
    }
    SR_STAMP(35);
  }
  // readout's atom side: ResidualNorm of the last layer, after_Lc, GlobalAttention query / key (scann_model.py:424, attention.py:269-272)
  sr_atom_phase<2>(a, g, L, sH, sL, sE, sPar, sX, sY, tid, lane, wave);
}

void launch_struct(const SrArgs& a, int nt_max, hipStream_t s) {
  if (a.n_group <= 0) return;
  static bool said = false;
  if (!said && getenv("SCANN_SR_OCCUPANCY")) {  // diagnostic: resident workgroups per CU the runtime computes for the two kernels
    said = true;
    int n3 = -1, n6 = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n3, sr_kernel<3>, 256, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n6, sr_kernel<6>, 256, 0);
    hipFuncAttributes fa{};
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(sr_kernel<3>));
    fprintf(stderr, "sr_kernel<3>: %d workgroups per CU (regs %d, lds %zu, scratch %zu); sr_kernel<6>: %d\n", n3, fa.numRegs, fa.sharedSizeBytes,
            fa.localSizeBytes, n6);
  }
  const dim3 grid(a.n_group), block(256);
  if (nt_max <= 3) hipLaunchKernelGGL((sr_kernel<3>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((sr_kernel<6>), grid, block, 0, s, a);
}

}  // namespace scann
