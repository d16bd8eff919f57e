// Structure-resident forward (gfx950): ONE workgroup runs every LocalAttention / ResidualNorm iteration of a run of whole
// structures without leaving the CU between layers (scann_model.py:413-421 loop, attention.py:118-216, :37-40, scann_model.py:424,
// attention.py:269-272).
//
// Why: in the layer-streamed forward (scann_kernels.hip) every layer reads and writes the geometry tensor [n_edge,128] -- E x 1 KiB
// per layer, 80 % of the path's traffic and two thirds of edge_kernel's vector-memory instructions -- only because a layer's atom
// side needs the contexts of ALL atoms before any edge of the next layer can run.  That dependency is local to a structure: every
// neighbour of an atom is an atom of the same structure (custom_layers.py:18-28).  Here a workgroup owns a GROUP (a run of whole
// structures, <= NT edge tiles of <= 64 edges of whole atoms) and keeps the group's geometry rows IN REGISTERS from the basis MLP to
// the last layer, exact fp32, in the accumulator layout of the 32x32 MFMA (lane = edge row, 16 of the wave's 32 columns): they are
// the residual of the geometry update and the source of the hi / lo operand planes, which are staged in LDS one tile at a time exactly
// as edge_kernel stages them.  Atom rows (centres, P1, P3, q, context) go through the batch's [n_atom,128] scratch arrays, written
// and read by this workgroup only (L1 / L2 hits; workgroup-scope barriers order them).  Per layer:
//
//   atom phase   c = ResidualNorm(ctx) (layer 0: the embedding LUT row); P1 = c W1 + bg, P3 = c W3, q = c Wq + bq   32-row tiles
//   phase A      W2 resident in registers; per tile: G planes <- registers, U = G W2, T = swish(U + P1[i] + P3[j]) + G,
//                G' = LayerNorm_g(T) -> registers (never stored)
//   phase B      Wk resident; per tile: ang = c[j] * G' -> planes, K = ang Wk + bk, logits, softmax over each atom's edges,
//                context + unscaled query, LayerNorm -> ctx
//
// and after the last layer the readout's atom launch (after_Lc, GlobalAttention query / key).  Every formula is the instruction
// sequence of atom_kernel / edge_kernel on the same rows (a row's result there does not depend on its tile), so a structure gives
// the same BYTES on either path (tests/test_gpu_parity.py::test_resident_and_streamed_structures_mixed).
#include "scann_internal.h"
#include "scann_mma.h"

namespace scann {

// acc (+)= X[32 rows][0 .. 128) . W[0 .. 128)[32 wave .. +32): one row tile against a whole resident weight slab; the MFMA order per
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

// ---- atom phase: atom_kernel<FFN, MODE, 1>'s arithmetic on the 32-row atom tiles of the group -----------------------------------
// MODE 0: c, P1, P3, q of layer l.  MODE 2: gq, gk (readout).  ffn: ResidualNorm of layer l - 1 first (uniform per launch).
template <int MODE>
__device__ __forceinline__ void sr_atom_phase(const SrArgs& a, const SrGroup& g, int l, _Float16* sH, _Float16* sL, float* sRed, float* sPar,
                                              float* sX, int tid, int lane, int wave) {
  constexpr int RT = 1;
  constexpr float WINV = 1.0f / WSCALE;
  const int lrow = lane & 31, lh = lane >> 5, cbase = 32 * wave + 4 * lh;
  const bool ffn = l > 0 && a.use_attn_norm;
  const float* const x = l == 0 ? a.x0 : a.ctx;
  const int32_t* const x_index = l == 0 ? a.x0_index : nullptr;
  const LayerParams* const pp = a.layers + (l > 0 ? l - 1 : 0);  // the ResidualNorm that follows LocalAttention l - 1
  const _Float16 *WA, *WB, *WC, *WD;
  const float *bA, *bC, *bD;
  float *oA, *oB, *oC;
  if (MODE == 0) {
    const LayerParams* const p = a.layers + l;
    WA = p->W1h; WB = p->W3h; WC = p->Wqh; WD = nullptr; bA = p->bg; bC = p->bq; bD = p->bq;
    oA = a.P1; oB = nullptr; oC = a.q;
  } else {
    WA = a.head.Wah; WB = nullptr; WC = a.head.Wgqh; WD = a.head.Wgkh; bA = a.head.ba; bC = a.head.bgq; bD = a.head.bgk;
    oA = nullptr; oB = a.gk; oC = a.gq;
  }
  for (int row0 = g.atom_begin; row0 < g.atom_end; row0 += 32) {
    const int nrows = min(32, g.atom_end - row0);
    f16x8 whA[4], wlA[4], whB[4], wlB[4];
    load_wsplit<4, 8>(ffn ? pp->Wf1h : WA, wave, lane, whA, wlA, 0);
    load_wsplit<4, 8>(ffn ? pp->Wf1h : WA, wave, lane, whB, wlB, 4);
    {
      const float* const tab[7] = {ffn ? pp->bf1 : bC, ffn ? pp->bf2 : bC, ffn ? pp->lnr_g : bC, ffn ? pp->lnr_b : bC, bA, bC, bD};
      float pv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pv[i] = (tid < D ? tab[2 * i] : tab[2 * i + 1 < 7 ? 2 * i + 1 : 6])[tid & (D - 1)];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (256 * i + tid < 7 * D) sPar[256 * i + tid] = pv[i];
    }
    float4 xr[4];
    const int rc = row0 + min(lrow, nrows - 1);
    const int src = x_index ? x_index[rc] : rc;
    const unsigned ooff = ((unsigned)(row0 + lrow) * D + cbase) * 4;
    {
      const unsigned soff = ((unsigned)src * D + cbase) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) xr[j] = ld4(x, soff + 32 * j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = lrow < nrows ? xr[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!ffn && lrow < nrows && MODE == 0) st4(a.c, ooff + 32 * j, v);  // centres = staged rows (layer 0 / no ResidualNorm)
      xr[j] = v;
      f16x4 h, lo;
      split4(v, h, lo);
      *reinterpret_cast<f16x4*>(sH + lrow * PLANE_STRIDE + cbase + 8 * j) = h;
      *reinterpret_cast<f16x4*>(sL + lrow * PLANE_STRIDE + cbase + 8 * j) = lo;
    }
    __syncthreads();
    f32x16 acc[RT];
    if (ffn) {
      // ResidualNorm (attention.py:37-40): h = swish(x W1 + b1)
      gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, pp->Wf2h, wave, lane, acc);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bv = *reinterpret_cast<const float4*>(&sPar[cbase + 8 * j]);
        const float4 pre = make_float4(fmaf(acc[0][4 * j], WINV, bv.x), fmaf(acc[0][4 * j + 1], WINV, bv.y),
                                       fmaf(acc[0][4 * j + 2], WINV, bv.z), fmaf(acc[0][4 * j + 3], WINV, bv.w));
        const float4 hh = f4swish(pre);
        f16x4 h, lo;
        split4(hh, h, lo);
        *reinterpret_cast<f16x4*>(sH + lrow * PLANE_STRIDE + cbase + 8 * j) = h;
        *reinterpret_cast<f16x4*>(sL + lrow * PLANE_STRIDE + cbase + 8 * j) = lo;
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
          f16x4 h, lo;
          split4(y, h, lo);
          *reinterpret_cast<f16x4*>(sH + lrow * PLANE_STRIDE + cbase + 8 * j) = h;
          *reinterpret_cast<f16x4*>(sL + lrow * PLANE_STRIDE + cbase + 8 * j) = lo;
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
          st4(oA, ooff + 32 * j, make_float4(fmaf(acc[0][4 * j], WINV, bg.x), fmaf(acc[0][4 * j + 1], WINV, bg.y),
                                             fmaf(acc[0][4 * j + 2], WINV, bg.z), fmaf(acc[0][4 * j + 3], WINV, bg.w)));
      }
      gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, WC, wave, lane, acc);
      // (P3 rows go to the group's atom-row cache in LDS: phase A gathers them by neighbour from there)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (lrow < nrows)
          *reinterpret_cast<float4*>(&sX[(row0 - g.atom_begin + lrow) * LDS_STRIDE + cbase + 8 * j]) =
              make_float4(acc[0][4 * j] * WINV, acc[0][4 * j + 1] * WINV, acc[0][4 * j + 2] * WINV, acc[0][4 * j + 3] * WINV);
      gemm_tile<false, RT>(sH, sL, whA, wlA, whB, wlB, nullptr, wave, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bq = *reinterpret_cast<const float4*>(&sPar[5 * D + cbase + 8 * j]);
        if (lrow < nrows)
          st4(oC, ooff + 32 * j, make_float4(fmaf(acc[0][4 * j], WINV, bq.x), fmaf(acc[0][4 * j + 1], WINV, bq.y),
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
        f16x4 h, lo;
        split4(z, h, lo);
        *reinterpret_cast<f16x4*>(sH + lrow * PLANE_STRIDE + cbase + 8 * j) = h;
        *reinterpret_cast<f16x4*>(sL + lrow * PLANE_STRIDE + cbase + 8 * j) = lo;
      }
      __syncthreads();
      gemm_tile<true, RT>(sH, sL, whA, wlA, whB, wlB, WD, wave, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bq = *reinterpret_cast<const float4*>(&sPar[5 * D + cbase + 8 * j]);
        if (lrow < nrows)
          st4(oC, ooff + 32 * j, make_float4(fmaf(acc[0][4 * j], WINV, bq.x), fmaf(acc[0][4 * j + 1], WINV, bq.y),
                                             fmaf(acc[0][4 * j + 2], WINV, bq.z), fmaf(acc[0][4 * j + 3], WINV, bq.w)));
      }
      gemm_tile<false, RT>(sH, sL, whA, wlA, whB, wlB, nullptr, wave, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bk = *reinterpret_cast<const float4*>(&sPar[6 * D + cbase + 8 * j]);
        if (lrow < nrows)
          st4(oB, ooff + 32 * j, make_float4(fmaf(acc[0][4 * j], WINV, bk.x), fmaf(acc[0][4 * j + 1], WINV, bk.y),
                                             fmaf(acc[0][4 * j + 2], WINV, bk.z), fmaf(acc[0][4 * j + 3], WINV, bk.w)));
      }
    }
    __syncthreads();  // the rows just stored are visible to the whole workgroup; planes / parameters may be overwritten
  }
}

template <int NT>
__global__ __launch_bounds__(256, NT <= 3 ? 2 : 1) void sr_kernel(SrArgs a) {
#pragma clang fp contract(off)  // fusions are written out (fmaf), as in edge_kernel: a row's result must not depend on its place
  __shared__ __attribute__((aligned(16))) unsigned char sTile[2 * TE_MAX * PLANE_STRIDE * 2];  // hi / lo planes; afterwards K as fp32
  __shared__ __attribute__((aligned(16))) float sQ[TQ * LDS_STRIDE];   // P1 rows (phase A) / query rows, then context (phase B)
  __shared__ __attribute__((aligned(16))) float sE[TE_MAX * NHEAD];    // LayerNorm partial statistics, then logits
  __shared__ __attribute__((aligned(16))) float sPar[7 * D];
  __shared__ int sOff[TQ + 1];
  // atom-row cache of the group: P3 rows during phase A, centre rows during phase B (both gathered by NEIGHBOUR atom)
  __shared__ __attribute__((aligned(16))) float sX[(NT <= 3 ? SR_ATOMS_SMALL : SR_ATOMS_BIG) * LDS_STRIDE];
  static_assert(sizeof(sTile) >= TE_MAX * LDS_STRIDE * sizeof(float), "K tile must fit the plane buffer");
  _Float16* const sH = reinterpret_cast<_Float16*>(sTile);
  _Float16* const sL = sH + TE_MAX * PLANE_STRIDE;
  float* const sK = reinterpret_cast<float*>(sTile);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5, cbase = 32 * wave + 4 * lh;
  const SrGroup g = a.groups[blockIdx.x];
  const int nt = g.n_tile;
  constexpr float WINV = 1.0f / WSCALE;

  // the group's geometry rows: tile p, row tile rt, columns cbase + 8 j .. + 3 of row lrow + 32 rt -- registers for the whole forward
  float4 greg[NT][2][4];

  // ---- geom0 = swish(Gauss(dist) Wd + bd) * swish(Gauss(weight) Ww + bw) (scann_model.py:378-389): basis_kernel's sequence ------
  {
    _Float16* const bH = reinterpret_cast<_Float16*>(sTile);
    _Float16* const bL = bH + TE_MAX * BASIS_STRIDE;
    f16x8 bdh[2], bdl[2], bwh[2], bwl[2];
    load_wsplit<2>(a.basis.Wdh, wave, lane, bdh, bdl);
    load_wsplit<2>(a.basis.Wwh, wave, lane, bwh, bwl);
#pragma unroll
    for (int p = 0; p < NT; ++p) {
      if (p < nt) {
        const EdgeTile tile = a.tiles[g.tile_begin + p];
        const int eb = tile.edge_begin, ne = tile.edge_end - eb, nem1 = ne > 0 ? ne - 1 : 0;
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
            greg[p][rt][j] = f4mul(sd, sw);
          }
        __syncthreads();  // every wave is done reading the basis planes
      }
    }
  }

  const int L = a.n_layer;
  for (int l = 0; l < L; ++l) {
    sr_atom_phase<0>(a, g, l, sH, sL, sE, sPar, sX, tid, lane, wave);
    const LayerParams* const P = a.layers + l;
    f16x8 wh[8], wl[8];

    // ---- phase A: geometry update (attention.py:141-153), W2 resident ------------------------------------------------------------
    load_wsplit<8>(P->W2h, wave, lane, wh, wl);
    sPar[tid] = (tid < D ? P->lng_g : P->lng_b)[tid & (D - 1)];
    sPar[2 * D + tid] = (tid < D ? P->ln_g : P->ln_b)[tid & (D - 1)];
    if (tid < D) sPar[4 * D + tid] = P->bk[tid];
#pragma unroll
    for (int p = 0; p < NT; ++p) {
      if (p < nt) {
        const EdgeTile tile = a.tiles[g.tile_begin + p];
        const int eb = tile.edge_begin, ne = tile.edge_end - eb, natom = tile.atom_end - tile.atom_begin;
        const int nem1 = ne > 0 ? ne - 1 : 0;
        const bool two = ne > 32;  // the second row tile holds edges
        int nbx[2];  // (neighbour atom's row in the cache, this lane's first column)
        int ctr[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int e = ne > 0 ? eb + min(lrow + 32 * rt, nem1) : 0;
          const int nb = ne > 0 ? a.edge_col[e] : g.atom_begin;
          nbx[rt] = (nb - g.atom_begin) * LDS_STRIDE + cbase;
          ctr[rt] = ne > 0 ? a.edge_row[e] - tile.atom_begin : 0;
        }
        {
          float4 p1reg[3];  // centre thirds P1 = c_i W1 + bg of the tile's atoms
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int idx = tid + 256 * i, la = min(idx >> 5, natom - 1), c4 = idx & 31;
            p1reg[i] = ld4(a.P1, ((unsigned)(tile.atom_begin + la) * 32 + c4) * 16);
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4*>(&sQ[(idx >> 5) * LDS_STRIDE + 4 * (idx & 31)]) = p1reg[i];
          }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int row = lrow + 32 * rt;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (row >= ne) greg[p][rt][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            f16x4 h, lo;
            split4(greg[p][rt][j], h, lo);
            *reinterpret_cast<f16x4*>(sH + row * PLANE_STRIDE + cbase + 8 * j) = h;
            *reinterpret_cast<f16x4*>(sL + row * PLANE_STRIDE + cbase + 8 * j) = lo;
          }
        }
        __syncthreads();  // planes, P1 rows (and, first tile, the layer's parameters) complete
        // T = swish(U + P1[i] + P3[j]) + G on the accumulators, one row tile at a time (T takes G's registers: G is dead once T
        // exists); LayerNorm_g partial statistics
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          if (rt == 0 || two) {
            f32x16 acc;
            mma_slab(sH + rt * 32 * PLANE_STRIDE, sL + rt * 32 * PLANE_STRIDE, wh, wl, lane, acc);
            const float* p1 = sQ + ctr[rt] * LDS_STRIDE + cbase;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 p1v = *reinterpret_cast<const float4*>(p1 + 8 * j);
              const float4 p3v = *reinterpret_cast<const float4*>(&sX[nbx[rt] + 8 * j]);  // neighbour third P3[j] = c_j W3
              float4 v;
              v.x = fmaf(acc[4 * j], WINV, p1v.x) + p3v.x;
              v.y = fmaf(acc[4 * j + 1], WINV, p1v.y) + p3v.y;
              v.z = fmaf(acc[4 * j + 2], WINV, p1v.z) + p3v.z;
              v.w = fmaf(acc[4 * j + 3], WINV, p1v.w) + p3v.w;
              const float4 t = f4add(f4swish(v), greg[p][rt][j]);
              greg[p][rt][j] = t;
              s += f4sum(t);
            }
            const float mean32 = xor32(s) * (1.0f / 32.0f);
            float v2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float d;
              d = greg[p][rt][j].x - mean32; v2 = fmaf(d, d, v2);
              d = greg[p][rt][j].y - mean32; v2 = fmaf(d, d, v2);
              d = greg[p][rt][j].z - mean32; v2 = fmaf(d, d, v2);
              d = greg[p][rt][j].w - mean32; v2 = fmaf(d, d, v2);
            }
            const float m2 = xor32(v2);
            if (lh == 0) *reinterpret_cast<float2*>(&sE[((lrow + 32 * rt) * 4 + wave) * 2]) = make_float2(mean32, m2);
          }
        __syncthreads();  // statistics complete; every wave is done with the G planes and the P1 rows
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          if (rt == 0 || two) {
            const int row = lrow + 32 * rt;
            const float4 sa = *reinterpret_cast<const float4*>(&sE[row * 8]), sb = *reinterpret_cast<const float4*>(&sE[row * 8 + 4]);
            const float mean = ((sa.x + sa.z) + (sb.x + sb.z)) * 0.25f;
            const float d0 = sa.x - mean, d1 = sa.z - mean, d2 = sb.x - mean, d3 = sb.z - mean;
            const float var = (((sa.y + sa.w) + (sb.y + sb.w)) + 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3))) * (1.0f / D);
            const float rstd = 1.0f / sqrtf(var + 1e-6f);
            if (!(var < RANGE_FINITE) && row < ne) flag_range(a.range_flag, 1, l);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 gm = *reinterpret_cast<const float4*>(&sPar[cbase + 8 * j]);
              const float4 be = *reinterpret_cast<const float4*>(&sPar[D + cbase + 8 * j]);
              const float4 t = greg[p][rt][j];
              float4 y;
              float inv;
              inv = rstd * gm.x; y.x = fmaf(t.x, inv, be.x - mean * inv);
              inv = rstd * gm.y; y.y = fmaf(t.y, inv, be.y - mean * inv);
              inv = rstd * gm.z; y.z = fmaf(t.z, inv, be.z - mean * inv);
              inv = rstd * gm.w; y.w = fmaf(t.w, inv, be.w - mean * inv);
              greg[p][rt][j] = y;  // geom' (scann_model.py:415): the next layer's input, never stored
            }
          }
        __builtin_amdgcn_sched_barrier(0);  // one tile at a time (register pressure)
      }
    }

    // ---- phase B: gate, keys, softmax, context (attention.py:136,157-214), Wk resident ---------------------------------------------
    load_wsplit<8>(P->Wkh, wave, lane, wh, wl);
    {  // centre rows of the group -> atom-row cache (every wave is past its last read of the P3 rows: the barrier above)
      const int nrow4 = (g.atom_end - g.atom_begin) * 32;
      for (int i = tid; i < nrow4; i += 256)
        *reinterpret_cast<float4*>(&sX[(i >> 5) * LDS_STRIDE + 4 * (i & 31)]) = ld4(a.c, ((unsigned)g.atom_begin * 32 + i) * 16);
    }
#pragma unroll
    for (int p = 0; p < NT; ++p) {
      if (p < nt) {
        const EdgeTile tile = a.tiles[g.tile_begin + p];
        const int eb = tile.edge_begin, ne = tile.edge_end - eb, natom = tile.atom_end - tile.atom_begin;
        const int nem1 = ne > 0 ? ne - 1 : 0;
        const bool two = ne > 32;
        int nbx[2];
        int ctr[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int e = ne > 0 ? eb + min(lrow + 32 * rt, nem1) : 0;
          const int nb = ne > 0 ? a.edge_col[e] : g.atom_begin;
          nbx[rt] = (nb - g.atom_begin) * LDS_STRIDE + cbase;
          ctr[rt] = ne > 0 ? a.edge_row[e] - tile.atom_begin : 0;
        }
        const int voff = a.edge_offset[tile.atom_begin + min(tid, natom)];
        const int qa = tid >> 5;
        const unsigned qoff = (tid & 31) * 16;
        const float4 q0 = ld4(a.q, qoff + (unsigned)(tile.atom_begin + min(qa, natom - 1)) * (D * 4));
        const float4 q1 = ld4(a.q, qoff + (unsigned)(tile.atom_begin + min(qa + 8, natom - 1)) * (D * 4));
        const float4 q2 = ld4(a.q, qoff + (unsigned)(tile.atom_begin + min(qa + 16, natom - 1)) * (D * 4));
        if (p == 0) __syncthreads();  // the centre rows are in the cache
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int row = lrow + 32 * rt;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 ang = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rt == 0 || two) ang = f4mul(*reinterpret_cast<const float4*>(&sX[nbx[rt] + 8 * j]), greg[p][rt][j]);  // c[j] * geom' (attention.py:136,157)
            if (row >= ne) ang = make_float4(0.f, 0.f, 0.f, 0.f);
            f16x4 h, lo;
            split4(ang, h, lo);
            *reinterpret_cast<f16x4*>(sH + row * PLANE_STRIDE + cbase + 8 * j) = h;
            *reinterpret_cast<f16x4*>(sL + row * PLANE_STRIDE + cbase + 8 * j) = lo;
          }
        }
        if (tid <= natom) sOff[tid] = voff - eb;
        *reinterpret_cast<float4*>(&sQ[qa * LDS_STRIDE + 4 * (tid & 31)]) = q0;
        *reinterpret_cast<float4*>(&sQ[(qa + 8) * LDS_STRIDE + 4 * (tid & 31)]) = q1;
        *reinterpret_cast<float4*>(&sQ[(qa + 16) * LDS_STRIDE + 4 * (tid & 31)]) = q2;
        __syncthreads();  // ang planes and query rows complete
        f32x16 acc[2];
        mma_slab(sH, sL, wh, wl, lane, acc[0]);
        if (two) mma_slab(sH + 32 * PLANE_STRIDE, sL + 32 * PLANE_STRIDE, wh, wl, lane, acc[1]);
        // logits e[n, h] = (q[i, h, :] * 16^-0.5) . K[n, h, :] (attention.py:180-183) from the accumulators
        float lg[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          if (rt == 0 || two) {
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
        // softmax over each atom's edges + context + unscaled-query residual (attention.py:186-212); edge_kernel's loop
        {
          const int lgp = tid >> 5, c4 = tid & 31, h = c4 >> 2;
          for (int la = lgp; la < natom; la += 8) {
            const int e0 = sOff[la], e1 = sOff[la + 1];
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
            float4* qp = reinterpret_cast<float4*>(&sQ[la * LDS_STRIDE + 4 * c4]);
            const float4 q4 = *qp;
            *qp = make_float4(fmaf(cx.x, rs, q4.x), fmaf(cx.y, rs, q4.y), fmaf(cx.z, rs, q4.z), fmaf(cx.w, rs, q4.w));
          }
        }
        __syncthreads();
        // LayerNorm of the context rows (attention.py:214): 8 threads per atom row
        {
          const int rr = tid >> 3, sb = tid & 7;
          if (rr < natom) {
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
              st4(a.ctx, ((unsigned)(tile.atom_begin + rr) * 32 + c4) * 16, y);
            }
          }
        }
        __syncthreads();  // context rows stored (visible to the workgroup); sQ / sOff / the planes may be overwritten
      }
    }
  }
  // readout's atom side: ResidualNorm of the last layer, after_Lc, GlobalAttention query / key (scann_model.py:424, attention.py:269-272)
  sr_atom_phase<2>(a, g, L, sH, sL, sE, sPar, sX, tid, lane, wave);
}

void launch_struct(const SrArgs& a, int nt_max, hipStream_t s) {
  if (a.n_group <= 0) return;
  const dim3 grid(a.n_group), block(256);
  if (nt_max <= 3) hipLaunchKernelGGL((sr_kernel<3>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((sr_kernel<6>), grid, block, 0, s, a);
}

}  // namespace scann
