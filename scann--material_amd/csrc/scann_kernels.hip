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
#include <algorithm>
#include <cstdlib>
#include "scann_internal.h"
#include "scann_mma.h"

namespace scann {

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
// KEEP: the training forward's instantiation (Dropout, keep_* stores); never exact
// (workgroups per CU: three at 64 rows, four at 32 -- except where the register allocator could not be kept below the limit without
// spilling and occupancy is not what bounds the launch: the exact-fp32 re-run at 64 rows (1/16 of the matrix rate), the base branch's
// query-only kernel at 32 rows; tests/test_host.py reads scratch sizes and register counts from the built library)
template <bool FFN, int MODE, int RT, bool EX = false, bool KEEP = false>
__global__ __launch_bounds__(256, RT == 2 ? (EX ? 2 : 3) : (MODE == 1 ? 3 : 4)) void atom_kernel(AtomArgs a) {
#pragma clang fp contract(off)  // fusions are written out: both row-tile copies of a formula round alike (see edge_kernel)
  static_assert(!(EX && KEEP), "the training forward runs the split-fp16 kernels");
  constexpr int TAR = 32 * RT;  // atom rows per tile
  __shared__ __attribute__((aligned(16))) unsigned char sTile[2 * TAR * PLANE_STRIDE * 2];  // hi / lo planes of the current GEMM input
  __shared__ __attribute__((aligned(16))) float sRed[TAR * 8];  // LayerNorm partial statistics [wave][row][mean, m2]
#ifdef SCANN_DIAG_SE_ROWMAJOR  // A/B build: round 5's [row][wave][2]
#define SR_STAT(w, r) (((r) * 4 + (w)) * 2)
#else
#define SR_STAT(w, r) (((w) * TAR + (r)) * 2)
#endif
  __shared__ __attribute__((aligned(16))) float sPar[7 * D];   // bf1 | bf2 | lnr_g | lnr_b | bA | bC | bD
#define SCANN_ATOM_BIX blockIdx.x
#include "scann_atom_body.inc"
#undef SR_STAT
#undef SCANN_ATOM_BIX
}

void launch_atom(const AtomArgs& a, hipStream_t s) {
  if (a.n_atom <= 0) return;
  // 32-row tiles (<= 128 VGPRs, 22 KB of LDS: four workgroups per CU = 1,024 slots) while they all fit ONE round of workgroups: the
  // launch is then the latency chain of a tile, and a 32-row tile's chain is shorter (one batch per launch: 490 k -> 568 k
  // molecules/s, training step 1.14 -> 1.09 ms).  Beyond that 64-row tiles (half the weight traffic per row).
  // (re-measured in round 5 at the driver's 10-batch shape, 23 k atoms: 64-row tiles -- half the weight traffic, 360 tiles -- 0.159-0.170 ms
  //  of atom launches per forward against 0.152-0.157 ms for the 720 32-row tiles: the threshold stays)
  const int rows = a.n_atom <= 32 * 1024 ? 32 : 64;
  const dim3 grid((a.n_atom + rows - 1) / rows), block(256);
  const bool keep = !a.exact && (a.drop_p > 0.f || a.keep_pre1 || a.keep_T2 || a.keep_preA);  // training forward
#define SCANN_ATOM_CASE(F, M)                                                                               \
  do {                                                                                                      \
    if (a.exact) {                                                                                          \
      if (rows == 32) hipLaunchKernelGGL((atom_kernel<F, M, 1, true>), grid, block, 0, s, a);               \
      else hipLaunchKernelGGL((atom_kernel<F, M, 2, true>), grid, block, 0, s, a);                          \
    } else if (keep) {                                                                                      \
      if (rows == 32) hipLaunchKernelGGL((atom_kernel<F, M, 1, false, true>), grid, block, 0, s, a);        \
      else hipLaunchKernelGGL((atom_kernel<F, M, 2, false, true>), grid, block, 0, s, a);                   \
    } else if (rows == 32) hipLaunchKernelGGL((atom_kernel<F, M, 1>), grid, block, 0, s, a);                \
    else hipLaunchKernelGGL((atom_kernel<F, M, 2>), grid, block, 0, s, a);                                  \
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
// KEEP: the training forward's instantiation -- the keep_* stores and the attention dropout exist only there (as uniform branches in
// the inference kernels they cut every epilogue into basic blocks of one LDS read -> wait -> arithmetic -> store each).
// DEAD: the last layer of an inference forward -- nobody reads geom' (scann_model.py:415-421 threads it to the NEXT layer only).
template <bool GUPD, int RT, bool FB = false, bool EX = false, bool KEEP = false, bool DEAD = false>
__global__ __launch_bounds__(256, RT == 2 ? 3 : 4) void edge_kernel(EdgeArgs a) {
#pragma clang fp contract(off)  // fusions are written out (fmaf): both unrolled row-tile copies of a formula must round alike,
                                // so that a row's result does not depend on where in a tile it lands (batch-composition invariance)
  static_assert(!FB || GUPD, "the fused basis exists for the g_update kernel only");
  static_assert(!(FB && EX), "the exact fallback runs the plain first layer");
  static_assert(!KEEP || (!FB && !EX && !DEAD), "the training forward runs the plain split-fp16 kernel and keeps every layer's geometry");
  static_assert(!DEAD || GUPD, "the base branch stores no geometry");
  constexpr int TEK = 32 * RT;  // edge rows per tile: 64 (three workgroups per CU) or, for launches of one round, 32 (four)
  // piece-major geometry tiles (scann_edge_body.inc): the inference kernels, whose geometry nobody else reads.  The training / debug
  // forward (KEEP) and the exact re-run (EX: its first layer reads basis_kernel's row-major geom0) keep [n_edge,128] rows.
  constexpr bool BLK = GUPD && !EX && !KEEP;
  // hi / lo planes of the A operand (G or the basis, then ang): 2 x 64 x 272 B; afterwards K as fp32 [64][LDS_STRIDE]
  __shared__ __attribute__((aligned(16))) unsigned char sTile[2 * TEK * PLANE_STRIDE * 2];
  __shared__ __attribute__((aligned(16))) float sQ[TQ * LDS_STRIDE];  // P1 rows, then query rows of the tile's atoms, then context
  // LayerNorm_g partial statistics [wave][row] (mean, m2), then logits [head][row] with one float of padding per head: lanes that one LDS
  // cycle serves hold consecutive rows, so the row is the fastest index of both (scann_edge_body.inc)
  __shared__ __attribute__((aligned(16))) float sE[NHEAD * (TEK + 1)];
#ifdef SCANN_DIAG_SE_ROWMAJOR  // A/B build: round 5's [row][wave][2] / [row][head] layouts
#define SE_STAT(w, r) (((r) * 4 + (w)) * 2)
#define SE_LOGIT(r, h) ((r) * NHEAD + (h))
#else
#define SE_STAT(w, r) (((w) * TEK + (r)) * 2)
#define SE_LOGIT(r, h) ((h) * (TEK + 1) + (r))
#endif
  __shared__ __attribute__((aligned(16))) float sPar[5 * D];          // layer_norm_g gamma/beta (base: filter bias), layer_norm gamma/beta, key bias
  __shared__ int sOff[TQ + 1];
#ifdef SCANN_DIAG_OCC2  // diagnostic: two workgroups per CU instead of three (how much of the time is latency hidden by occupancy?)
  __shared__ float sDummy[7 * 1024];
  if (a.n_tile < 0) sDummy[threadIdx.x] = 1.f;
#endif
  static_assert(2 * TEK * PLANE_STRIDE * 2 >= TEK * LDS_STRIDE * (int)sizeof(float), "K tile must fit the plane buffer");
#define SCANN_EDGE_TIX (a.xcd_remap ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x)
#include "scann_edge_body.inc"
#undef SCANN_EDGE_TIX
#undef SE_STAT
#undef SE_LOGIT
#ifdef SCANN_DIAG_TAILWAIT  // diagnostic: what the layer launch's producer side costs an edge tile (its stores acknowledged before it ends)
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
#endif
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
  const float rstd = ln_rstd((sRed[1][0] + sRed[1][1]) * (1.0f / D));
  ctx[(size_t)atom * D + c] = ln_apply(t, rstd, -mean * rstd, ln_g[c], ln_b[c]);
}

void launch_edge_merge(const int32_t* big_tab, int n_big, const float* part_buf, const float* q, const float* ln_g,
                       const float* ln_b, float* ctx, int32_t* range_flag, int layer, hipStream_t s) {
  if (n_big > 0)
    hipLaunchKernelGGL(edge_merge_kernel, dim3(n_big), dim3(128), 0, s, big_tab, part_buf, q, ln_g, ln_b, ctx, range_flag, layer);
}

// centre atom of every edge: one thread per atom writes its CSR row's entries (a batch has ~8 edges per atom)
// (zero_word: a device word this launch clears on its way -- the flag of the pack_padded_kernel launch behind it)
__global__ void edge_row_kernel(const int32_t* __restrict__ edge_offset, int n_atom, int32_t* __restrict__ edge_row, int32_t* __restrict__ zero_word) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a == 0 && zero_word) *zero_word = 0;
  if (a >= n_atom) return;
  for (int e = edge_offset[a]; e < edge_offset[a + 1]; ++e) edge_row[e] = a;
}
void launch_edge_row(const int32_t* edge_offset, int n_atom, int32_t* edge_row, hipStream_t s, int32_t* zero_word) {
  if (n_atom > 0 || zero_word)
    hipLaunchKernelGGL(edge_row_kernel, dim3(std::max(1, (n_atom + 255) / 256)), dim3(256), 0, s, edge_offset, n_atom, edge_row, zero_word);
}

// Padded -> CSR (what gather_shape + the masks express: custom_layers.py:18-28, datagenerator.py:69-135): one thread per padded atom
// slot.  A real atom walks its N neighbour slots in slot order and writes the unmasked ones behind edge_offset[row] -- the neighbour's
// intra-structure index becomes its packed row through row_of -- HBM-bound streaming work (13 bytes read, 12 written per kept slot;
// 160 B per atom slot read in all), a few tens of microseconds per 2,048-structure chunk.
__global__ __launch_bounds__(256) void pack_padded_kernel(PackPaddedArgs a) {
  const int bm = blockIdx.x * blockDim.x + threadIdx.x;
  if (bm >= a.B * a.M) return;
  const int row = a.row_of[bm];
  if (row < 0) return;
  const int z = a.atomic[bm];
  int bad = (z < 0 || z >= a.n_species) ? 2 : 0;
  a.out_atomic[row] = bad ? 0 : z;  // (a valid table row either way: the forward must not fault on an input that is about to be refused)
  const int b0 = (bm / a.M) * a.M;  // first atom slot of this structure
  int e = a.edge_offset[row];
  const size_t base = (size_t)bm * a.N;
  for (int n = 0; n < a.N; ++n) {
    const bool on = a.mask_size == 1 ? static_cast<const uint8_t*>(a.neighbor_mask)[base + n] != 0
                                     : (static_cast<const uint32_t*>(a.neighbor_mask)[base + n] & 0x7fffffffu) != 0;
    if (!on) continue;
    const int tgt = a.neighbors[base + n];
    const int r = (tgt >= 0 && tgt < a.M) ? a.row_of[b0 + tgt] : -1;
    if (r < 0) bad |= 1;
    a.out_col[e] = r < 0 ? row : r;
    a.out_dist[e] = a.dist[base + n];
    a.out_weight[e] = a.weight[base + n];
    ++e;
  }
  if (bad) atomicOr(a.flag, bad);
}
void launch_pack_padded(const PackPaddedArgs& a, hipStream_t s) {
  const int n = a.B * a.M;
  if (n > 0) hipLaunchKernelGGL(pack_padded_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a);
}

void launch_edge(const EdgeArgs& a, hipStream_t s) {
  if (a.n_tile <= 0) return;
  const dim3 grid(a.n_tile), block(256);
  // tile_rows is the height the batch's tile plan was made for (scann_batch_upload: 32 for launches of one round of workgroups)
  // three families: EX (exact-fp32 re-run of an inference forward), KEEP (training / debug forwards: row-major geometry, keep_* stores,
  // attention dropout), and the inference kernels proper (piece-major geometry tiles; FB = the first layer, DEAD = the last)
  const bool keep = !a.exact && (a.geom_rows || a.keep_V || a.keep_T || a.keep_ang || a.keep_K || a.attn_drop_p > 0.f);
  const bool dead = a.g_update && a.geom_dead && !keep;
#define SCANN_EDGE_GO(...) hipLaunchKernelGGL((edge_kernel<__VA_ARGS__>), grid, block, 0, s, a)
#define SCANN_EDGE_ROWS(G, ...)                                             \
  do {                                                                      \
    if (a.tile_rows == 32) SCANN_EDGE_GO(G, 1, __VA_ARGS__);                \
    else SCANN_EDGE_GO(G, 2, __VA_ARGS__);                                  \
  } while (0)
  if (a.exact) {  // never with the fused first layer (run_forward launches basis_kernel for it)
    if (!a.g_update) SCANN_EDGE_ROWS(false, false, true, false, false);
    else if (dead) SCANN_EDGE_ROWS(true, false, true, false, true);
    else SCANN_EDGE_ROWS(true, false, true, false, false);
  } else if (keep) {
    if (a.g_update) SCANN_EDGE_ROWS(true, false, false, true, false);
    else SCANN_EDGE_ROWS(false, false, false, true, false);
  } else if (!a.g_update) {
    SCANN_EDGE_ROWS(false, false, false, false, false);
  } else if (a.fuse_basis) {  // (a one-layer model's only launch stores its geometry: no instantiation for that corner)
    SCANN_EDGE_ROWS(true, true, false, false, false);
  } else {
    if (dead) SCANN_EDGE_ROWS(true, false, false, false, true);
    else SCANN_EDGE_ROWS(true, false, false, false, false);
  }
#undef SCANN_EDGE_ROWS
#undef SCANN_EDGE_GO
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

// (DPP / lane-swap forms of the xor butterflies: the same additions and maxima without six LDS round trips each -- in the readout
//  they sit one behind the other: norm -> maximum -> sum of the exponentials -> head)
__device__ __forceinline__ float wave_sum(float v) { return wave_sum64(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_max64(v); }

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
      const float v = sum32(part[i]);  // over the 32 query columns of this lane's half
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
    // (four key rows requested per step -- the terms past the last atom are a valid row times an exact zero, added in the order of the
    //  one-row-per-step loop: each step was a memory round trip of its own)
    for (int i = half; i < n; i += 8) {
      float kv[4], av[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int iu = i + 2 * u;
        kv[u] = a.gk[(size_t)(a0 + min(iu, n - 1)) * D + f];
        av[u] = iu < n ? sAgg[iu] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) rsum += av[u] * kv[u];
    }
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
    for (int k0 = 0; k0 < 64; k0 += 16) {  // sixteen weights requested per step: four memory round trips instead of sixteen
      float wv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) wv[u] = wcol[(size_t)(k0 + u) * D];
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        h0 = fmaf(rp[k0 + u], wv[u], h0);
        h1 = fmaf(rp[k0 + u + 1], wv[u + 1], h1);
        h2 = fmaf(rp[k0 + u + 2], wv[u + 2], h2);
        h3 = fmaf(rp[k0 + u + 3], wv[u + 3], h3);
      }
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

