// Internal declarations shared by the HIP kernels (scann_kernels.hip) and the C-ABI runtime
// (scann_runtime.cpp).  Not part of the public boundary (include/scann_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace scann {

constexpr int D = 128;           // local_dim = global_dim = dense_out (all shipped configs)
constexpr int NHEAD = 8;         // num_head
constexpr int HDIM = D / NHEAD;  // 16
constexpr int NG = 20;           // Gaussian basis size (scann_model.py:378)
constexpr int LDS_STRIDE = 132;  // floats per staged row: 128 + 4 pad (conflict-free b128 A-fragment reads)
constexpr int TE_MAX = 64;       // edge rows per edge tile (two 32-row MFMA row tiles)
constexpr int TA = 64;           // atom rows per atom tile (two 32-row MFMA row tiles)
constexpr int TQ = 24;           // atoms per EDGE tile (the query-row buffer of edge_kernel)
constexpr int WPACK = D * D;     // floats in one packed 128x128 weight (fp32 fragment order, or its split-fp16 image: same bytes)
constexpr float WSCALE = 256.f;  // split-fp16 weights are stored times 2^8 (their lo parts stay normal fp16 numbers); exact inverse in the epilogues
constexpr float WMAX = 65504.f / WSCALE;  // largest weight magnitude the fp16 hi part can hold

// Inverted dropout keyed by (seed, tag, element index): the backward pass regenerates the identical mask.
// Returns 0 (dropped) or 1/(1-p).
__host__ __device__ inline float drop_scale(unsigned long long seed, unsigned tag, size_t idx, float p) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1) + ((unsigned long long)tag << 48);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  return u < p ? 0.f : 1.0f / (1.0f - p);
}
#ifdef __HIPCC__
// GlobalAttention score of atom i of a structure (attention.py:279-292): sum over j != i of gk_i . gq_j, exact products, fp64 sums, four
// interleaved partial sums per dot product (shared by the forward and by the pooling backward, which must see the SAME scores)
__device__ __forceinline__ double gen_pool_score(const float* __restrict__ gq, const float* __restrict__ gk, int a0, int n, int i, int dg) {
  const float* ki = gk + (size_t)(a0 + i) * dg;
  double agg = 0.0;
  for (int j = 0; j < n; ++j) {
    if (j == i) continue;
    const float* qj = gq + (size_t)(a0 + j) * dg;
    double e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;
    int k = 0;
    for (; k + 4 <= dg; k += 4) {
      e0 = fma((double)ki[k], (double)qj[k], e0);
      e1 = fma((double)ki[k + 1], (double)qj[k + 1], e1);
      e2 = fma((double)ki[k + 2], (double)qj[k + 2], e2);
      e3 = fma((double)ki[k + 3], (double)qj[k + 3], e3);
    }
    for (; k < dg; ++k) e0 = fma((double)ki[k], (double)qj[k], e0);
    agg += (e0 + e1) + (e2 + e3);
  }
  return agg;
}
#endif

// Range guard of the split-fp16 projections (|activation| and |weight| * 2^8 must stay below 65504, the largest fp16): an operand
// that overflows becomes inf in its hi part, the product NaN, and the NaN reaches the statistics of the next LayerNorm.  The
// kernels therefore test every LayerNorm variance (and the one activation that no LayerNorm follows) for finiteness and, on
// failure, store (site << 8 | layer + 1) in a host-pinned word that the entry points returning results check after their
// synchronisation (SCANN_ERR_RANGE).  Sites: 1 layer_norm_g, 2 layer_norm (context), 3 ResidualNorm, 4 after_Lc activation,
// 5 a weight after an optimiser step.
__device__ __forceinline__ void flag_range(int32_t* flag, int site, int layer) {
  if (flag && *reinterpret_cast<volatile int32_t*>(flag) == 0)  // the first report stands: kernels run in layer order on their stream
    *flag = (site << 8) | (layer + 1);
}
constexpr float RANGE_FINITE = 3.0e38f;  // !(x < RANGE_FINITE): inf or NaN

constexpr unsigned DROP_TAG_ATTN = 2000;   // + layer: Dropout(0.05) on attention weights, element = edge * 8 + head
constexpr unsigned DROP_TAG_EMBED = 1000;  // Dropout(0.1) after dense_embed (scann_model.py:374); ResidualNorm l uses tag l

// One tile of the edge kernel: a run of whole atoms whose CSR rows are contiguous, <= TE edges.
struct EdgeTile {
  int32_t atom_begin, atom_end;
  int32_t edge_begin, edge_end;
};

// Device pointers of one LocalAttention(+ResidualNorm) iteration.  128x128 kernels are stored in
// the MFMA fragment order produced by pack_weight() (see scann_kernels.hip: gemm128).
struct LayerParams {
  // atom-tile kernel
  const float *bg;              // filter_geo bias
  const float *bq;              // query bias
  const _Float16 *W1h, *W3h, *Wqh, *Wf1h, *Wf2h;  // split-fp16 images (pack_weight_f16): filter_geo rows [0,128) (centre) and [256,384)
                                                  // (neighbour), query, ResidualNorm dense_1 / dense_2
  // edge-tile kernel
  const float *bk;              // key bias
  const _Float16 *W2h, *Wkh;    // filter_geo rows [128,256) (geometry) and key as split-fp16 images: what edge_kernel multiplies with
  const _Float16 *Wfh;          // base branch: filter_geo [20,128] zero-padded to K = 32, split-fp16 image
  const float *lng_g, *lng_b;   // layer_norm_g
  const float *ln_g, *ln_b;     // layer_norm
  const float *Wfg, *bfg;       // base branch (g_update False): filter_geo [20,128] raw + bias (backward of the basis filter)
  // ResidualNorm that follows this LocalAttention (applied at the head of the next atom kernel)
  const float *bf1, *bf2, *lnr_g, *lnr_b;
  // fp32 fragment-order images (pack_weight) of the seven 128x128 kernels: the exact-fp32 fallback of the forward (EX kernels)
  const float *W1p, *W2p, *W3p, *Wqp, *Wkp, *Wf1p, *Wf2p;
};

struct HeadParams {
  const float *Wap, *ba;                // after_Lc
  const float *Wgqp, *bgq, *Wgkp, *bgk; // global_attention query / key
  const _Float16 *Wah, *Wgqh, *Wgkh;    // split-fp16 images (atom_kernel mode 2)
  const float *Wb, *bb;                 // bf_property (row-major [128,128])
  const float *wo, *bo;                 // predict_property [128], [1]
};

struct BasisParams {
  const float *Wd, *bd, *Ww, *bw;  // neighbor_d / neighbor_w [20,128]
  const float *cd, *cw;            // Gaussian centres (20 each)
  const _Float16 *Wdh, *Wwh;       // split-fp16 images of the two kernels, K padded to 32 (basis_kernel)
};

// ---- launch wrappers (defined in scann_kernels.hip) -------------------------------------------

// swish(E . W + b) per species -> lut[n_species,128]   (Embedding + dense_embed folded at load time)
void launch_embed_lut(const float* emb, const float* W, const float* b, int n_species, int emb_dim,
                      float* lut, hipStream_t s);

// geom0[e,:] = swish(G(dist) Wd + bd) * swish(G(weight) Ww + bw)         (g_update)
// or gd[e, 0:20] = G(dist) (base branch: the per-layer filter consumes the raw basis)
void launch_basis(const BasisParams& p, const float* dist, const float* weight, int n_edge,
                  float* geom, hipStream_t s);
void launch_basis_raw(const float* cd, const float* dist, int n_edge, float* gd, hipStream_t s);

// General embedding (use_ring / feature="cgcnn"); the plain Embedding path uses the per-species LUT instead.
struct EmbedArgs {
  int32_t n_atom, emb_dim;
  const int32_t* atomic;   // [n_atom] (feature="atomic")
  const float* cgcnn;      // [n_atom,92] or null
  const float* ring;       // [n_atom,2] or null
  const float *emb;        // embed_atom/embeddings [n_atoms,emb]
  const float *We, *be;    // embed_atom/kernel [92,emb], bias (cgcnn)
  const float *Wr, *br;    // extra_embed [2,10], [10]
  const float *Wde, *bde;  // dense_embed [emb(+10),128], [128]
  float* c0;               // [n_atom,128]
};
void launch_embed(const EmbedArgs& a, hipStream_t s);

struct AtomArgs {
  const float* x;          // [n_atom,128] input rows (context of previous layer, or the LUT)
  const int32_t* x_index;  // optional row indirection (layer 0: atomic number -> LUT row) or null
  int32_t n_atom;
  // ResidualNorm (ffn != 0): c = LN(x + W2 swish(W1 x + b1) + b2)
  int32_t ffn;
  const _Float16 *Wf1h, *Wf2h;   // split-fp16 images (pack_weight_f16)
  const float *bf1, *bf2, *lnr_g, *lnr_b;
  float* c;                // [n_atom,128] centres out (always written)
  // training only: Dropout(0.1) on the staged rows (layer 0, scann_model.py:374) or on the ResidualNorm branch
  // (attention.py:29); drop_p == 0 in inference
  float drop_p;
  uint32_t drop_tag;
  unsigned long long drop_seed;
  // projections
  int32_t mode;            // 0: P1,P3,q (g_update)  1: q only (base)  2: readout (after_Lc -> gq, gk)
  const _Float16 *WAh, *WBh, *WCh, *WDh;  // split-fp16 images: mode 0: W1, W3, Wq | mode 1: -, -, Wq | mode 2: after_Lc, -, ga query, ga key
  const float *bA;         // mode 0: bg        | mode 2: after_Lc bias
  const float *bC;         // mode 0/1: bq      | mode 2: ga query bias
  const float *bD;         // mode 2: ga key bias
  float *oA, *oB, *oC;     // mode 0: P1,P3,q | mode 1: -, -, q | mode 2: -, gk, gq
  // training forward: ResidualNorm intermediates the backward would otherwise recompute (null in inference), [n_atom,128]:
  // pre1 = x W1 + b1, H1 = swish(pre1), T2 = x + drop(H1 W2 + b2) (the LayerNorm input)
  float *keep_pre1, *keep_H1, *keep_T2;
  float *keep_preA, *keep_z;   // mode 2, training forward: after_Lc pre-activation and swish output [n_atom,128]
  unsigned long long* stamps;  // diagnostic build (-DSCANN_STAMPS, env SCANN_STAMP_ATOM=1) only: [n_tiles,16] phase clocks
  int32_t* range_flag;         // host-pinned range-guard word (flag_range) or null
  int32_t layer;               // layer whose projections this launch computes (for the range-guard message)
  // exact-fp32 projections (EX instantiation): Wf1h, Wf2h, WAh .. WDh then point at fp32 fragment-order images (pack_weight)
  int32_t exact;
};
void launch_atom(const AtomArgs& a, hipStream_t s);

struct EdgeArgs {
  const EdgeTile* tiles;       // whole atoms per tile: <= TE_MAX edges and <= TQ atoms
  int32_t n_tile;
  int32_t xcd_remap;           // contiguous run of tiles per XCD
  // atoms with more than 64 neighbours: their edges are cut into chunk tiles of one atom each; tile_part[tile] = partial slot
  // of a chunk tile, -1 for ordinary tiles (null: no such atom in the batch); a chunk tile leaves (running max, sum,
  // unnormalised context) per column in part_buf[slot][3][128] for edge_merge_kernel
  const int32_t* tile_part;
  float* part_buf;
  int32_t g_update;
  const int32_t* edge_offset;  // [n_atom+1]
  const int32_t* edge_col;     // [n_edge]
  const int32_t* edge_row;     // [n_edge] centre atom of each edge
  float* geom;                 // [n_edge,128] in/out (g_update)
  // training forward: per-edge tensors the backward would otherwise recompute (null in inference):
  // V = G.W2 + P1[i] + P3[j], T = swish(V) + G (LayerNorm_g input), ang = c[j] * geom', K = ang.Wk + bk -- all [n_edge,128]
  float *keep_V, *keep_T, *keep_ang, *keep_K;
  float* geom_out;             // where geom' goes (null: in place) -- keep-mode writes each layer's geometry to its own slice
  int32_t tile_rows;           // 32 | 64: edge rows per tile of this batch's plan (selects the kernel instantiation)
  int32_t n_edge;              // edges of the batch: geom has ONE SPARE ROW behind them (scann_batch_upload), the dump row of edge-less tiles
  int32_t geom_rows;           // the geometry tensors are [n_edge,128] ROWS (training, debug and exact forwards: someone else reads them); 0: piece-major tiles
  int32_t geom_dead;           // last layer of an inference forward: nobody reads geom' (scann_model.py:415-421 threads it to the NEXT layer only)
  const float* gd;             // [n_edge,20] raw distance basis (base)
  const float* edge_weight;    // [n_edge] (base; g_update with fuse_basis: the second input of the basis MLP)
  // first layer of an inference forward (g_update, 64-row tiles): the tile's geometry rows are COMPUTED from (dist, weight) in the
  // prologue -- basis_kernel's arithmetic on the tile's rows -- instead of being written by basis_kernel and read back here
  int32_t fuse_basis;
  // ... and, with the per-species embedding table (feature = "atomic", no ring input), the layer's atom rows c, P1, P3, q are
  // functions of the atomic number alone: they are [n_species,128] tables computed when the weights change, the kernel indexes them
  // through species[atom], and the first atom launch of the forward does not happen at all
  const int32_t* species;      // [n_atom] atomic numbers, or null: c, P1, P3, q are per-atom arrays
  const float* dist;           // [n_edge]
  BasisParams basis;
  const float *c, *P1, *P3, *q;  // [n_atom,128]
  float* ctx;                  // [n_atom,128] out: LayerNorm(context)
  // training with use_drop: Dropout(0.05) on the attention weights (attention.py:116,191); 0 in inference
  float attn_drop_p;
  uint32_t attn_drop_tag;
  unsigned long long attn_drop_seed;
  unsigned long long* stamps;  // diagnostic build (-DSCANN_STAMPS) only: [n_tile,16] phase clocks, else null
  int32_t* range_flag;         // host-pinned range-guard word (flag_range) or null
  int32_t layer;
  int32_t exact;               // exact-fp32 projections (EX instantiation): p.W2h, p.Wkh then point at fp32 fragment-order images
  LayerParams p;
};
void launch_edge(const EdgeArgs& a, hipStream_t s);
// softmax merge of the chunk tiles of every big atom (+ unscaled-query residual + LayerNorm, attention.py:189-214)
void launch_edge_merge(const int32_t* big_tab, int n_big, const float* part_buf, const float* q, const float* ln_g,
                       const float* ln_b, float* ctx, int32_t* range_flag, int layer, hipStream_t s);

// ---- generic-width forward (scann_generic.hip): plain fp32 kernels for local_dim / num_head / global_dim / dense_out other than 128 / 8 ----
struct GenSeg {
  const float* p;      // [*, w] rows
  const int32_t* idx;  // row r of the operand reads p[idx[r]] (null: p[r])
  int32_t w;
};
struct GenDenseArgs {
  GenSeg seg[3];
  int32_t n_seg;  // x = concat of the segments' rows ...
  int32_t prod;   // ... or (prod) the elementwise product of segments 0 and 1, both K wide
  const float* W;  // [K, N] row-major (the Keras kernel)
  const float* b;  // [N] or null
  int32_t K, N, rows;
  int32_t act;  // 0 none, 1 swish
  const float* res;        // added after the activation: res[(res_idx ? res_idx[r] : r)][o]
  const int32_t* res_idx;
  const float* row_scale;  // [rows] or null: multiplied last
  float* Y;                // [rows, N]
  // training forward (zero / null in inference): the pre-activation kept for the backward, Dropout on the layer's output
  float* pre;              // [rows, N] or null
  float drop_p;
  uint32_t drop_tag;
  unsigned long long drop_seed;
};
void launch_gen_dense(const GenDenseArgs& a, hipStream_t s);
void launch_gen_layernorm(const float* X, const float* res, const float* gamma, const float* beta, int rows, int N, float* Y, hipStream_t s);
void launch_gen_gauss(const float* x, const float* centres, int n, float* out, hipStream_t s);
void launch_gen_mul(const float* a, const float* b, size_t n, float* out, hipStream_t s);
void launch_gen_attn(const float* q, const float* K, const int32_t* edge_offset, int n_atom, int d, int H, int max_degree, float* ctx, hipStream_t s,
                     float drop_p = 0.f, unsigned drop_tag = 0, unsigned long long drop_seed = 0);
// rep_out non-null (training forward): the pooled rows [n_struct, dg] are stored and the property head is NOT evaluated
void launch_gen_readout(const int32_t* mol_offset, int n_struct, int max_atoms, const float* gq, const float* gk, int dg, int dout, int use_ga_norm,
                        int relu_out, const float* Wb, const float* bb, const float* wo, const float* bo, float* ga_attn, float* y, hipStream_t s,
                        float* rep_out = nullptr);

// ---- generic-width training (scann_generic_train.hip) ----
struct GenTransDesc {  // WT[dst + o * kn + kk] = W[src + (k0 + kk) * N + o]: rows [k0, k0 + kn) of a row-major [*, N] kernel, transposed
  int64_t src, dst;
  int32_t k0, kn, N;
};
struct GenDwArgs {
  GenSeg seg[3];  // the forward operand X, assembled as in GenDenseArgs
  int32_t n_seg, prod;
  const float* dZ;  // [rows, N] gradient of the layer's pre-activation
  int32_t K, N, rows;
  float *dW, *db;   // [K, N], [N] (null: no bias): ACCUMULATED into
  float* part;      // gen_dw_part_floats(rows, K, N) floats of per-slab partial tiles
  int32_t slab_rows;  // (set by the launcher)
};
int gen_dw_slabs(int rows, int K, int N);
size_t gen_dw_part_floats(int rows, int K, int N);
void launch_gen_transpose(const GenTransDesc* descs, int n_desc, int max_elems, const float* W, float* WT, hipStream_t s);
void launch_gen_act_bwd(const float* dY, const float* pre, const float* row_scale, int rows, int N, float drop_p, unsigned drop_tag,
                        unsigned long long drop_seed, float* out, hipStream_t s);
void launch_gen_mul_gather(const float* a, const float* B, const int32_t* idx, const float* c, int rows, int N, float* out, hipStream_t s);
void launch_gen_relu(float* y, int n, hipStream_t s);
void launch_gen_dense_dw(const GenDwArgs& a, hipStream_t s);
int gen_ln_chunks(int rows);  // row chunks of the gamma / beta sums: `part` holds gen_ln_chunks(rows) * 2 * N floats
void launch_gen_layernorm_bwd(const float* X, const float* res, const float* gamma, const float* dY, int rows, int N, float* dX, float* stats,
                              float* part, float* dgamma, float* dbeta, hipStream_t s);
void launch_gen_attn_bwd(const float* q, const float* K, const int32_t* edge_offset, int n_atom, int d, int H, int max_degree, const float* dctx,
                         float drop_p, unsigned drop_tag, unsigned long long drop_seed, float* dq, float* dK, hipStream_t s);
void launch_gen_pool_bwd(const int32_t* mol_offset, int n_struct, int max_atoms, const float* gq, const float* gk, int dg, int use_ga_norm,
                         const float* drep, float* dgq, float* dgk, hipStream_t s);
void launch_gen_edge_to_atom(const int32_t* edge_offset, const int32_t* in_off, const int32_t* in_edge, const float* S_out, const float* S_in,
                             const float* P_a, const float* P_b, const float* acc, int n_atom, int d, float* out, hipStream_t s);
// part: ceil(n_atom / 64) * n_species * emb floats of per-chunk sums
void launch_gen_table_grad(const int32_t* atomic, int n_atom, const float* dV, int emb, int n_species, float* part, float* dTable, hipStream_t s);

struct ReadoutArgs {
  const int32_t* mol_offset;  // [n_struct+1]
  int32_t n_struct;
  int32_t max_atoms;          // largest structure (sizes the dynamic LDS score buffer)
  const float *gq, *gk;       // [n_atom,128]
  int32_t use_ga_norm, relu_out;
  HeadParams p;
  float* ga_attn;             // [n_atom]
  float* y;                   // [n_struct]
};
void launch_readout(const ReadoutArgs& a, hipStream_t s);

// Host-side edge-tile plan of a packed batch (scann_pack.cpp): whole atoms per tile, <= tile_rows edges and <= tile_atoms
// atoms; atoms with more than TE_MAX neighbours become chunk tiles when allow_chunks.  Validates the CSR arrays and, with
// fill_edge_row, fills edge_row (centre atom of each edge).  Returns SCANN_OK or a negative status with `err` set.
int plan_tiles(const int32_t* mol_offset, int32_t B, const int32_t* edge_offset, const int32_t* edge_col, int32_t A, int32_t E,
               int tile_rows_req, int tile_atoms, bool allow_chunks, std::vector<EdgeTile>& tiles, std::vector<int32_t>& tile_part,
               std::vector<int32_t>& big_tab, std::vector<int32_t>& edge_row, int* tile_rows_out, int32_t* max_degree,
               int32_t* n_slot, std::string& err, bool fill_edge_row = true);
// edge_row[e] = centre atom of edge e from the CSR offsets, on the device (scann_batch_upload: behind the input copy)
void launch_edge_row(const int32_t* edge_offset, int n_atom, int32_t* edge_row, hipStream_t s, int32_t* zero_word = nullptr);

// Padded Keras input -> packed CSR on the device (scann_upload_padded): the payload arrays of the padded dict as they came over the bus.
struct PackPaddedArgs {
  int32_t B, M, N, n_species;
  const int32_t* row_of;        // [B*M] packed row of every padded atom slot, -1 = padded (host: scann_count_padded)
  const int32_t* edge_offset;   // [n_atom+1]
  const int32_t* atomic;        // [B*M]
  const int32_t* neighbors;     // [B*M*N] intra-structure atom index
  const void* neighbor_mask;    // [B*M*N] 1- or 4-byte elements
  int32_t mask_size;
  const float *weight, *dist;   // [B*M*N]
  int32_t* out_atomic;          // [n_atom]
  int32_t* out_col;             // [n_edge] global atom row of the neighbour
  float *out_dist, *out_weight; // [n_edge]
  int32_t* flag;                // |= 1: an unmasked slot points at a padded atom, |= 2: atomic number outside the embedding table
};
void launch_pack_padded(const PackPaddedArgs& a, hipStream_t s);

// Host-side permutation of a row-major [128,128] (in,out) kernel into MFMA fragment order.
void pack_weight(const float* W, int ld, float* Wp);
// Split-fp16 image of rows [0, k_real) of a row-major [*,128] kernel for v_mfma_f32_32x32x16_f16 (edge_kernel: mma_split):
// [wave 4][k-step ks][plane hi|lo][lane 64][8 halfs], element j of lane l = fp16 part of WSCALE * W[16 s + 8 (l >> 5) + j][32 w + (l & 31)]
// (zero for k >= k_real).  4 * ks * 2 * 64 * 8 halfs = ks * 2048 floats' worth of bytes.
void pack_weight_f16(const float* W, int ld, int k_real, int ks, uint16_t* out);

}  // namespace scann
