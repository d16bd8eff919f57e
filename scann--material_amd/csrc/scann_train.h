// Declarations of the training-path kernels (scann_train.hip).  Internal.
#pragma once
#include "scann_internal.h"

#include <vector>

namespace scann {

struct ReadoutBwdArgs {
  const int32_t* mol_offset;
  int32_t n_struct, max_atoms, use_ga_norm;
  const float *gq, *gk, *ga;   // forward tensors [n_atom,128], [n_atom]
  const float* dy;             // [n_struct] d loss / d y
  const float *Wb, *bb, *wo;   // bf_property (row-major), predict_property
  float *dgq, *dgk;            // [n_atom,128] out
  float *rep_out, *dpre_out;   // [n_struct,128] out (inputs of the bf_property weight gradient)
  float *dwo, *dbo;            // dwo: [n_struct,128] slots (reserve_vec); dbo: the scalar bias gradient (atomic)
};

// One region of the device weight arena, regenerated from the flat master parameters after each optimiser step.
struct RepackDesc {
  int64_t src;        // element offset into the master (canonical) parameter vector
  int64_t dst;        // element offset into the weight arena
  int32_t transpose;  // packed image of W^T instead of W (backward dX GEMMs)
  int32_t raw;        // > 0: plain copy of this many elements (<= 16384); 0: 128x128 fp32 fragment-order pack;
                      // -1: split-fp16 image of a [128,128] kernel, -2: of a [20,128] kernel padded to K = 32 (pack_weight_f16)
};

// Weight gradients are bit-reproducible: every launch_wgrad* stores per-slab partial sums into slots of `ctx.arena` (bump
// allocated, never reused within a step) and records (destination, slots); wgrad_flush adds them in slab order (one launch per
// WGRAD_REDUCE_MAX tensors; the backward pass flushes once per layer, on the side stream).  wgrad_slabs(rows) slots of 128*128 (+128 with a bias) floats per gradient.
struct WgradReduceEntry {
  float* dst;
  const float* part;
  int32_t n_slab, numel;
};
constexpr int WGRAD_MAX_JOBS = 8;
constexpr int WGRAD_REDUCE_MAX = 32;  // gradient tensors per reduce launch (a layer has <= 19; the last layer + the readout 29)
struct WgradCtx {
  struct Job {
    const float *X, *dY;
    float *part, *bpart;
    int32_t rows, chunks;
    const int32_t* nb = nullptr;  // gated operand: row r of the X operand is X[nb[r]] * X2[r] (ang = c[j] * G', attention.py:157,
    const float* X2 = nullptr;    // which the training forward then does not have to keep)
  };
  std::vector<Job> jobs;                  // queued by wgrad_add, launched together by wgrad_launch
  float* arena = nullptr;                 // partial slots (device)
  size_t off = 0;                         // floats used
  std::vector<WgradReduceEntry> entries;  // one per gradient tensor since the last wgrad_flush
};
int wgrad_slabs(int rows);
void wgrad_add(WgradCtx& ctx, const float* X, const float* dY, float* dW, float* db, int rows, const int32_t* nb = nullptr,
               const float* X2 = nullptr);
void wgrad_launch(WgradCtx& ctx, hipStream_t s);
void wgrad_flush(WgradCtx& ctx, hipStream_t s);

void launch_linear(const float* X, const float* Wp, const float* bias, float* Y, float* P, int rows, int flags, hipStream_t s);

// Y (+)= X0.W0 + X1.W1 + X2.W2 (packed [128,128] kernels; null X1 / X2: fewer terms)
// swish_pre (optional): the result is multiplied by swish'(swish_pre) before it is stored (fused swish backward)
void launch_linear_sum(const float* X0, const float* W0, const float* X1, const float* W1, const float* X2, const float* W2, float* Y,
                       int rows, int accumulate, hipStream_t s, const float* swish_pre = nullptr);
void launch_swish_bwd(const float* pre, const float* dout, float* dpre, size_t n, hipStream_t s);
void launch_dropout(float* x, size_t n, unsigned long long seed, unsigned tag, float p, hipStream_t s);
int ln_bwd_slots(int rows);
int attn_bwd_slots(int n_atom, int max_degree);
void launch_ln_bwd(WgradCtx& ctx, const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, int rows,
                   int accumulate, hipStream_t s);
// dcn[e] = dang[e] * g[e] (per edge; summed per neighbour atom with launch_gather_sum), dg_tot = dang * c[nb] (+ dg_in)
void launch_edge_dang(const float* c, const int* nb, const float* g, const float* dang, const float* dg_in, float* dcn,
                      float* dg_tot, int n_edge, hipStream_t s);
// out[a] (+)= sum of val[e] over the edges whose neighbour is atom a (reverse adjacency in_off / in_edge); fixed order, no atomics
void launch_gather_sum(const float* val, const int* in_off, const int* in_edge, float* out, int n_atom, int accumulate, hipStream_t s);
void launch_ln_bwd_edge(WgradCtx& ctx, const float* T, const float* gamma, const float* dang, const float* c, const int* nb,
                        const float* dg_in, const float* V, float* dT, float* dV, float* dgamma, float* dbeta, int rows, hipStream_t s);
void launch_gather_prod_sum(const float* x, const float* y, const int* in_off, const int* in_edge, float* out, int n_atom, int accumulate,
                            hipStream_t s);
void launch_atom_sums(const float* dV, const int* edge_offset, const int* in_off, const int* in_edge, float* own, float* in, int n_atom,
                      hipStream_t s);
void launch_dropout_copy(float* dst, const float* src, size_t n, unsigned long long seed, unsigned tag, float p, hipStream_t s);
void launch_attn_bwd(WgradCtx& ctx, const float* q, const float* K, const int* edge_offset, const float* dctx, const float* gamma, float* dq,
                     float* dK, float* dgamma, float* dbeta, int n_atom, int max_degree, float drop_p, unsigned drop_tag,
                     unsigned long long drop_seed, hipStream_t s);
// ---- fused backward chains (scann_train_fused.hip) ----
float* reserve_vec(WgradCtx& ctx, float* dst, int n_slot);  // records (dst, slots) of a gradient vector a kernel stores per workgroup
int tile_slots(int rows);                                   // workgroups (= gamma / beta slots) of a fused kernel over `rows`
struct RnBwdArgs {
  const float *dC, *T2, *pre1, *gamma;          // [n_atom,128] x3 (dC may be null: zeros), layer_norm gamma
  const _Float16 *Wf2Th, *Wf1Th;                // split-fp16 images of dense_2^T, dense_1^T
  float *dY, *dpre1, *dCtx;                     // out [n_atom,128]
  float *dgamma, *dbeta;                        // per-workgroup slots (set by the launcher)
  float* dC_out;                                // optional: the complete d loss / d c' (null: not stored)
  int32_t n_atom;
  float drop_p;
  unsigned long long drop_seed;
  unsigned drop_tag;
  int32_t n_pre;                                // 0..3 projections of the layer above added to dC first: dC += X[t] . W[t]^T
  const float* X[3];
  const _Float16* Wh[3];
};
struct EdgeBwdArgs {
  const float *dK, *c, *dG_in, *T, *V, *gamma;  // [n_edge,128] (c: [n_atom,128]; dG_in may be null), layer_norm_g gamma
  const float* G;                               // T == null: T is recomputed as swish(V) + G from the geometry entering the layer
  const int* nb;                                // [n_edge] neighbour atom row
  const _Float16 *WkTh, *W2Th;                  // split-fp16 images of key^T, filter_geo (geometry third)^T
  float *dang, *dV, *dG;                        // out [n_edge,128]
  float *dgamma, *dbeta;
  int32_t n_edge;
};
// the attention backward of the same layer (attn_bwd16_kernel's arithmetic), run by edge_bwd_kernel<1, true> on the tile's own atoms
// ahead of its own chain: the forward's 32-row tile plan holds whole atoms, so the dK rows a tile needs are the ones its atoms produce
struct AttnPart {
  const float *q, *K, *dctx, *gamma;  // [n_atom,128], [n_edge,128], [n_atom,128], layer_norm gamma
  const int32_t* edge_offset;
  const EdgeTile* tiles;              // the batch's 32-row tile plan (whole atoms, contiguous edges)
  float *dq, *dK;                     // out [n_atom,128], [n_edge,128] (dK still goes to memory: the key weight gradient reads it)
  float *dgamma, *dbeta;              // layer_norm gamma / beta slots, one per workgroup
  float drop_p;
  uint32_t drop_tag;
  unsigned long long drop_seed;
};
void launch_rn_bwd(WgradCtx& ctx, RnBwdArgs a, float* dgamma, float* dbeta, hipStream_t s);
void launch_edge_bwd(WgradCtx& ctx, EdgeBwdArgs a, float* dgamma, float* dbeta, hipStream_t s);
// attention backward + edge_bwd in one launch over the n_tile tiles of the plan (32-row tiles, every degree <= 16)
void launch_attn_edge_bwd(WgradCtx& ctx, EdgeBwdArgs a, AttnPart b, int n_tile, float* dgamma_g, float* dbeta_g, float* dgamma_ln,
                          float* dbeta_ln, hipStream_t s, hipEvent_t done = nullptr);
// (`done`: an event recorded by the kernel's own completion signal, for a side stream to wait on -- see launch_attn_edge_bwd)
void launch_atom_gather3(const float* dang, const float* G, const float* dV, const int* edge_offset, const int* in_off, const int* in_edge,
                         float* dC, float* dP1, float* dP3, int n_atom, hipStream_t s, hipEvent_t done = nullptr);
void launch_readout_bwd(const ReadoutBwdArgs& a, hipStream_t s);
void launch_basis_bwd(const BasisParams& p, const float* dist, const float* weight, const float* dgeom, int n_edge, float* dWd,
                      float* dbd, float* dWw, float* dbw, hipStream_t s);
void launch_base_geom_bwd(const float* gd, const float* Wf, const float* bf, const float* wgt, const float* dgeomL, int n_edge,
                          float* dWf, float* dbf, hipStream_t s);
void launch_embed_bwd(const float* dc0, const int* atomic, int n_atom, const float* emb, const float* W, const float* b,
                      float* dlut, int n_species, int emb_dim, float* dEmb, float* dW, float* db, unsigned long long drop_seed,
                      unsigned drop_tag, float drop_p, hipStream_t s);
void launch_embed_general_bwd(const EmbedArgs& a, const float* dc0, float* dEmb, float* dWe, float* dbe, float* dWr, float* dbr,
                              float* dWde, float* dbde, hipStream_t s);
void launch_sse(const float* y, const float* t, int n, double* out, float* t_dev, float* dy, double* host_stat, hipStream_t s);
void launch_dy(const float* y, const float* t, int n, float scale, const double* stat, float* dy, hipStream_t s);
void launch_adam(float* w, float* g, float* m, float* v, const float* l2mask, size_t n, float lr_hat, float b1, float b2,
                 float eps, float l2, int zero_g, hipStream_t s);
void launch_repack(const RepackDesc* descs, int n, const float* master, float* arena, int32_t* range_flag, hipStream_t s);

}  // namespace scann
