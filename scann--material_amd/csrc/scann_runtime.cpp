// C-ABI runtime of libscann_hip.so (include/scann_hip.h): handle / weight container / packed-batch
// residency / launch schedule of the forward graph built by the reference's create_model
// (scann_model.py:329-453).  Host side only -- all arithmetic is in scann_kernels.hip.
#include "../../include/scann_hip.h"
#include "scann_internal.h"
#include "scann_train.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace scann;

namespace {

constexpr int MAX_STREAM = 16;

// Per-device cache of freed device blocks: resident batches, their keep/debug buffers and training workspaces are
// allocated per batch, and pipelines (predict_dataset, the training loop) create and drop one per step -- reusing a
// block of similar size avoids a hipMalloc/hipFree pair (hundreds of microseconds, and an implicit device sync) each time.
struct BlockCache {
  std::multimap<size_t, void*> free_blocks;
  std::map<void*, size_t> size_of;
  size_t cached_bytes = 0;
};
std::map<int, BlockCache> g_block_cache;
std::mutex g_block_mu;
constexpr size_t BLOCK_CACHE_LIMIT = (size_t)16 << 30;  // per device

hipError_t cached_malloc(void** p, size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t gran = bytes >= ((size_t)1 << 20) ? ((size_t)1 << 20) : ((size_t)64 << 10);
  const size_t want = (std::max<size_t>(bytes, 1) + gran - 1) / gran * gran;
  {
    std::lock_guard<std::mutex> lk(g_block_mu);
    BlockCache& c = g_block_cache[dev];
    auto it = c.free_blocks.lower_bound(want);
    if (it != c.free_blocks.end() && it->first <= 2 * want) {
      *p = it->second;
      c.cached_bytes -= it->first;
      c.free_blocks.erase(it);
      return hipSuccess;
    }
  }
  const hipError_t e = hipMalloc(p, want);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(g_block_mu);
    g_block_cache[dev].size_of[*p] = want;
  }
  return e;
}

// The caller guarantees no kernel still uses the block (scann_batch_free synchronises first).
void cached_free(void* p) {
  if (!p) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_block_mu);
  BlockCache& c = g_block_cache[dev];
  auto it = c.size_of.find(p);
  if (it == c.size_of.end()) {
    (void)hipFree(p);
    return;
  }
  if (c.cached_bytes + it->second > BLOCK_CACHE_LIMIT) {
    c.size_of.erase(it);
    (void)hipFree(p);
    return;
  }
  c.free_blocks.emplace(it->second, p);
  c.cached_bytes += it->second;
}

void cache_release(int dev) {  // at handle destruction: give the idle blocks of this device back
  std::lock_guard<std::mutex> lk(g_block_mu);
  BlockCache& c = g_block_cache[dev];
  for (auto& kv : c.free_blocks) {
    c.size_of.erase(kv.second);
    (void)hipFree(kv.second);
  }
  c.free_blocks.clear();
  c.cached_bytes = 0;
}
thread_local std::string g_create_error;

struct WeightSpec {
  std::string name;
  int64_t rows, cols;  // cols == 0: vector of length rows
  int64_t numel() const { return cols ? rows * cols : rows; }
};

}  // namespace

namespace {
struct GenKeep;
}

struct scann_handle {
  scann_config_t cfg{};
  int device = 0;
  std::string err;
  hipStream_t streams[MAX_STREAM]{};
  int nstream = 2;  // HIP streams batches are spread over (env SCANN_STREAMS, 1..16).  Two launch groups in flight fill each other's
                    // latency-bound launches; with the upload off the launching thread more only split the caches (1.83 M vs 1.71 M
                    // molecules/s host-inclusive at 4, tools/e2e_size.py)
  std::vector<WeightSpec> specs;
  bool loaded = false;
  bool debug = false;
  int tile_atoms = TQ;     // atoms per edge tile the tile builder allows (edge_kernel's query-row buffer)
  int n_cu = 256;      // compute units of the device
  int time_every = 0;  // > 0: sample edge-kernel launch durations on every n-th forward (scann_edge_timing)
  int64_t time_count = 0;
  std::vector<hipEvent_t> time_ev;  // pairs (start, stop)
  std::vector<int> time_edges;
  int xcd_remap = 1;   // env SCANN_XCD_REMAP=0 disables the XCD-contiguous tile order
  int fuse_basis = 1;  // env SCANN_FUSE_BASIS=0: basis_kernel writes geom0 and layer 0 reads it, as in training (A/B switch)
  int species_tables = 1;  // env SCANN_SPECIES_TABLES=0: the first layer's atom rows come from an atom launch, not from per-species tables
  bool generic = false;        // widths other than 128 / 8: the plain-fp32 kernels of scann_generic.hip / scann_generic_train.hip
  float* g_weights = nullptr;  // generic: the flat fp32 parameter vector on the device (spec order, spec_off offsets)
  float* g_centres = nullptr;  // generic: 20 + 20 Gaussian centres (distance, Voronoi weight)
  std::map<std::string, int64_t> g_off;  // generic: tensor name -> offset in g_weights
  // generic-width training: W^T images of the kernels (refreshed at the head of every backward), one descriptor per transposed block
  std::vector<GenTransDesc> gt_descs;
  GenTransDesc* d_gt_descs = nullptr;
  std::map<std::string, int64_t> gt_off;  // "<tensor name>#<block>" -> offset in g_WT
  float* g_WT = nullptr;
  int gt_max = 0;                         // elements of the largest block
  GenKeep* gen_keep = nullptr;            // inside a training forward: where run_forward_generic keeps its tensors
  bool weights_exact = false;  // a loaded 128x128 kernel has |w| >= 255.9: the split-fp16 images cannot hold it, inference runs exact
  bool force_exact = false;    // env SCANN_EXACT=1: every inference forward on the exact-fp32 kernels (test / diagnosis switch)
  bool strict_range = false;   // env SCANN_STRICT_RANGE=1: SCANN_ERR_RANGE instead of the exact-fp32 re-run of an inference forward
  int64_t exact_reruns = 0;    // forwards re-run on the exact-fp32 kernels so far (scann_exact_reruns)
  float* d_weights = nullptr;  // one arena with every device-side weight image
  std::vector<LayerParams> layers;
  HeadParams head{};
  BasisParams basis{};
  const float* lut = nullptr;  // [n_atoms,128] swish(Embedding . dense_embed)
  // per-species rows of the first layer (feature = "atomic" without ring): P1 = lut W1 + bg, P3 = lut W3, q = lut Wq + bq of layer 0
  // [n_atoms,128] each and a copy of the centres; recomputed on the next inference forward after the weights changed (sp_dirty)
  float *sp_c = nullptr, *sp_P1 = nullptr, *sp_P3 = nullptr, *sp_q = nullptr;
  bool sp_dirty = true;
  const float* cd = nullptr;   // distance Gaussian centres
  EmbedArgs embed{};           // weight pointers of the general embedding path (use_ring / cgcnn)
  // canonical (spec-order) flat parameter vector and how the device arena is derived from it
  std::vector<float> host_master;
  std::vector<int64_t> spec_off;
  std::vector<RepackDesc> descs;
  size_t arena_floats = 0, o_lut = 0, o_emb = 0, o_Wde = 0, o_bde = 0;
  struct LayerT {
    const float *W1T, *W2T, *W3T, *WqT, *WkT, *Wf1T, *Wf2T;                  // fp32 fragment order (modular backward)
    const _Float16 *W1Th, *W2Th, *W3Th, *WqTh, *WkTh, *Wf1Th, *Wf2Th;        // split-fp16 images (fused backward kernels)
  };
  std::vector<LayerT> layersT;  // packed transposes for the backward dX GEMMs
  const float *WaT = nullptr, *WgqT = nullptr, *WgkT = nullptr;
  const _Float16* WaTh = nullptr;  // split-fp16 image of after_Lc^T (folded into the first rn_bwd_kernel)
  // training state (scann_train_begin)
  float *t_master = nullptr, *t_grad = nullptr, *t_m = nullptr, *t_v = nullptr, *t_l2 = nullptr;
  RepackDesc* t_descs = nullptr;
  int64_t t_step = 0;
  float train_drop_p = 0.f;            // > 0 only inside scann_train_forward
  float attn_drop_p = 0.f;             // use_drop: Dropout(0.05) on attention weights (scann_set_attention_dropout)
  bool in_train_forward = false;
  unsigned long long train_seed = 0;
  ncclComm_t comm = nullptr;
  // scann_train_step_begin / _end: up to two steps may be enqueued before the first is ended (the host prepares step k + 1 while the
  // device runs step k); slot = step number & 1.  Slot 2 of the targets belongs to the synchronous scann_train_forward.
  double* h_stat = nullptr;              // pinned [2][4]: {sse, count, sum |y - t|} of the step in that slot
  float* h_targets[3] = {nullptr, nullptr, nullptr};  // pinned staging of a step's targets (read by the loss kernel directly)
  size_t h_targets_cap[3] = {0, 0, 0};
  hipEvent_t step_ev[2] = {nullptr, nullptr};         // recorded at the end of the step in that slot
  int64_t step_begun = 0, step_ended = 0;
  bool grads_zeroed = false;             // the gradient vector is known to be all zeros (Adam of scann_train_step leaves it so)
  // scann_batch_upload: pinned staging buffers (a ring, grow-only) copied to the device on a stream of their own -- the call returns
  // when the copy is ENQUEUED; the batch's first launches wait for it through the batch's event
  struct Stage { char* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
  static constexpr int N_STAGE = 8;
  Stage stage[N_STAGE];
  int stage_next = 0;
  // results come back through one pinned block per stream slot (one D2H for y and the GlobalAttention scores, which sit next to each
  // other in the batch arena) instead of two staged copies into the caller's pageable arrays
  struct PadScratch { std::vector<int32_t> gidx, at, mol, eoff, col; std::vector<float> dist, wgt, ga; };  // scann_forward_padded
  PadScratch pad_scratch;
  struct DlStage { char* p = nullptr; size_t cap = 0; };
  DlStage dl_stage[MAX_STREAM];
  hipStream_t copy_stream = nullptr;
  hipStream_t train_aux2 = nullptr;      // second side stream: the basis-MLP gradients beside the embedding chain
  bool train_fused = true;               // fused backward chains (scann_train_fused.hip); SCANN_TRAIN_FUSED=0: modular kernels
  bool train_aux_borrowed = false;       // train_aux is streams[1] (not destroyed separately)
  hipStream_t train_aux = nullptr;       // side stream of the backward pass: weight-gradient GEMMs run beside the data-gradient chain
  std::vector<hipEvent_t> train_ev;      // ring of fork / join events between the two streams
  // reusable scratch of the synchronous scann_forward path (grow-only device arena + pinned host staging)
  char* sc_arena = nullptr;
  size_t sc_cap = 0;
  // scann_forward_padded: the padded payload goes over the bus BEFORE the host reads the masks (pinned + device block, grow-only)
  char* pp_host = nullptr;
  char* pp_dev = nullptr;
  size_t pp_cap = 0;
  char* sc_host = nullptr;
  size_t sc_host_cap = 0;
  struct scann_dbatch* sc_db = nullptr;
  int comm_world = 1;
  int32_t* range_flag = nullptr;  // host-pinned [MAX_STREAM], one word per stream slot (training: slot 0), written by the kernels'
                                  // range guard (flag_range), read after that stream's synchronisation
};

struct scann_dbatch {
  int32_t n_struct = 0, n_atom = 0, n_edge = 0, n_tile = 0, max_atoms = 0, tile_rows = 64, max_degree = 0, tile_atoms = TQ;
  char* arena = nullptr;  // inputs + workspace, one allocation
  // inputs
  int32_t *atomic = nullptr, *mol_offset = nullptr, *edge_offset = nullptr, *edge_col = nullptr, *edge_row = nullptr;
  float *dist = nullptr, *weight = nullptr, *ring = nullptr, *cgcnn = nullptr, *c0 = nullptr;
  EdgeTile* tiles = nullptr;
  float *keep_q = nullptr, *keep_V = nullptr, *keep_T = nullptr, *keep_ang = nullptr, *keep_K = nullptr;  // [L][rows,128], training forward (owned by the train workspace)
  float *keep_pre1 = nullptr, *keep_H1 = nullptr, *keep_T2 = nullptr;  // ResidualNorm intermediates, [L][n_atom,128]
  float *keep_preA = nullptr, *keep_z = nullptr;  // after_Lc pre-activation / output [n_atom,128]
  bool kept = false;  // the last training forward filled them
  hipEvent_t busy_ev = nullptr;  // end of the last scann_train_step that used the batch (scann_batch_release)
  bool idle = false;             // nothing enqueued on the batch since its last scann_batch_download returned (scann_batch_release)
  int32_t *in_off = nullptr, *in_edge = nullptr;  // reverse adjacency: edges sorted by their neighbour atom (backward pass)
  bool has_rev = false;          // in_off / in_edge are filled (uploads of a handle in training mode; else built on first backward)
  hipEvent_t upload_ev = nullptr;  // end of the asynchronous input copy (scann_batch_upload); null: the copy was synchronous
  bool upload_done = false;        // ... and it has been seen complete: launches on the batch no longer wait for it (wait_upload)
  int32_t* tile_part = nullptr;  // per tile: partial slot of a chunk tile or -1 (null without big atoms)
  int32_t* big_tab = nullptr;    // per atom with > 64 neighbours: atom row, first slot, number of slots
  float* part_buf = nullptr;     // [n_slot][3][128] softmax state of the chunk tiles
  int32_t n_big = 0, n_slot = 0;
  int32_t* pack_flag = nullptr;  // device packing (scann_upload_padded): what pack_padded_kernel found wrong with the input, behind y
  size_t gen_ws_bytes = 0;
  char* gen_ws = nullptr;  // generic-width forward: its per-batch workspace (sized by the handle's widths; cached_malloc)
  // workspace
  float *geom = nullptr, *gd = nullptr, *c = nullptr, *ctx = nullptr, *P1 = nullptr, *P3 = nullptr, *q = nullptr;
  float *gq = nullptr, *gk = nullptr, *ga = nullptr, *y = nullptr;
  // debug copies (allocated on demand)
  float *dbg_c = nullptr, *dbg_g = nullptr, *dbg_ctx = nullptr;
  unsigned long long* stamps = nullptr;  // diagnostic build only
  int n_stamp = 0;
  int dbg_layers = -1;
  int last_slot = 0;
  bool owns_arena = true;  // false: the arena belongs to the handle's scratch (scann_forward)
};

static void free_train_ws(scann_dbatch* db);

namespace {

// The inputs' copy runs on the copy stream (scann_batch_upload returns when it is ENQUEUED): the first launches on the batch wait for
// its event.  A resident batch is launched on again and again; once the event has been seen complete the wait -- a barrier packet
// that costs the stream ~5 us even when it has nothing to wait for -- is left out.
hipError_t wait_upload(scann_dbatch* db, hipStream_t s) {
  if (!db->upload_ev || db->upload_done) return hipSuccess;
  if (hipEventQuery(db->upload_ev) == hipSuccess) {
    db->upload_done = true;
    return hipSuccess;
  }
  (void)hipGetLastError();  // (hipErrorNotReady is an answer, not an error)
  return hipStreamWaitEvent(s, db->upload_ev, 0);
}

int fail(scann_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  else g_create_error = msg;
  return code;
}

// After a synchronisation: did a kernel of the finished work trip the range guard (flag_range)?  The word is cleared, so the handle
// stays usable once the caller has dealt with the cause.
// One word per stream slot: with two launch groups in flight, the download of the group on stream A must not consume (and clear) a
// report raised by the kernels of the group on stream B.
int check_range(scann_handle* h, const char* where, int slot = 0) {
  if (!h->range_flag) return SCANN_OK;
  const int32_t code = *reinterpret_cast<volatile int32_t*>(h->range_flag + slot);
  if (!code) return SCANN_OK;
  h->range_flag[slot] = 0;
  const int site = code >> 8, layer = (code & 0xff) - 1;
  static const char* const names[] = {"?", "layer_norm_g statistics (geometry update)", "layer_norm statistics (attention context)",
                                      "ResidualNorm statistics", "after_Lc activation", "a weight after the optimiser step"};
  std::string m = std::string(where) + ": value outside the range of the split-fp16 projections (|activation| < 65504, |weight| < 255.9): " +
                  (site >= 1 && site <= 5 ? names[site] : names[0]);
  if (site != 5) m += layer >= h->cfg.n_attention ? ", readout" : ", local_attention_" + std::to_string(layer);
  m += "; the results of this call are not valid";
  return fail(h, SCANN_ERR_RANGE, m);
}

#define HIPCHK(h, expr)                                                                               \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return fail(h, e_ == hipErrorOutOfMemory ? SCANN_ERR_OOM : SCANN_ERR_HIP,                       \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                 \
  } while (0)

// Canonical tensor list; mirrors create_model (scann_model.py:329-453) and the layer constructors
// (attention.py:25-35, 95-113, 260-262).  Must stay in step with oracle/scann_oracle.py:weight_shapes.
std::vector<WeightSpec> build_specs(const scann_config_t& c) {
  std::vector<WeightSpec> s;
  const int64_t d = c.local_dim, dg = c.global_dim, dout = c.dense_out, emb = c.embedding_dim;
  if (c.feature_cgcnn) {
    s.push_back({"embed_atom/kernel", 92, emb});
    s.push_back({"embed_atom/bias", emb, 0});
  } else {
    s.push_back({"embed_atom/embeddings", c.n_atoms, emb});
  }
  int64_t cin = emb;
  if (c.use_ring) {
    s.push_back({"extra_embed/kernel", 2, 10});
    s.push_back({"extra_embed/bias", 10, 0});
    cin += 10;
  }
  s.push_back({"dense_embed/kernel", cin, d});
  s.push_back({"dense_embed/bias", d, 0});
  if (c.g_update) {
    s.push_back({"neighbor_d/kernel", c.n_gauss, d});
    s.push_back({"neighbor_d/bias", d, 0});
    s.push_back({"neighbor_w/kernel", c.n_gauss, d});
    s.push_back({"neighbor_w/bias", d, 0});
  }
  for (int i = 0; i < c.n_attention; ++i) {
    const std::string p = "local_attention_" + std::to_string(i) + "/";
    s.push_back({p + "query/kernel", d, d});
    s.push_back({p + "query/bias", d, 0});
    s.push_back({p + "key/kernel", d, d});
    s.push_back({p + "key/bias", d, 0});
    s.push_back({p + "filter_geo/kernel", c.g_update ? 3 * d : (int64_t)c.n_gauss, d});
    s.push_back({p + "filter_geo/bias", d, 0});
    s.push_back({p + "layer_norm/gamma", d, 0});
    s.push_back({p + "layer_norm/beta", d, 0});
    if (c.g_update) {
      s.push_back({p + "layer_norm_g/gamma", d, 0});
      s.push_back({p + "layer_norm_g/beta", d, 0});
    }
    if (c.use_attn_norm) {
      const std::string r = "residual_norm_" + std::to_string(i) + "/";
      s.push_back({r + "dense_1/kernel", d, d});
      s.push_back({r + "dense_1/bias", d, 0});
      s.push_back({r + "dense_2/kernel", d, d});
      s.push_back({r + "dense_2/bias", d, 0});
      s.push_back({r + "layer_norm/gamma", d, 0});
      s.push_back({r + "layer_norm/beta", d, 0});
    }
  }
  s.push_back({"after_Lc/kernel", d, dg});
  s.push_back({"after_Lc/bias", dg, 0});
  s.push_back({"global_attention/query/kernel", dg, dg});
  s.push_back({"global_attention/query/bias", dg, 0});
  s.push_back({"global_attention/key/kernel", dg, dg});
  s.push_back({"global_attention/key/bias", dg, 0});
  s.push_back({"bf_property/kernel", dg, dout});
  s.push_back({"bf_property/bias", dout, 0});
  s.push_back({"predict_property/kernel", dout, 1});
  s.push_back({"predict_property/bias", 1, 0});
  return s;
}

// np.linspace(0, stop, 20, dtype="float32") (scann_model.py:378,384): computed in double, cast once.
void linspace20(double stop, float* out) {
  const double step = stop / (NG - 1);
  for (int i = 0; i < NG; ++i) out[i] = (float)(i * step);
  out[NG - 1] = (float)stop;
}

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace

namespace scann {
// Fragment order consumed by gemm128 (scann_kernels.hip): element ((w*16 + t)*64 + lane)*4 + i holds
// W[8t + 4(lane>>5) + i][32w + (lane&31)].
void pack_weight(const float* W, int ld, float* Wp) {
  for (int w = 0; w < 4; ++w)
    for (int t = 0; t < 16; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 4; ++i)
          Wp[((size_t)(w * 16 + t) * 64 + lane) * 4 + i] = W[(size_t)(8 * t + 4 * (lane >> 5) + i) * ld + 32 * w + (lane & 31)];
}

// fp32 -> fp16 bits, round to nearest even, subnormals kept (host twin of the device's v_cvt_f16_f32)
static uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7FFFFFFFu;
  if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0));  // inf / nan
  if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                    // rounds to >= 65520: inf
  if (x < 0x33000001u) return (uint16_t)sign;                                                  // <= 2^-25: zero
  const int e = (int)(x >> 23) - 127;
  uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
  int shift = 13;
  uint32_t base;
  if (e < -14) {  // subnormal result
    shift += -14 - e;
    base = 0;
  } else {
    base = (uint32_t)(e + 15) << 10;
    m &= 0x7FFFFFu;
  }
  const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
  uint32_t r = base + q;  // a mantissa carry runs into the exponent field, which is the right result
  if (rem > halfway || (rem == halfway && (q & 1))) ++r;
  return (uint16_t)(sign | r);
}
static float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31, m = h & 0x3FFu;
  float out;
  if (e == 0) {
    out = std::ldexp((float)m, -24);
  } else if (e == 31) {
    uint32_t x = 0x7F800000u | (m << 13);
    memcpy(&out, &x, 4);
  } else {
    out = std::ldexp((float)(m | 0x400u), (int)e - 25);
  }
  uint32_t x;
  memcpy(&x, &out, 4);
  x |= sign;
  memcpy(&out, &x, 4);
  return out;
}
void pack_weight_f16(const float* W, int ld, int k_real, int ks, uint16_t* out) {
  for (int w = 0; w < 4; ++w)
    for (int s = 0; s < ks; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int k = 16 * s + 8 * (lane >> 5) + j;
          const float x = k < k_real ? WSCALE * W[(size_t)k * ld + 32 * w + (lane & 31)] : 0.f;
          const uint16_t hi = f32_to_f16_bits(x);
          const uint16_t lo = f32_to_f16_bits(x - f16_bits_to_f32(hi));
          const size_t base = ((size_t)(w * ks + s) * 2) * 64 * 8 + (size_t)lane * 8 + j;
          out[base] = hi;
          out[base + 64 * 8] = lo;
        }
}
}  // namespace scann

extern "C" {

int scann_abi_version(void) { return SCANN_ABI_VERSION; }

int scann_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* scann_last_error(const scann_handle_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int scann_create(const scann_config_t* cfg, int device_id, scann_handle_t** out) {
  if (!cfg || !out) return fail(nullptr, SCANN_ERR_INVALID, "scann_create: null argument");
  *out = nullptr;
  if (cfg->n_gauss != NG) return fail(nullptr, SCANN_ERR_UNSUPPORTED, "scann_create: 20 Gaussians (custom_layers.py:39-53 as called by scann_model.py:378,384)");
  // every shipped reference config is 128 / 8 / 128 / 128: the MFMA kernels; any other widths the reference accepts
  // (scann_model.py:330-434): the plain-fp32 forward of scann_generic.hip, inference only
  const bool generic = cfg->local_dim != D || cfg->global_dim != D || cfg->dense_out != D || cfg->num_head != NHEAD ||
                       (getenv("SCANN_GENERIC") && atoi(getenv("SCANN_GENERIC")));
  if (generic) {
    if (cfg->local_dim <= 0 || cfg->num_head <= 0 || cfg->global_dim <= 0 || cfg->dense_out <= 0 || cfg->local_dim % cfg->num_head != 0)
      return fail(nullptr, SCANN_ERR_INVALID, "scann_create: local_dim must be a positive multiple of num_head (attention.py:170-173), global_dim and dense_out positive");
    if (cfg->local_dim > 1024 || cfg->global_dim > 1024 || cfg->dense_out > 1024 || cfg->embedding_dim > 1024)
      return fail(nullptr, SCANN_ERR_UNSUPPORTED, "scann_create: widths above 1024 are not implemented");
  } else if (cfg->embedding_dim + (cfg->use_ring ? 10 : 0) > 160) {
    return fail(nullptr, SCANN_ERR_UNSUPPORTED, "scann_create: embedding_dim (+10 ring features) must be <= 160");
  }
  if (cfg->n_atoms <= 0 || cfg->embedding_dim <= 0 || cfg->n_attention < 0 || !(cfg->gaussian_d > 0))
    return fail(nullptr, SCANN_ERR_INVALID, "scann_create: bad hyper-parameter");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, SCANN_ERR_NO_DEVICE, "scann_create: no HIP device visible (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, SCANN_ERR_NO_DEVICE, "scann_create: device_id out of range");
  scann_handle* h = new scann_handle();
  h->cfg = *cfg;
  h->generic = generic;
  h->device = device_id;
  h->specs = build_specs(*cfg);
  if (const char* xr = getenv("SCANN_XCD_REMAP")) h->xcd_remap = atoi(xr) != 0;
  if (const char* fb = getenv("SCANN_FUSE_BASIS")) h->fuse_basis = atoi(fb) != 0;
  if (const char* st = getenv("SCANN_SPECIES_TABLES")) h->species_tables = atoi(st) != 0;
  if (const char* sg = getenv("SCANN_STRICT_RANGE")) h->strict_range = atoi(sg) != 0;
  if (const char* fe = getenv("SCANN_EXACT")) h->force_exact = atoi(fe) != 0;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
  }
  if (const char* ns = getenv("SCANN_STREAMS")) h->nstream = std::min(MAX_STREAM, std::max(1, atoi(ns)));
  if (hipSetDevice(device_id) != hipSuccess) {
    delete h;
    return fail(nullptr, SCANN_ERR_HIP, "scann_create: hipSetDevice failed");
  }
  if (hipHostMalloc((void**)&h->range_flag, 128, hipHostMallocDefault) != hipSuccess) {  // [0, 16): range guard per stream
    delete h;
    return fail(nullptr, SCANN_ERR_HIP, "scann_create: hipHostMalloc failed");
  }
  for (int i = 0; i < 32; ++i) h->range_flag[i] = 0;  // one range-guard word per stream slot
  for (int i = 0; i < h->nstream; ++i) {
    if (hipStreamCreateWithFlags(&h->streams[i], hipStreamNonBlocking) != hipSuccess) {
      delete h;
      return fail(nullptr, SCANN_ERR_HIP, "scann_create: hipStreamCreate failed");
    }
  }
  *out = h;
  return SCANN_OK;
}

void scann_destroy(scann_handle_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < MAX_STREAM; ++i)
    if (h->streams[i]) (void)hipStreamDestroy(h->streams[i]);
  if (h->d_weights) (void)hipFree(h->d_weights);
  for (scann_handle::DlStage& st : h->dl_stage)
    if (st.p) (void)hipHostFree(st.p);
  if (h->g_weights) (void)hipFree(h->g_weights);
  if (h->g_centres) (void)hipFree(h->g_centres);
  if (h->g_WT) (void)hipFree(h->g_WT);
  if (h->d_gt_descs) (void)hipFree(h->d_gt_descs);
  if (h->sp_c) (void)hipFree(h->sp_c);
  for (scann_handle::Stage& st : h->stage) {
    if (st.p) (void)hipHostFree(st.p);
    if (st.ev) (void)hipEventDestroy(st.ev);
  }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  for (void* q : {(void*)h->t_master, (void*)h->t_grad, (void*)h->t_m, (void*)h->t_v, (void*)h->t_l2, (void*)h->t_descs})
    if (q) (void)hipFree(q);
  if (h->comm) ncclCommDestroy(h->comm);
  if (h->train_aux && !h->train_aux_borrowed) (void)hipStreamDestroy(h->train_aux);
  if (h->h_stat) (void)hipHostFree(h->h_stat);
  if (h->range_flag) (void)hipHostFree(h->range_flag);
  for (float* t : h->h_targets)
    if (t) (void)hipHostFree(t);
  for (hipEvent_t e : h->step_ev)
    if (e) (void)hipEventDestroy(e);
  if (h->train_aux2) (void)hipStreamDestroy(h->train_aux2);
  for (hipEvent_t e : h->train_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->time_ev) (void)hipEventDestroy(e);
  if (h->sc_db) {
    cached_free(h->sc_db->gen_ws);
    cached_free(h->sc_db->dbg_c);
    cached_free(h->sc_db->dbg_g);
    cached_free(h->sc_db->dbg_ctx);
    free_train_ws(h->sc_db);
    delete h->sc_db;
  }
  if (h->sc_arena) (void)hipFree(h->sc_arena);
  if (h->pp_dev) (void)hipFree(h->pp_dev);
  if (h->pp_host) (void)hipHostFree(h->pp_host);
  if (h->sc_host) (void)hipHostFree(h->sc_host);
  cache_release(h->device);
  delete h;
}

int scann_num_streams(const scann_handle_t* h) { return h ? h->nstream : 0; }

int scann_weight_count(const scann_handle_t* h) { return h ? (int)h->specs.size() : SCANN_ERR_INVALID; }

int scann_weight_name(const scann_handle_t* h, int index, const char** name, int64_t* rows, int64_t* cols) {
  if (!h || index < 0 || index >= (int)h->specs.size()) return SCANN_ERR_INVALID;
  if (name) *name = h->specs[index].name.c_str();
  if (rows) *rows = h->specs[index].rows;
  if (cols) *cols = h->specs[index].cols;
  return SCANN_OK;
}

int scann_load_weights(scann_handle_t* h, const float* blob, const scann_tensor_desc_t* manifest, int n) {
  if (!h || !blob || !manifest || n <= 0) return fail(h, SCANN_ERR_INVALID, "scann_load_weights: null argument");
  HIPCHK(h, hipSetDevice(h->device));
  std::map<std::string, const scann_tensor_desc_t*> by_name;
  for (int i = 0; i < n; ++i)
    if (manifest[i].name) by_name[manifest[i].name] = &manifest[i];
  // canonical flat parameter vector: the tensors in spec order
  std::map<std::string, const float*> src;
  {
    size_t total = 0;
    for (const WeightSpec& s : h->specs) {
      auto it = by_name.find(s.name);
      if (it == by_name.end()) return fail(h, SCANN_ERR_WEIGHTS, "scann_load_weights: missing tensor " + s.name);
      if (it->second->numel != s.numel() || it->second->offset < 0)
        return fail(h, SCANN_ERR_WEIGHTS, "scann_load_weights: wrong size for tensor " + s.name);
      total += (size_t)s.numel();
    }
    h->host_master.resize(total);
    h->spec_off.clear();
    size_t off = 0;
    for (const WeightSpec& s : h->specs) {
      memcpy(h->host_master.data() + off, blob + by_name[s.name]->offset, (size_t)s.numel() * sizeof(float));
      src[s.name] = h->host_master.data() + off;
      h->spec_off.push_back((int64_t)off);
      off += (size_t)s.numel();
    }
  }
  if (h->generic) {  // the plain-fp32 forward reads the Keras tensors as they are
    for (float v : h->host_master)
      if (!std::isfinite(v)) return fail(h, SCANN_ERR_WEIGHTS, "scann_load_weights: a parameter is not finite");
    if (h->g_weights) (void)hipFree(h->g_weights);
    h->g_weights = nullptr;
    HIPCHK(h, hipMalloc((void**)&h->g_weights, h->host_master.size() * sizeof(float)));
    HIPCHK(h, hipMemcpy(h->g_weights, h->host_master.data(), h->host_master.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!h->g_centres) {
      float cen[2 * NG];
      linspace20((double)h->cfg.gaussian_d, cen);
      linspace20(M_PI * 2.0, cen + NG);
      HIPCHK(h, hipMalloc((void**)&h->g_centres, sizeof(cen)));
      HIPCHK(h, hipMemcpy(h->g_centres, cen, sizeof(cen), hipMemcpyHostToDevice));
    }
    h->g_off.clear();
    for (size_t i = 0; i < h->specs.size(); ++i) h->g_off[h->specs[i].name] = h->spec_off[i];
    h->loaded = true;
    return SCANN_OK;
  }
  h->descs.clear();
  const float* const mbase = h->host_master.data();
  const float* const mend = mbase + h->host_master.size();
  const scann_config_t& c = h->cfg;
  const int L = c.n_attention;
  // host image of the device arena
  std::vector<float> img;
  img.reserve((size_t)(L * 8 + 8) * WPACK);
  auto put_raw = [&](const float* p, size_t numel) {
    const size_t off = img.size();
    img.insert(img.end(), p, p + numel);
    while (img.size() % 64) img.push_back(0.f);  // keep every tensor 256-byte aligned
    if (p >= mbase && p < mend)                  // derived from a parameter: re-copied after every optimiser step
      for (size_t done = 0; done < numel; done += 16384)
        h->descs.push_back(RepackDesc{(int64_t)(p - mbase + done), (int64_t)(off + done), 0, (int32_t)std::min<size_t>(16384, numel - done)});
    return off;
  };
  auto put_packed = [&](const float* W) {  // W: row-major [128,128] slice, leading dim 128
    const size_t off = img.size();
    img.resize(off + WPACK);
    pack_weight(W, D, img.data() + off);
    h->descs.push_back(RepackDesc{(int64_t)(W - mbase), (int64_t)off, 0, 0});
    return off;
  };
  auto put_packedT = [&](const float* W) {  // fragment-order image of W^T (backward: dX = dY . W^T)
    const size_t off = img.size();
    img.resize(off + WPACK);
    std::vector<float> wt((size_t)D * D);
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) wt[(size_t)j * D + i] = W[(size_t)i * D + j];
    pack_weight(wt.data(), D, img.data() + off);
    h->descs.push_back(RepackDesc{(int64_t)(W - mbase), (int64_t)off, 1, 0});
    return off;
  };
  auto put_f16 = [&](const float* W, int k_real, int ks) {  // split-fp16 image (edge_kernel); ks * 2048 floats' worth of bytes
    const size_t off = img.size();
    img.resize(off + (size_t)ks * 2048);
    pack_weight_f16(W, D, k_real, ks, reinterpret_cast<uint16_t*>(img.data() + off));
    h->descs.push_back(RepackDesc{(int64_t)(W - mbase), (int64_t)off, 0, ks == 8 ? -1 : -2});
    return off;
  };
  auto put_f16T = [&](const float* W) {  // split-fp16 image of W^T ([128,128]; fused backward kernels: dX = dY . W^T)
    const size_t off = img.size();
    img.resize(off + (size_t)8 * 2048);
    std::vector<float> wt((size_t)D * D);
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) wt[(size_t)j * D + i] = W[(size_t)i * D + j];
    pack_weight_f16(wt.data(), D, D, 8, reinterpret_cast<uint16_t*>(img.data() + off));
    h->descs.push_back(RepackDesc{(int64_t)(W - mbase), (int64_t)off, 1, -1});
    return off;
  };
  // The fp16 hi part of a split weight holds |w| * 2^8 < 65504.  A 128x128 kernel beyond that (the reference loads any fp32
  // checkpoint, scann_model.py:79) sends every inference forward of the handle to the exact-fp32 kernels; the K = 20 filters exist
  // in the split form only, and nothing can be done with a value that is not finite.
  h->weights_exact = false;
  for (const WeightSpec& sp : h->specs)
    if (sp.cols) {
      const float* wp = src[sp.name];
      const bool split_only = sp.name == "neighbor_d/kernel" || sp.name == "neighbor_w/kernel" ||
                              (!c.g_update && sp.name.size() > 17 && sp.name.compare(sp.name.size() - 17, 17, "filter_geo/kernel") == 0);
      const bool projection = sp.rows == D * (sp.name.find("filter_geo") != std::string::npos ? 3 : 1) && sp.cols == D &&
                              sp.name != "bf_property/kernel" && sp.name != "dense_embed/kernel";
      for (int64_t i = 0; i < sp.numel(); ++i)
        if (!(std::fabs(wp[i]) < WMAX)) {
          if (std::isfinite(wp[i]) && projection && !split_only) {
            h->weights_exact = true;
            break;
          }
          if (!std::isfinite(wp[i]) || split_only)
            return fail(h, SCANN_ERR_UNSUPPORTED, "scann_load_weights: |" + sp.name + "| reaches " + std::to_string(std::fabs(wp[i])) +
                                                      (split_only ? "; this kernel is multiplied in split-fp16 form only, which needs |w| < 255.9"
                                                                  : "; the value is not finite"));
          break;  // an fp32-only tensor (embedding, dense_embed, bf_property, head): any finite value
        }
    }
  struct LTOff { size_t W1T, W2T, W3T, WqT, WkT, Wf1T, Wf2T, W1Th, W2Th, W3Th, WqTh, WkTh, Wf1Th, Wf2Th; };
  std::vector<LTOff> lto(L);
  struct LOff {
    size_t bg, bq, bk, lng_g, lng_b, ln_g, ln_b, Wfg, bfg, bf1, bf2, lnr_g, lnr_b;
    size_t W2h, Wkh, Wfh, W1h, W3h, Wqh, Wf1h, Wf2h;
    size_t W1p, W2p, W3p, Wqp, Wkp, Wf1p, Wf2p;  // fp32 fragment-order images: the exact-fp32 fallback of the forward
  };
  std::vector<LOff> lo(L);
  const size_t NONE = (size_t)-1;
  for (int i = 0; i < L; ++i) {
    const std::string p = "local_attention_" + std::to_string(i) + "/";
    LOff& o = lo[i];
    o = LOff{NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE,
             NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE};
    const float* fg = src[p + "filter_geo/kernel"];
    lto[i] = LTOff{NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE};
    if (c.g_update) {  // concat order [centre, geometry, neighbour] (attention.py:143-149)
      o.W1h = put_f16(fg, D, 8);
      o.W3h = put_f16(fg + (size_t)2 * D * D, D, 8);
      o.W2h = put_f16(fg + (size_t)D * D, D, 8);
      o.W1p = put_packed(fg);
      o.W2p = put_packed(fg + (size_t)D * D);
      o.W3p = put_packed(fg + (size_t)2 * D * D);
      lto[i].W1T = put_packedT(fg);
      lto[i].W2T = put_packedT(fg + (size_t)D * D);
      lto[i].W3T = put_packedT(fg + (size_t)2 * D * D);
      lto[i].W1Th = put_f16T(fg);
      lto[i].W2Th = put_f16T(fg + (size_t)D * D);
      lto[i].W3Th = put_f16T(fg + (size_t)2 * D * D);
      o.bg = put_raw(src[p + "filter_geo/bias"], D);
      o.lng_g = put_raw(src[p + "layer_norm_g/gamma"], D);
      o.lng_b = put_raw(src[p + "layer_norm_g/beta"], D);
    } else {
      o.Wfg = put_raw(fg, (size_t)NG * D);
      o.Wfh = put_f16(fg, NG, 2);
      o.bfg = put_raw(src[p + "filter_geo/bias"], D);
    }
    o.Wqh = put_f16(src[p + "query/kernel"], D, 8);
    o.Wqp = put_packed(src[p + "query/kernel"]);
    o.Wkp = put_packed(src[p + "key/kernel"]);
    lto[i].WqT = put_packedT(src[p + "query/kernel"]);
    lto[i].WkT = put_packedT(src[p + "key/kernel"]);
    lto[i].WqTh = put_f16T(src[p + "query/kernel"]);
    lto[i].WkTh = put_f16T(src[p + "key/kernel"]);
    o.bq = put_raw(src[p + "query/bias"], D);
    o.Wkh = put_f16(src[p + "key/kernel"], D, 8);
    o.bk = put_raw(src[p + "key/bias"], D);
    o.ln_g = put_raw(src[p + "layer_norm/gamma"], D);
    o.ln_b = put_raw(src[p + "layer_norm/beta"], D);
    if (c.use_attn_norm) {
      const std::string r = "residual_norm_" + std::to_string(i) + "/";
      lto[i].Wf1T = put_packedT(src[r + "dense_1/kernel"]);
      lto[i].Wf2T = put_packedT(src[r + "dense_2/kernel"]);
      lto[i].Wf1Th = put_f16T(src[r + "dense_1/kernel"]);
      lto[i].Wf2Th = put_f16T(src[r + "dense_2/kernel"]);
      o.Wf1h = put_f16(src[r + "dense_1/kernel"], D, 8);
      o.Wf2h = put_f16(src[r + "dense_2/kernel"], D, 8);
      o.Wf1p = put_packed(src[r + "dense_1/kernel"]);
      o.Wf2p = put_packed(src[r + "dense_2/kernel"]);
      o.bf1 = put_raw(src[r + "dense_1/bias"], D);
      o.bf2 = put_raw(src[r + "dense_2/bias"], D);
      o.lnr_g = put_raw(src[r + "layer_norm/gamma"], D);
      o.lnr_b = put_raw(src[r + "layer_norm/beta"], D);
    }
  }
  const size_t oWaTh = put_f16T(src["after_Lc/kernel"]);
  const size_t oWaT = put_packedT(src["after_Lc/kernel"]), oWgqT = put_packedT(src["global_attention/query/kernel"]),
               oWgkT = put_packedT(src["global_attention/key/kernel"]);
  const size_t oWah = put_f16(src["after_Lc/kernel"], D, 8), oWgqh = put_f16(src["global_attention/query/kernel"], D, 8),
               oWgkh = put_f16(src["global_attention/key/kernel"], D, 8);
  const size_t oWa = put_packed(src["after_Lc/kernel"]), oba = put_raw(src["after_Lc/bias"], D);
  const size_t oWgq = put_packed(src["global_attention/query/kernel"]), obgq = put_raw(src["global_attention/query/bias"], D);
  const size_t oWgk = put_packed(src["global_attention/key/kernel"]), obgk = put_raw(src["global_attention/key/bias"], D);
  const size_t oWb = put_raw(src["bf_property/kernel"], (size_t)D * D), obb = put_raw(src["bf_property/bias"], D);
  const size_t owo = put_raw(src["predict_property/kernel"], D), obo = put_raw(src["predict_property/bias"], 1);
  size_t oWd = NONE, obd = NONE, oWw = NONE, obw = NONE, oWdh = NONE, oWwh = NONE;
  if (c.g_update) {
    oWdh = put_f16(src["neighbor_d/kernel"], NG, 2);
    oWwh = put_f16(src["neighbor_w/kernel"], NG, 2);
    oWd = put_raw(src["neighbor_d/kernel"], (size_t)NG * D);
    obd = put_raw(src["neighbor_d/bias"], D);
    oWw = put_raw(src["neighbor_w/kernel"], (size_t)NG * D);
    obw = put_raw(src["neighbor_w/bias"], D);
  }
  float cen[2 * NG];
  linspace20((double)c.gaussian_d, cen);
  linspace20(M_PI * 2.0, cen + NG);
  const size_t ocd = put_raw(cen, NG), ocw = put_raw(cen + NG, NG);
  const bool general_embed = c.use_ring || c.feature_cgcnn;
  const int64_t cin = c.embedding_dim + (c.use_ring ? 10 : 0);
  size_t oemb = NONE, oWc = NONE, obc = NONE, oWr = NONE, obr = NONE;
  if (c.feature_cgcnn) {
    oWc = put_raw(src["embed_atom/kernel"], (size_t)92 * c.embedding_dim);
    obc = put_raw(src["embed_atom/bias"], c.embedding_dim);
  } else {
    oemb = put_raw(src["embed_atom/embeddings"], (size_t)c.n_atoms * c.embedding_dim);
  }
  if (c.use_ring) {
    oWr = put_raw(src["extra_embed/kernel"], 20);
    obr = put_raw(src["extra_embed/bias"], 10);
  }
  const size_t oWe = put_raw(src["dense_embed/kernel"], (size_t)cin * D);
  const size_t obe = put_raw(src["dense_embed/bias"], D);
  const size_t olut = img.size();
  img.resize(olut + (size_t)c.n_atoms * D, 0.f);

  if (h->d_weights) {
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipFree(h->d_weights));
    h->d_weights = nullptr;
    h->loaded = false;
  }
  HIPCHK(h, hipMalloc((void**)&h->d_weights, img.size() * sizeof(float)));
  HIPCHK(h, hipMemcpy(h->d_weights, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
  const float* base = h->d_weights;
  auto P = [&](size_t off) -> const float* { return off == NONE ? nullptr : base + off; };
  h->layers.assign(L, LayerParams{});
  for (int i = 0; i < L; ++i) {
    const LOff& o = lo[i];
    LayerParams& lp = h->layers[i];
    lp.bg = P(o.bg); lp.bq = P(o.bq); lp.bk = P(o.bk);
    lp.lng_g = P(o.lng_g); lp.lng_b = P(o.lng_b); lp.ln_g = P(o.ln_g); lp.ln_b = P(o.ln_b);
    lp.Wfg = P(o.Wfg); lp.bfg = P(o.bfg);
    lp.W2h = reinterpret_cast<const _Float16*>(P(o.W2h)); lp.Wkh = reinterpret_cast<const _Float16*>(P(o.Wkh));
    lp.Wfh = reinterpret_cast<const _Float16*>(P(o.Wfh));
    lp.W1h = reinterpret_cast<const _Float16*>(P(o.W1h)); lp.W3h = reinterpret_cast<const _Float16*>(P(o.W3h));
    lp.Wqh = reinterpret_cast<const _Float16*>(P(o.Wqh)); lp.Wf1h = reinterpret_cast<const _Float16*>(P(o.Wf1h));
    lp.Wf2h = reinterpret_cast<const _Float16*>(P(o.Wf2h));
    lp.bf1 = P(o.bf1); lp.bf2 = P(o.bf2);
    lp.lnr_g = P(o.lnr_g); lp.lnr_b = P(o.lnr_b);
    lp.W1p = P(o.W1p); lp.W2p = P(o.W2p); lp.W3p = P(o.W3p); lp.Wqp = P(o.Wqp); lp.Wkp = P(o.Wkp); lp.Wf1p = P(o.Wf1p); lp.Wf2p = P(o.Wf2p);
  }
  h->layersT.assign(L, scann_handle::LayerT{});
  for (int i = 0; i < L; ++i) {
    auto PH = [&](size_t o) { return reinterpret_cast<const _Float16*>(P(o)); };
    h->layersT[i] = scann_handle::LayerT{P(lto[i].W1T), P(lto[i].W2T), P(lto[i].W3T), P(lto[i].WqT), P(lto[i].WkT), P(lto[i].Wf1T), P(lto[i].Wf2T),
                                         PH(lto[i].W1Th), PH(lto[i].W2Th), PH(lto[i].W3Th), PH(lto[i].WqTh), PH(lto[i].WkTh), PH(lto[i].Wf1Th),
                                         PH(lto[i].Wf2Th)};
  }
  h->WaT = P(oWaT); h->WgqT = P(oWgqT); h->WgkT = P(oWgkT);
  h->WaTh = reinterpret_cast<const _Float16*>(P(oWaTh));
  h->arena_floats = img.size(); h->o_lut = olut; h->o_emb = oemb; h->o_Wde = oWe; h->o_bde = obe;
  h->head = HeadParams{P(oWa), P(oba), P(oWgq), P(obgq), P(oWgk), P(obgk),
                       reinterpret_cast<const _Float16*>(P(oWah)), reinterpret_cast<const _Float16*>(P(oWgqh)),
                       reinterpret_cast<const _Float16*>(P(oWgkh)), P(oWb), P(obb), P(owo), P(obo)};
  h->basis = BasisParams{P(oWd), P(obd), P(oWw), P(obw), P(ocd), P(ocw), reinterpret_cast<const _Float16*>(P(oWdh)),
                         reinterpret_cast<const _Float16*>(P(oWwh))};
  h->cd = P(ocd);
  h->lut = P(olut);
  h->embed = EmbedArgs{};
  h->embed.emb_dim = c.embedding_dim;
  h->embed.emb = P(oemb); h->embed.We = P(oWc); h->embed.be = P(obc); h->embed.Wr = P(oWr); h->embed.br = P(obr);
  h->embed.Wde = P(oWe); h->embed.bde = P(obe);
  h->sp_dirty = true;
  if (!general_embed && !h->sp_c && c.g_update) {  // c | P1 | P3 | q tables of the first layer, one allocation
    const size_t tab = (size_t)c.n_atoms * D;
    HIPCHK(h, hipMalloc((void**)&h->sp_c, 4 * tab * sizeof(float)));
    h->sp_P1 = h->sp_c + tab; h->sp_P3 = h->sp_c + 2 * tab; h->sp_q = h->sp_c + 3 * tab;
  }
  if (!general_embed) {
    // Embedding + dense_embed folded into a per-species table, computed on the device.
    launch_embed_lut(P(oemb), P(oWe), P(obe), c.n_atoms, c.embedding_dim, h->d_weights + olut, h->streams[0]);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->streams[0]));
  }
  h->loaded = true;
  return SCANN_OK;
}

int scann_set_debug(scann_handle_t* h, int on) {
  if (!h) return SCANN_ERR_INVALID;
  if (on && h->generic) return fail(h, SCANN_ERR_UNSUPPORTED, "scann_set_debug: per-layer intermediates exist for the 128-wide kernels only");
  h->debug = on != 0;
  return SCANN_OK;
}

static void free_train_ws(scann_dbatch* db);

void scann_batch_free(scann_handle_t* h, scann_dbatch_t* db) {
  if (!db) return;
  if (h) (void)hipSetDevice(h->device);
  if (h) (void)hipDeviceSynchronize();
  if (db->arena && db->owns_arena) cached_free(db->arena);
  cached_free(db->gen_ws);
  cached_free(db->dbg_c);
  cached_free(db->dbg_g);
  cached_free(db->dbg_ctx);
  if (db->stamps) (void)hipFree(db->stamps);
  if (db->upload_ev) (void)hipEventDestroy(db->upload_ev);
  free_train_ws(db);
  delete db;
}

// scann_batch_free without the device-wide synchronisation: for a batch whose last use was a scann_train_step that has been ended
// (its event has fired), or a forward whose results have been downloaded (scann_batch_download waits for the batch's stream), while
// LATER work on other batches may still be running.  Falls back to the synchronising free otherwise.
void scann_batch_release(scann_handle_t* h, scann_dbatch_t* db) {
  if (!db) return;
  const bool step_done = db->busy_ev && hipEventQuery(db->busy_ev) == hipSuccess;
  if (!h || !(step_done || db->idle)) {
    scann_batch_free(h, db);
    return;
  }
  (void)hipSetDevice(h->device);
  if (db->arena && db->owns_arena) cached_free(db->arena);
  cached_free(db->gen_ws);
  cached_free(db->dbg_c);
  cached_free(db->dbg_g);
  cached_free(db->dbg_ctx);
  if (db->stamps) (void)hipFree(db->stamps);
  if (db->upload_ev) (void)hipEventDestroy(db->upload_ev);
  free_train_ws(db);
  delete db;
}

// Device packing (scann_upload_padded): the payload arrays of a padded Keras input dict, which go to the device AS THEY ARE and are
// compacted there (pack_padded_kernel); `b` then carries the counts and the two offset arrays only (host: scann_count_padded).
struct PaddedSrc {
  int32_t M, N;
  const int32_t *atomic, *neighbors;
  const void* neighbor_mask;
  int32_t mask_size;
  const float *weight, *dist;
  const int32_t* row_of;  // [B*M] host
  // the payload arrays are ALREADY on their way to the device (scann_forward_padded enqueued the copy on the stream the upload uses,
  // before it read the masks): their device addresses; null: upload_impl stages and copies them itself
  const int32_t *d_atomic = nullptr, *d_neighbors = nullptr;
  const void* d_mask = nullptr;
  const float *d_weight = nullptr, *d_dist = nullptr;
};

// (the threaded staging copy of the padded payload: scann_host_copy, scann_pack.cpp -- host-only code, built under ThreadSanitizer too)
static inline void par_memcpy(void* dst, const void* src, size_t bytes) { (void)scann_host_copy(dst, src, (int64_t)bytes); }

static int upload_impl(scann_handle_t* h, const scann_batch_t* b, scann_dbatch_t** out, bool scratch, const PaddedSrc* pad = nullptr) {
  if (!h || !b || !out) return fail(h, SCANN_ERR_INVALID, "scann_batch_upload: null argument");
  *out = nullptr;
  const int32_t B = b->n_struct, A = b->n_atom, E = b->n_edge;
  if (B <= 0 || A <= 0 || E < 0) return fail(h, SCANN_ERR_INVALID, "scann_batch_upload: empty batch");
  if (!b->mol_offset || !b->edge_offset ||
      (!pad && ((!b->atomic && !h->cfg.feature_cgcnn) || (E > 0 && (!b->edge_col || !b->edge_dist || !b->edge_weight)))))
    return fail(h, SCANN_ERR_INVALID, "scann_batch_upload: null array");
  if (pad && (h->cfg.feature_cgcnn || h->cfg.use_ring || h->t_master))
    return fail(h, SCANN_ERR_UNSUPPORTED, "scann_upload_padded: atomic feature without ring, inference handles (a training handle needs the edges on the host)");
  if (b->mol_offset[0] != 0 || b->mol_offset[B] != A || b->edge_offset[0] != 0 || b->edge_offset[A] != E)
    return fail(h, SCANN_ERR_INVALID, "scann_batch_upload: offsets do not cover the batch");
  int32_t max_atoms = 0;
  for (int s = 0; s < B; ++s) {
    const int32_t n = b->mol_offset[s + 1] - b->mol_offset[s];
    if (n <= 0) return fail(h, SCANN_ERR_INVALID, "scann_batch_upload: structure without atoms");
    max_atoms = std::max(max_atoms, n);
  }
  if ((size_t)max_atoms * 5 * sizeof(float) > 60000) return fail(h, SCANN_ERR_UNSUPPORTED, "scann_batch_upload: structure too large");
  if ((uint64_t)std::max(A, E) * D * 4 >= (1ull << 32))  // edge_kernel addresses a tensor row as base + 32-bit byte offset
    return fail(h, SCANN_ERR_UNSUPPORTED, "scann_batch_upload: more than 8,388,607 atoms or edges in one batch; split it");
  if (pad) {
    // (atomic numbers and neighbour indices are checked where they are read: pack_padded_kernel's flag word, scann_batch_download)
  } else if (!h->cfg.feature_cgcnn) {
    for (int a = 0; a < A; ++a)
      if (b->atomic[a] < 0 || b->atomic[a] >= h->cfg.n_atoms)
        return fail(h, SCANN_ERR_INVALID, "scann_batch_upload: atomic number outside the embedding table (n_atoms)");
  } else if (!b->cgcnn) {
    return fail(h, SCANN_ERR_INVALID, "scann_batch_upload: feature=cgcnn needs batch.cgcnn [n_atom,92]");
  }
  if (h->cfg.use_ring && !b->ring) return fail(h, SCANN_ERR_INVALID, "scann_batch_upload: use_ring needs batch.ring [n_atom,2]");
  std::vector<int32_t> edge_row;
  std::vector<EdgeTile> tiles;
  std::vector<int32_t> tile_part, big_tab;  // atoms with more than TE_MAX neighbours (edge_kernel_lean only)
  int32_t n_slot = 0, max_degree = 0;
  int tile_rows = TE_MAX;
  {
    std::string err;
    // A launch that fits ONE round of workgroups is the latency chain of a tile: 32-row tiles (four workgroups per CU = 1,024 slots)
    // make that chain shorter.  Only when no atom needs chunking at 32 rows: such a batch is planned at 32 rows first (one pass for
    // the reference's batch of 128) and again at 64 if an atom turns out to have more than 32 neighbours.
    const bool small = E > 0 && E <= 32 * 1024;
    int r = plan_tiles(b->mol_offset, B, b->edge_offset, pad ? nullptr : b->edge_col, A, E, small ? 32 : TE_MAX, h->tile_atoms, true, tiles,
                       tile_part, big_tab, edge_row, &tile_rows, &max_degree, &n_slot, err, false);
    if (r) return fail(h, r, "scann_batch_upload: " + err);
    if (small && max_degree > 32) {
      r = plan_tiles(b->mol_offset, B, b->edge_offset, pad ? nullptr : b->edge_col, A, E, TE_MAX, h->tile_atoms, true, tiles, tile_part, big_tab, edge_row,
                     &tile_rows, &max_degree, &n_slot, err, false);
      if (r) return fail(h, r, "scann_batch_upload: " + err);
    }
  }
  const int32_t n_big = (int32_t)big_tab.size() / 3;
  HIPCHK(h, hipSetDevice(h->device));
  scann_dbatch* db = nullptr;
  if (scratch) {
    if (!h->sc_db) h->sc_db = new scann_dbatch();
    db = h->sc_db;
    if (db->dbg_c || db->stamps) (void)hipStreamSynchronize(h->streams[0]);
    cached_free(db->dbg_c);
    cached_free(db->dbg_g);
    cached_free(db->dbg_ctx);
    if (db->stamps) (void)hipFree(db->stamps);
    free_train_ws(db);
    char* const gen_ws = db->gen_ws;  // (the generic-width forward's workspace is kept across calls, like the arena below)
    const size_t gen_ws_bytes = db->gen_ws_bytes;
    *db = scann_dbatch();
    db->gen_ws = gen_ws; db->gen_ws_bytes = gen_ws_bytes;
    db->owns_arena = false;
  } else {
    db = new scann_dbatch();
  }
  db->n_struct = B; db->n_atom = A; db->n_edge = E; db->n_tile = (int32_t)tiles.size(); db->max_atoms = max_atoms; db->tile_rows = tile_rows; db->max_degree = max_degree; db->tile_atoms = h->tile_atoms; db->n_big = n_big; db->n_slot = n_slot;
  // arena layout: inputs first (one H2D copy), then workspace
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes); return o; };
  const size_t o_atomic = take((size_t)A * 4), o_mol = take((size_t)(B + 1) * 4), o_eoff = take((size_t)(A + 1) * 4);
  const size_t o_col = take((size_t)E * 4), o_row = take((size_t)E * 4), o_dist = take((size_t)E * 4), o_wgt = take((size_t)E * 4);
  const size_t o_tiles = take(tiles.size() * sizeof(EdgeTile));
  const size_t o_tpart = take(n_big ? tiles.size() * 4 : 0), o_big = take((size_t)n_big * 3 * 4);
  const size_t o_inoff = take((size_t)(A + 1) * 4), o_inedge = take((size_t)E * 4);
  const size_t o_ring = take(h->cfg.use_ring ? (size_t)A * 2 * 4 : 0), o_cg = take(h->cfg.feature_cgcnn ? (size_t)A * 92 * 4 : 0);
  // device packing: the padded payload as it came (transient: read once by pack_padded_kernel), the row map and the flag word
  const size_t BM = pad ? (size_t)B * pad->M : 0, BMN = pad ? BM * pad->N : 0;
  const bool pre = pad && pad->d_atomic;  // the payload is already on the device
  const size_t o_prow = take(BM * 4), o_pat = take(pre ? 0 : BM * 4), o_pnbr = take(pre ? 0 : BMN * 4);
  const size_t o_pmask = take(pad && !pre ? BMN * pad->mask_size : 0), o_pw = take(pre ? 0 : BMN * 4), o_pd = take(pre ? 0 : BMN * 4);
  const size_t in_bytes = off;
  const size_t rowA = (size_t)A * D * 4, rowE = (size_t)std::max(E, 1) * D * 4;
  const size_t o_geom = take(h->cfg.g_update ? rowE + D * 4 : 0);  // + the spare row edge-less tiles store to (EdgeArgs::n_edge)
  const size_t o_gd = take(h->cfg.g_update ? 0 : (size_t)std::max(E, 1) * NG * 4);
  const size_t o_c0 = take((h->cfg.use_ring || h->cfg.feature_cgcnn) ? rowA : 0);
  const size_t o_c = take(rowA), o_ctx = take(rowA), o_P1 = take(rowA), o_P3 = take(rowA), o_q = take(rowA);
  const size_t o_gq = take(rowA), o_gk = take(rowA), o_ga = take((size_t)A * 4), o_y = take((size_t)B * 4);
  const size_t o_pflag = take(pad ? 4 : 0);  // right behind y: fetched with the results in one copy
  const size_t o_pbuf = take((size_t)n_slot * 3 * D * 4);
  hipError_t e = hipSuccess;
  scann_handle::Stage* stage = nullptr;
  char* img_ptr = nullptr;
  if (scratch) {
    if (off > h->sc_cap) {  // grow-only (dynamic M, N: SURVEY 8b "workspace sized on first call and grown monotonically")
      if (h->sc_arena) { (void)hipStreamSynchronize(h->streams[0]); (void)hipFree(h->sc_arena); h->sc_arena = nullptr; h->sc_cap = 0; }
      const size_t want = off + off / 2;
      e = hipMalloc((void**)&h->sc_arena, want);
      if (e == hipSuccess) h->sc_cap = want;
    }
    if (e == hipSuccess && in_bytes > h->sc_host_cap) {
      if (h->sc_host) { (void)hipStreamSynchronize(h->streams[0]); (void)hipHostFree(h->sc_host); h->sc_host = nullptr; h->sc_host_cap = 0; }
      const size_t want = in_bytes + in_bytes / 2;
      e = hipHostMalloc((void**)&h->sc_host, want, hipHostMallocDefault);
      if (e == hipSuccess) h->sc_host_cap = want;
    }
    db->arena = h->sc_arena;
    img_ptr = h->sc_host;
  } else {
    e = cached_malloc((void**)&db->arena, off);
    if (e == hipSuccess && !h->copy_stream) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) {  // next pinned staging buffer of the ring: free once its previous copy has completed (normally long ago)
      stage = &h->stage[h->stage_next++ % scann_handle::N_STAGE];
      if (stage->used) (void)hipEventSynchronize(stage->ev);
      if (!stage->ev) e = hipEventCreateWithFlags(&stage->ev, hipEventDisableTiming);  // (the host waits on it: a default, fenced event)
      if (e == hipSuccess && in_bytes > stage->cap) {
        if (stage->p) (void)hipHostFree(stage->p);
        stage->p = nullptr; stage->cap = 0;
        const size_t want = in_bytes + in_bytes / 4;
        e = hipHostMalloc((void**)&stage->p, want, hipHostMallocDefault);
        if (e == hipSuccess) stage->cap = want;
      }
      img_ptr = stage->p;
    }
  }
  if (e != hipSuccess) {
    if (!scratch) {
      if (db->arena) cached_free(db->arena);
      delete db;
    }
    return fail(h, e == hipErrorOutOfMemory ? SCANN_ERR_OOM : SCANN_ERR_HIP, std::string("hipMalloc(batch arena): ") + hipGetErrorString(e));
  }
  struct ImgView { char* p; char* data() const { return p; } } img{img_ptr};
  if (b->atomic && !pad) memcpy(img.data() + o_atomic, b->atomic, (size_t)A * 4);
  if (pad) memcpy(img.data() + o_prow, pad->row_of, BM * 4);
  if (pad && !pre) {
    memcpy(img.data() + o_pat, pad->atomic, BM * 4);
    par_memcpy(img.data() + o_pnbr, pad->neighbors, BMN * 4);
    par_memcpy(img.data() + o_pmask, pad->neighbor_mask, BMN * pad->mask_size);
    par_memcpy(img.data() + o_pw, pad->weight, BMN * 4);
    par_memcpy(img.data() + o_pd, pad->dist, BMN * 4);
  }
  if (h->cfg.use_ring) memcpy(img.data() + o_ring, b->ring, (size_t)A * 2 * 4);
  if (h->cfg.feature_cgcnn) memcpy(img.data() + o_cg, b->cgcnn, (size_t)A * 92 * 4);
  memcpy(img.data() + o_mol, b->mol_offset, (size_t)(B + 1) * 4);
  memcpy(img.data() + o_eoff, b->edge_offset, (size_t)(A + 1) * 4);
  if (E > 0 && !pad) {
    memcpy(img.data() + o_col, b->edge_col, (size_t)E * 4);
    memcpy(img.data() + o_dist, b->edge_dist, (size_t)E * 4);
    memcpy(img.data() + o_wgt, b->edge_weight, (size_t)E * 4);
  }
  memcpy(img.data() + o_tiles, tiles.data(), tiles.size() * sizeof(EdgeTile));
  // reverse adjacency (counting sort of the edges by neighbour atom, stable): the backward pass sums per neighbour without atomics.
  // Only a handle in training mode (scann_train_begin) pays for it at upload; ensure_reverse builds it for a batch that was
  // uploaded before, on its first backward pass.
  const bool want_rev = h->t_master != nullptr;
  if (want_rev) {
    int32_t* in_off = reinterpret_cast<int32_t*>(img.data() + o_inoff);
    int32_t* in_edge = reinterpret_cast<int32_t*>(img.data() + o_inedge);
    memset(in_off, 0, (size_t)(A + 1) * 4);
    for (int e = 0; e < E; ++e) ++in_off[b->edge_col[e] + 1];
    for (int a = 0; a < A; ++a) in_off[a + 1] += in_off[a];
    std::vector<int32_t> fill(in_off, in_off + A);
    for (int e = 0; e < E; ++e) in_edge[fill[b->edge_col[e]]++] = e;
  }
  if (n_big) {
    memcpy(img.data() + o_tpart, tile_part.data(), tiles.size() * 4);
    memcpy(img.data() + o_big, big_tab.data(), (size_t)n_big * 3 * 4);
  }
  // (the centre atom of every edge is derived from the offsets on the device, behind the copy: no host loop, no bytes over the bus)
  int32_t* const d_eoff = (int32_t*)(db->arena + o_eoff);
  int32_t* const d_erow = (int32_t*)(db->arena + o_row);
  PackPaddedArgs pa{};
  if (pad) {
    char* a0 = db->arena;
    pa.B = B; pa.M = pad->M; pa.N = pad->N; pa.n_species = h->cfg.n_atoms;
    pa.row_of = (const int32_t*)(a0 + o_prow); pa.edge_offset = d_eoff; pa.atomic = pre ? pad->d_atomic : (const int32_t*)(a0 + o_pat);
    pa.neighbors = pre ? pad->d_neighbors : (const int32_t*)(a0 + o_pnbr); pa.neighbor_mask = pre ? pad->d_mask : a0 + o_pmask;
    pa.mask_size = pad->mask_size;
    pa.weight = pre ? pad->d_weight : (const float*)(a0 + o_pw); pa.dist = pre ? pad->d_dist : (const float*)(a0 + o_pd);
    pa.out_atomic = (int32_t*)(a0 + o_atomic); pa.out_col = (int32_t*)(a0 + o_col);
    pa.out_dist = (float*)(a0 + o_dist); pa.out_weight = (float*)(a0 + o_wgt);
    pa.flag = (int32_t*)(a0 + o_pflag);
  }
  // (the pack kernel's flag word is zeroed by edge_row_kernel, launched BEFORE it: no memset command of its own)
  if (scratch) {
    if (e == hipSuccess) e = hipMemcpyAsync(db->arena, img.data(), in_bytes, hipMemcpyHostToDevice, h->streams[0]);
    if (e == hipSuccess && (E > 0 || pad)) launch_edge_row(d_eoff, E > 0 ? A : 0, d_erow, h->streams[0], pad ? pa.flag : nullptr);
    if (e == hipSuccess && pad) launch_pack_padded(pa, h->streams[0]);
  } else {
    if (e == hipSuccess) e = hipMemcpyAsync(db->arena, img.data(), in_bytes, hipMemcpyHostToDevice, h->copy_stream);
    if (e == hipSuccess && (E > 0 || pad)) launch_edge_row(d_eoff, E > 0 ? A : 0, d_erow, h->copy_stream, pad ? pa.flag : nullptr);
    if (e == hipSuccess && pad) launch_pack_padded(pa, h->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(stage->ev, h->copy_stream);
    if (e == hipSuccess) stage->used = true;
    // a default (system-fenced) event: it orders a DMA engine's write into a REUSED arena (cached_malloc) before kernels on another
    // stream, whose caches may still hold lines of the arena's previous life -- not the place for the fence-free timing-event flavour
    if (e == hipSuccess) e = hipEventCreateWithFlags(&db->upload_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(db->upload_ev, h->copy_stream);
  }
  if (e != hipSuccess) {
    if (!scratch) {
      (void)hipStreamSynchronize(h->copy_stream);
      if (db->upload_ev) (void)hipEventDestroy(db->upload_ev);
      cached_free(db->arena);
      delete db;
    }
    return fail(h, SCANN_ERR_HIP, std::string("hipMemcpy(batch inputs): ") + hipGetErrorString(e));
  }
  char* a0 = db->arena;
  db->atomic = (int32_t*)(a0 + o_atomic); db->mol_offset = (int32_t*)(a0 + o_mol); db->edge_offset = (int32_t*)(a0 + o_eoff);
  db->edge_col = (int32_t*)(a0 + o_col); db->edge_row = (int32_t*)(a0 + o_row);
  db->dist = (float*)(a0 + o_dist); db->weight = (float*)(a0 + o_wgt); db->tiles = (EdgeTile*)(a0 + o_tiles);
  db->in_off = (int32_t*)(a0 + o_inoff); db->in_edge = (int32_t*)(a0 + o_inedge);
  db->has_rev = want_rev;
  db->ring = (float*)(a0 + o_ring); db->cgcnn = (float*)(a0 + o_cg); db->c0 = (float*)(a0 + o_c0);
  db->geom = (float*)(a0 + o_geom); db->gd = (float*)(a0 + o_gd);
  db->c = (float*)(a0 + o_c); db->ctx = (float*)(a0 + o_ctx); db->P1 = (float*)(a0 + o_P1); db->P3 = (float*)(a0 + o_P3);
  db->q = (float*)(a0 + o_q); db->gq = (float*)(a0 + o_gq); db->gk = (float*)(a0 + o_gk);
  db->ga = (float*)(a0 + o_ga); db->y = (float*)(a0 + o_y);
  db->pack_flag = pad ? (int32_t*)(a0 + o_pflag) : nullptr;
  if (n_big) {
    db->tile_part = (int32_t*)(a0 + o_tpart); db->big_tab = (int32_t*)(a0 + o_big); db->part_buf = (float*)(a0 + o_pbuf);
  }
  *out = db;
  return SCANN_OK;
}

int scann_batch_upload(scann_handle_t* h, const scann_batch_t* b, scann_dbatch_t** out) { return upload_impl(h, b, out, false); }

}  // extern "C"

namespace {

// Events that only time kernels on one stream: no system-scope cache write-back / invalidate when they fire (the HIP headers' own
// advice for timing events), so that a sampled launch is not lengthened by its own measurement.
constexpr unsigned kTimingEventFlags = hipEventDisableSystemFence;

struct Timer {
  hipStream_t s;
  bool on;
  std::vector<hipEvent_t> ev;
  std::vector<int> kind;
  void mark(int k) {
    if (!on) return;
    hipEvent_t e;
    (void)hipEventCreateWithFlags(&e, kTimingEventFlags);
    (void)hipEventRecord(e, s);
    ev.push_back(e);
    kind.push_back(k);
  }
};

int ensure_debug(scann_handle* h, scann_dbatch* db) {
  const int L = h->cfg.n_attention;
  if (db->dbg_layers == L) return SCANN_OK;
  HIPCHK(h, cached_malloc((void**)&db->dbg_c, (size_t)(L + 1) * db->n_atom * D * 4));
  HIPCHK(h, cached_malloc((void**)&db->dbg_ctx, (size_t)std::max(L, 1) * db->n_atom * D * 4));
  if (h->cfg.g_update) HIPCHK(h, cached_malloc((void**)&db->dbg_g, (size_t)(L + 1) * std::max(db->n_edge, 1) * D * 4));
  db->dbg_layers = L;
  return SCANN_OK;
}

// What the generic-width TRAINING forward keeps for the backward (gen_backward): every tensor a formula's derivative reads, in buffers of
// their own per layer (the inference forward rotates five atom-row and three edge-row buffers instead).
struct GenLayerKeep {
  float *cc_in = nullptr, *G_in = nullptr;  // centres / geometry entering the layer
  float *Z = nullptr;                       // filter_geo pre-activation
  float *T = nullptr;                       // g_update: swish(Z) + G_in, the input of layer_norm_g
  float *Gn = nullptr;                      // the geometry the key projection is gated with (g_update: the layer's output geometry)
  float *K = nullptr, *q = nullptr;
  float *t1 = nullptr, *ctx = nullptr;      // attention context + query (input of layer_norm), its LayerNorm
  float *pre1 = nullptr, *h1 = nullptr, *t2 = nullptr;  // ResidualNorm: dense_1 pre-activation, its swish, Dropout(dense_2)
  float *cc_out = nullptr;
};
struct GenKeep {
  char* arena = nullptr;   // forward tensors
  size_t bytes = 0;
  char* barena = nullptr;  // backward temporaries + the transposed kernels
  size_t bbytes = 0;
  std::vector<GenLayerKeep> layer;
  float *embE = nullptr, *ring10 = nullptr, *pre_e = nullptr, *cc0 = nullptr;
  float *gd = nullptr, *gw = nullptr, *pre_d = nullptr, *pre_w = nullptr, *Td = nullptr, *Tw = nullptr, *G0 = nullptr;
  float *cc_L = nullptr, *z_pre = nullptr, *z = nullptr, *gq = nullptr, *gk = nullptr, *rep = nullptr, *hid_pre = nullptr, *hid = nullptr;
  float drop_p = 0.f, attn_p = 0.f;
  unsigned long long seed = 0;
  std::map<std::string, std::pair<const float*, size_t>> dbg;  // scann_train_debug_read: tensors of the last backward's readout stage
};

// create_model (scann_model.py:362-447) for a handle whose widths are not 128 / 8: one plain-fp32 kernel per formula
// (scann_generic.hip) on the same packed batch.  kp non-null: the training forward -- Dropout layers active (kp->drop_p, kp->attn_p,
// kp->seed), every intermediate kept in kp, the property head as dense launches.
int run_forward_generic(scann_handle* h, scann_dbatch* db, hipStream_t s, GenKeep* kp = nullptr) {
  const scann_config_t& c = h->cfg;
  const int L = c.n_attention, A = db->n_atom, E = db->n_edge, B = db->n_struct;
  const int d = c.local_dim, dg = c.global_dim, dout = c.dense_out, H = c.num_head, emb = c.embedding_dim;
  const int cin = emb + (c.use_ring ? 10 : 0);
  if (kp) kp->dbg.clear();  // (scann_train_debug_read: the tensors it names belong to the backward of THIS forward's arena)
  if ((size_t)std::max(1, db->max_degree) * H * 4 * (kp ? 3 : 1) > 60000 || ((size_t)db->max_atoms * (kp ? 3 : 1) + dg + dout + 4) * 4 > 60000 ||
      (size_t)4 * 3 * d * 4 > 60000)
    return fail(h, SCANN_ERR_UNSUPPORTED, "forward (generic widths): an atom's neighbours x heads, or a structure's atoms, exceed one workgroup's LDS");
  auto W = [&](const std::string& name) -> const float* { return h->g_weights + h->g_off.at(name); };
  // workspace: atom rows, edge rows, Gaussian bases
  const size_t fA = (size_t)A, fE = (size_t)std::max(E, 1), fB = (size_t)B, Ls = (size_t)L;
  float* p = nullptr;
  char* p_end = nullptr;
  if (!kp) {
    const size_t need = 4 * (fA * (5 * (size_t)d + (size_t)cin + (size_t)emb + 10 + 3 * (size_t)dg) + fE * (3 * (size_t)d + 2 * NG)) + 4096;
    if (db->gen_ws_bytes < need) {
      HIPCHK(h, hipStreamSynchronize(s));
      cached_free(db->gen_ws);
      db->gen_ws = nullptr;
      db->gen_ws_bytes = 0;
      HIPCHK(h, cached_malloc((void**)&db->gen_ws, need));
      db->gen_ws_bytes = need;
    }
    p = reinterpret_cast<float*>(db->gen_ws);
    p_end = db->gen_ws + db->gen_ws_bytes;
  } else {
    const size_t need = 4 * (fA * ((size_t)emb + 10 + (2 + 7 * Ls) * (size_t)d + 4 * (size_t)dg) + fE * (2 * NG + (5 + 4 * Ls) * (size_t)d) +
                             fB * ((size_t)dg + 2 * (size_t)dout + 1)) + 256 * (32 + 12 * Ls);
    if (kp->bytes < need) {
      HIPCHK(h, hipStreamSynchronize(s));
      cached_free(kp->arena);
      kp->arena = nullptr;
      kp->bytes = 0;
      HIPCHK(h, cached_malloc((void**)&kp->arena, need));
      kp->bytes = need;
    }
    p = reinterpret_cast<float*>(kp->arena);
    p_end = kp->arena + kp->bytes;
    kp->layer.assign((size_t)L, GenLayerKeep{});
  }
  auto take = [&](size_t n) { float* q = p; p += (n + 63) & ~(size_t)63; return q; };
  float *cc = take(fA * d), *ctx = nullptr, *t1 = nullptr, *t2 = nullptr, *q = nullptr;
  if (!kp) { ctx = take(fA * d); t1 = take(fA * d); t2 = take(fA * d); q = take(fA * d); }
  float *embE = take(fA * emb), *ring10 = take(fA * 10);
  float *z = take(fA * dg), *gq = take(fA * dg), *gk = take(fA * dg);
  float *G = take(fE * d), *T = nullptr, *K = nullptr, *gd = take(fE * NG), *gw = take(fE * NG);
  if (!kp) { T = take(fE * d); K = take(fE * d); }
  const float tp = kp ? kp->drop_p : 0.f;
  const unsigned long long seed = kp ? kp->seed : 0;
  auto dense = [&](GenSeg s0, GenSeg s1, GenSeg s2, int n_seg, int prod, const std::string& name, int K_, int N_, int rows, int act,
                   const float* res, const float* row_scale, float* Y, float* pre = nullptr, float drop_p = 0.f, unsigned drop_tag = 0) {
    GenDenseArgs a{};
    a.seg[0] = s0; a.seg[1] = s1; a.seg[2] = s2; a.n_seg = n_seg; a.prod = prod;
    a.W = W(name + "/kernel"); a.b = W(name + "/bias"); a.K = K_; a.N = N_; a.rows = rows; a.act = act;
    a.res = res; a.res_idx = nullptr; a.row_scale = row_scale; a.Y = Y;
    a.pre = pre; a.drop_p = drop_p; a.drop_tag = drop_tag; a.drop_seed = seed;
    launch_gen_dense(a, s);
  };
  const GenSeg none{nullptr, nullptr, 0};
  // ---- embedding (scann_model.py:362-374; Dropout(0.1) on the centres in training, :374) ----
  float* pre_e = kp ? take(fA * d) : nullptr;
  GenSeg e0;
  if (c.feature_cgcnn) {
    dense(GenSeg{db->cgcnn, nullptr, 92}, none, none, 1, 0, "embed_atom", 92, emb, A, 0, nullptr, nullptr, embE);
    e0 = GenSeg{embE, nullptr, emb};
  } else {
    e0 = GenSeg{W("embed_atom/embeddings"), db->atomic, emb};
  }
  if (c.use_ring) {
    dense(GenSeg{db->ring, nullptr, 2}, none, none, 1, 0, "extra_embed", 2, 10, A, 0, nullptr, nullptr, ring10);
    dense(e0, GenSeg{ring10, nullptr, 10}, none, 2, 0, "dense_embed", emb + 10, d, A, 1, nullptr, nullptr, cc, pre_e, tp, DROP_TAG_EMBED);
  } else {
    dense(e0, none, none, 1, 0, "dense_embed", emb, d, A, 1, nullptr, nullptr, cc, pre_e, tp, DROP_TAG_EMBED);
  }
  if (kp) { kp->embE = embE; kp->ring10 = ring10; kp->pre_e = pre_e; kp->cc0 = cc; kp->gd = gd; kp->gw = gw; }
  // ---- Gaussian bases and the initial geometry (scann_model.py:376-391) ----
  launch_gen_gauss(db->dist, h->g_centres, E, gd, s);
  if (c.g_update) {
    launch_gen_gauss(db->weight, h->g_centres + NG, E, gw, s);
    float *Td = kp ? take(fE * d) : T, *Tw = kp ? take(fE * d) : K;
    float *pre_d = kp ? take(fE * d) : nullptr, *pre_w = kp ? take(fE * d) : nullptr;
    dense(GenSeg{gd, nullptr, NG}, none, none, 1, 0, "neighbor_d", NG, d, E, 1, nullptr, nullptr, Td, pre_d);
    dense(GenSeg{gw, nullptr, NG}, none, none, 1, 0, "neighbor_w", NG, d, E, 1, nullptr, nullptr, Tw, pre_w);
    launch_gen_mul(Td, Tw, (size_t)E * d, G, s);
    if (kp) { kp->Td = Td; kp->Tw = Tw; kp->pre_d = pre_d; kp->pre_w = pre_w; kp->G0 = G; }
  }
  // ---- LocalAttention iterations (scann_model.py:413-421; attention.py:118-216, :37-40) ----
  for (int l = 0; l < L; ++l) {
    const std::string la = "local_attention_" + std::to_string(l), rn = "residual_norm_" + std::to_string(l);
    GenLayerKeep b;
    b.cc_in = cc; b.G_in = G;
    if (kp) {
      b.Z = take(fE * d); b.Gn = take(fE * d); b.K = take(fE * d);
      if (c.g_update) b.T = take(fE * d);
      b.q = take(fA * d); b.t1 = take(fA * d); b.ctx = take(fA * d);
      if (c.use_attn_norm) { b.pre1 = take(fA * d); b.h1 = take(fA * d); b.t2 = take(fA * d); b.cc_out = take(fA * d); }
      else b.cc_out = b.ctx;
    } else {
      b.T = T; b.Gn = c.g_update ? G : T; b.K = K; b.q = q; b.t1 = t1; b.ctx = ctx; b.h1 = t1; b.t2 = t2;
      b.cc_out = c.use_attn_norm ? cc : ctx;
    }
    if (c.g_update) {
      dense(GenSeg{cc, db->edge_row, d}, GenSeg{G, nullptr, d}, GenSeg{cc, db->edge_col, d}, 3, 0, la + "/filter_geo", 3 * d, d, E, 1, G, nullptr, b.T, b.Z);
      launch_gen_layernorm(b.T, nullptr, W(la + "/layer_norm_g/gamma"), W(la + "/layer_norm_g/beta"), E, d, b.Gn, s);
    } else {
      dense(GenSeg{gd, nullptr, NG}, none, none, 1, 0, la + "/filter_geo", NG, d, E, 1, nullptr, db->weight, b.Gn, b.Z);
    }
    dense(GenSeg{cc, db->edge_col, d}, GenSeg{b.Gn, nullptr, d}, none, 2, 1, la + "/key", d, d, E, 0, nullptr, nullptr, b.K);
    dense(GenSeg{cc, nullptr, d}, none, none, 1, 0, la + "/query", d, d, A, 0, nullptr, nullptr, b.q);
    launch_gen_attn(b.q, b.K, db->edge_offset, A, d, H, db->max_degree, b.t1, s, kp ? kp->attn_p : 0.f, DROP_TAG_ATTN + (unsigned)l, seed);
    launch_gen_layernorm(b.t1, nullptr, W(la + "/layer_norm/gamma"), W(la + "/layer_norm/beta"), A, d, b.ctx, s);
    if (c.use_attn_norm) {  // ResidualNorm (attention.py:37-40): LayerNorm(x + Dropout(dense_2(swish(dense_1 x))))
      dense(GenSeg{b.ctx, nullptr, d}, none, none, 1, 0, rn + "/dense_1", d, d, A, 1, nullptr, nullptr, b.h1, b.pre1);
      dense(GenSeg{b.h1, nullptr, d}, none, none, 1, 0, rn + "/dense_2", d, d, A, 0, nullptr, nullptr, b.t2, nullptr, tp, (unsigned)l);
      launch_gen_layernorm(b.ctx, b.t2, W(rn + "/layer_norm/gamma"), W(rn + "/layer_norm/beta"), A, d, b.cc_out, s);
    }
    if (kp) {
      kp->layer[(size_t)l] = b;
      cc = b.cc_out;
      if (c.g_update) G = b.Gn;
    } else if (!c.use_attn_norm) {
      std::swap(cc, ctx);
    }
  }
  // ---- readout (scann_model.py:424-447; attention.py:267-318) ----
  float* z_pre = kp ? take(fA * dg) : nullptr;
  dense(GenSeg{cc, nullptr, d}, none, none, 1, 0, "after_Lc", d, dg, A, 1, nullptr, nullptr, z, z_pre);
  dense(GenSeg{z, nullptr, dg}, none, none, 1, 0, "global_attention/query", dg, dg, A, 0, nullptr, nullptr, gq);
  dense(GenSeg{z, nullptr, dg}, none, none, 1, 0, "global_attention/key", dg, dg, A, 0, nullptr, nullptr, gk);
  float* rep = kp ? take(fB * dg) : nullptr;
  launch_gen_readout(db->mol_offset, B, db->max_atoms, gq, gk, dg, dout, c.use_ga_norm, c.relu_out, W("bf_property/kernel"), W("bf_property/bias"),
                     W("predict_property/kernel"), W("predict_property/bias"), db->ga, db->y, s, rep);
  if (kp) {
    float *hid_pre = take(fB * dout), *hid = take(fB * dout);
    dense(GenSeg{rep, nullptr, dg}, none, none, 1, 0, "bf_property", dg, dout, B, 1, nullptr, nullptr, hid, hid_pre);
    dense(GenSeg{hid, nullptr, dout}, none, none, 1, 0, "predict_property", dout, 1, B, 0, nullptr, nullptr, db->y);
    if (c.relu_out) launch_gen_relu(db->y, B, s);  // mrelu forward (custom_layers.py:15); its gradient is the identity
    kp->cc_L = cc; kp->z_pre = z_pre; kp->z = z; kp->gq = gq; kp->gk = gk; kp->rep = rep; kp->hid_pre = hid_pre; kp->hid = hid;
  }
  if (reinterpret_cast<char*>(p) > p_end) return fail(h, SCANN_ERR_HIP, "forward (generic widths): workspace overrun");
  HIPCHK(h, hipGetLastError());
  return SCANN_OK;
}

// The forward graph of create_model (scann_model.py:362-447) as a launch schedule on one stream.
// kind codes for the timer: 0 basis, 1 atom, 2 edge, 3 readout.
int run_forward(scann_handle* h, scann_dbatch* db, hipStream_t s, Timer* tm, bool exact = false) {
  if ((h->force_exact || h->weights_exact) && !h->debug && !h->in_train_forward) exact = true;
  db->idle = false;  // work is being enqueued on the batch (scann_batch_release)
  if (!h->loaded) return fail(h, SCANN_ERR_WEIGHTS, "forward: weights not loaded");
  HIPCHK(h, wait_upload(db, s));  // the inputs' copy (scann_batch_upload returned when it was enqueued)
  if (h->generic) {
    if (tm) { tm->mark(-1); }
    const int r = run_forward_generic(h, db, s, h->in_train_forward ? h->gen_keep : nullptr);
    if (tm) tm->mark(3);
    return r;
  }
  const scann_config_t& c = h->cfg;
  const int L = c.n_attention;
  const size_t rowA = (size_t)db->n_atom * D * 4, rowE = (size_t)db->n_edge * D * 4;
  if (h->debug) {
    const int r = ensure_debug(h, db);
    if (r) return r;
  }
  // keep-mode (training / scann_set_debug): every layer writes its centres, context and geometry straight into its slice of
  // the per-layer buffers (base branch: no geometry to thread)
  const bool direct = h->debug;
  const size_t nA_ = (size_t)db->n_atom * D, nE_ = (size_t)db->n_edge * D;
  auto c_of = [&](int l) { return direct ? db->dbg_c + (size_t)l * nA_ : db->c; };
  auto ctx_of = [&](int l) { return direct ? db->dbg_ctx + (size_t)l * nA_ : db->ctx; };
  auto g_of = [&](int l) { return direct && c.g_update ? db->dbg_g + (size_t)l * nE_ : db->geom; };
  int32_t* const rflag = h->range_flag ? h->range_flag + db->last_slot : nullptr;  // this stream's range-guard word
  if (tm) tm->mark(-1);
  // exact: the forward's range guard fired (an operand outside the split-fp16 range): the same launches on the EX instantiations of
  // the atom / edge kernels -- exact-fp32 projections -- with the plain first layer (basis_kernel, no per-species tables)
  // inference: the first layer's edge kernel computes its geometry rows from (dist, weight) itself -- geom0 is never written by a
  // basis launch and read back (282 MB of the 16-batch forward's traffic and one launch)
  const bool fuse_basis = !exact && h->fuse_basis && c.g_update && L > 0 && !h->debug && !h->in_train_forward && db->n_edge > 0;
  if (fuse_basis) {}  // (nothing to launch)
  else if (c.g_update) launch_basis(h->basis, db->dist, db->weight, db->n_edge, g_of(0), s);
  else launch_basis_raw(h->cd, db->dist, db->n_edge, db->gd, s);
  if (tm) tm->mark(0);
  if (h->debug && !direct && c.g_update && db->n_edge) HIPCHK(h, hipMemcpyAsync(db->dbg_g, db->geom, rowE, hipMemcpyDeviceToDevice, s));

  const bool general_embed = c.use_ring || c.feature_cgcnn;
  if (general_embed) {
    EmbedArgs e = h->embed;
    e.n_atom = db->n_atom; e.atomic = db->atomic; e.c0 = db->c0;
    e.ring = c.use_ring ? db->ring : nullptr;
    e.cgcnn = c.feature_cgcnn ? db->cgcnn : nullptr;
    launch_embed(e, s);
    if (tm) tm->mark(0);
  }
  // first layer from per-species tables: no atom launch at all (see EdgeArgs::species)
  // (not with chunked atoms: edge_merge_kernel reads the query rows per atom)
  const bool species0 = fuse_basis && h->species_tables && !general_embed && h->train_drop_p == 0.f && h->sp_c && db->n_big == 0;
  if (species0 && h->sp_dirty) {
    AtomArgs a{};
    a.n_atom = c.n_atoms; a.x = h->lut; a.ffn = 0; a.c = h->sp_c;
    a.range_flag = rflag; a.layer = 0;
    const LayerParams& p = h->layers[0];
    a.mode = 0;
    a.WAh = p.W1h; a.bA = p.bg; a.WBh = p.W3h; a.WCh = p.Wqh; a.bC = p.bq;
    a.oA = h->sp_P1; a.oB = h->sp_P3; a.oC = h->sp_q;
    launch_atom(a, s);
    HIPCHK(h, hipStreamSynchronize(s));  // once per weight change: forwards on the handle's other streams read the tables too
    h->sp_dirty = false;
  }
  for (int l = 0; l <= L; ++l) {
    // training forward through edge_kernel_lean: q, V, T, ang, K of every layer are kept for the backward
    const bool keep = direct && h->in_train_forward && db->keep_K && l < L;
    // atom kernel at the head of layer l: ResidualNorm of layer l-1, centres, projections of layer l
    AtomArgs a{};
    a.n_atom = db->n_atom;
    if (l == 0) {
      a.x = general_embed ? db->c0 : h->lut;
      a.x_index = general_embed ? nullptr : db->atomic;
      a.ffn = 0;
    } else {
      a.x = ctx_of(l - 1);
      a.x_index = nullptr;
      a.ffn = c.use_attn_norm ? 1 : 0;
      const LayerParams& pp = h->layers[l - 1];
      a.Wf1h = pp.Wf1h; a.bf1 = pp.bf1; a.Wf2h = pp.Wf2h; a.bf2 = pp.bf2; a.lnr_g = pp.lnr_g; a.lnr_b = pp.lnr_b;
      if (a.ffn && direct && h->in_train_forward && db->keep_T2) {
        a.keep_pre1 = db->keep_pre1 + (size_t)(l - 1) * nA_; a.keep_H1 = db->keep_H1 + (size_t)(l - 1) * nA_;
        a.keep_T2 = db->keep_T2 + (size_t)(l - 1) * nA_;
      }
    }
    a.c = c_of(l);
    a.range_flag = rflag; a.layer = l;
    if (h->train_drop_p > 0.f) {  // training-mode Dropout(0.1) layers (scann_model.py:374, attention.py:29)
      a.drop_p = (l == 0 || c.use_attn_norm) ? h->train_drop_p : 0.f;
      a.drop_seed = h->train_seed;
      a.drop_tag = l == 0 ? DROP_TAG_EMBED : (unsigned)(l - 1);
    }
    if (l < L) {
      const LayerParams& p = h->layers[l];
      a.mode = c.g_update ? 0 : 1;
      a.WAh = p.W1h; a.bA = p.bg; a.WBh = p.W3h; a.WCh = p.Wqh; a.bC = p.bq;
      a.oA = db->P1; a.oB = db->P3; a.oC = keep ? db->keep_q + (size_t)l * nA_ : db->q;
    } else {
      a.mode = 2;
      a.WAh = h->head.Wah; a.bA = h->head.ba; a.WCh = h->head.Wgqh; a.bC = h->head.bgq; a.WDh = h->head.Wgkh; a.bD = h->head.bgk;
      a.oB = db->gk; a.oC = db->gq;
      if (direct && h->in_train_forward && db->keep_preA) { a.keep_preA = db->keep_preA; a.keep_z = db->keep_z; }
    }
#ifdef SCANN_STAMPS
    if (getenv("SCANN_STAMP_ATOM") && l >= 1 && l < L) {  // phase clocks of atom_kernel<true, 0> (the last such launch wins)
      const int nt = (db->n_atom + 31) / 32;  // 32- or 64-row tiles (launch_atom): room for either
      if (!db->stamps) HIPCHK(h, hipMalloc((void**)&db->stamps, (size_t)nt * 16 * sizeof(unsigned long long)));
      a.stamps = db->stamps;
      db->n_stamp = nt;
    }
#endif
    if (exact) {  // fp32 fragment-order images in place of the split-fp16 ones
      a.exact = 1;
      if (l > 0 && a.ffn) {
        const LayerParams& pp = h->layers[l - 1];
        a.Wf1h = reinterpret_cast<const _Float16*>(pp.Wf1p); a.Wf2h = reinterpret_cast<const _Float16*>(pp.Wf2p);
      }
      if (l < L) {
        const LayerParams& p = h->layers[l];
        a.WAh = reinterpret_cast<const _Float16*>(p.W1p); a.WBh = reinterpret_cast<const _Float16*>(p.W3p);
        a.WCh = reinterpret_cast<const _Float16*>(p.Wqp);
      } else {
        a.WAh = reinterpret_cast<const _Float16*>(h->head.Wap); a.WCh = reinterpret_cast<const _Float16*>(h->head.Wgqp);
        a.WDh = reinterpret_cast<const _Float16*>(h->head.Wgkp);
      }
    }
    if (!(species0 && l == 0)) launch_atom(a, s);
    if (tm) tm->mark(l < L ? 1 : 3);
    if (h->debug && !direct) HIPCHK(h, hipMemcpyAsync(db->dbg_c + (size_t)l * db->n_atom * D, db->c, rowA, hipMemcpyDeviceToDevice, s));
    if (l == L) break;
    EdgeArgs ea{};
    ea.tiles = db->tiles; ea.n_tile = db->n_tile; ea.g_update = c.g_update; ea.tile_rows = db->tile_rows;
    ea.edge_offset = db->edge_offset; ea.edge_col = db->edge_col; ea.edge_row = db->edge_row;
    ea.geom = g_of(l); ea.geom_out = direct && c.g_update ? g_of(l + 1) : nullptr; ea.gd = db->gd; ea.edge_weight = db->weight;
    if (fuse_basis && l == 0) { ea.fuse_basis = 1; ea.dist = db->dist; ea.basis = h->basis; }
    ea.n_edge = db->n_edge;
    ea.geom_rows = fuse_basis ? 0 : 1;  // piece-major tiles only when the first layer computed its own geometry rows (plain inference)
    ea.geom_dead = (l == L - 1 && !h->debug) ? 1 : 0;  // the geometry leaving the last layer is never consumed (141 MB of writes per 16-batch launch)
    ea.c = c_of(l); ea.P1 = db->P1; ea.P3 = db->P3; ea.q = keep ? db->keep_q + (size_t)l * nA_ : db->q; ea.ctx = ctx_of(l);
    if (species0 && l == 0) { ea.species = db->atomic; ea.c = h->sp_c; ea.P1 = h->sp_P1; ea.P3 = h->sp_P3; ea.q = h->sp_q; }
    if (keep) {
      ea.keep_V = db->keep_V + (size_t)l * nE_; ea.keep_K = db->keep_K + (size_t)l * nE_;
      // T = swish(V) + G and ang = c[j] * G' are formed again where the fused backward needs them (edge_bwd_kernel, the key weight
      // gradient's operand load): two of the six [n_edge,128] streams of the training forward's edge launch
      if (db->keep_T) ea.keep_T = db->keep_T + (size_t)l * nE_;
      if (db->keep_ang) ea.keep_ang = db->keep_ang + (size_t)l * nE_;
      db->kept = true;
    }
    ea.p = h->layers[l];
    if (exact) {
      ea.exact = 1;
      ea.p.W2h = reinterpret_cast<const _Float16*>(ea.p.W2p); ea.p.Wkh = reinterpret_cast<const _Float16*>(ea.p.Wkp);
    }
    ea.range_flag = rflag; ea.layer = l;
    // (the first layer's launch with the basis MLP fused in is a different kernel: not part of edge_kernel's sampled average)
    // (... nor is the last layer's, whose geometry is not stored -- the DEAD instantiation, ~10 % shorter: the sampled average is the
    //  kernel rocprofv3 lists as edge_kernel<true, RT, false, false, false, false>, and its algorithmic bytes include that store)
    const bool sample = !tm && h->time_every > 0 && (h->time_count % h->time_every) == 0 && !(fuse_basis && l == 0) && !(ea.geom_dead && L > 2);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (sample) {
      (void)hipEventCreateWithFlags(&ev0, kTimingEventFlags);
      (void)hipEventCreateWithFlags(&ev1, kTimingEventFlags);
      (void)hipEventRecord(ev0, s);
    }
    ea.tile_part = db->tile_part; ea.part_buf = db->part_buf;
    ea.xcd_remap = h->xcd_remap;
    if (h->in_train_forward && h->attn_drop_p > 0.f) {  // validation passes run with scann_set_attention_dropout(h, 0): trainer.fit
      ea.attn_drop_p = h->attn_drop_p;
      ea.attn_drop_seed = h->train_seed;
      ea.attn_drop_tag = DROP_TAG_ATTN + (unsigned)l;
    }
#ifdef SCANN_STAMPS
    if (!getenv("SCANN_STAMP_ATOM") && l == (getenv("SCANN_STAMP_LAYER") ? atoi(getenv("SCANN_STAMP_LAYER")) : L - 1)) {  // one launch's picture
      if (!db->stamps) HIPCHK(h, hipMalloc((void**)&db->stamps, (size_t)db->n_tile * 16 * sizeof(unsigned long long)));
      ea.stamps = db->stamps;
      db->n_stamp = db->n_tile;
    }
#endif
    launch_edge(ea, s);
    launch_edge_merge(db->big_tab, db->n_big, db->part_buf, ea.q, ea.p.ln_g, ea.p.ln_b, ea.ctx, rflag, l, s);
    if (sample) {
      (void)hipEventRecord(ev1, s);
      h->time_ev.push_back(ev0);
      h->time_ev.push_back(ev1);
      h->time_edges.push_back(db->n_edge);
    }
    if (tm) tm->mark(2);
    if (h->debug && !direct) {
      HIPCHK(h, hipMemcpyAsync(db->dbg_ctx + (size_t)l * db->n_atom * D, db->ctx, rowA, hipMemcpyDeviceToDevice, s));
      if (c.g_update && db->n_edge)
        HIPCHK(h, hipMemcpyAsync(db->dbg_g + (size_t)(l + 1) * db->n_edge * D, db->geom, rowE, hipMemcpyDeviceToDevice, s));
    }
  }
  if (!tm) h->time_count++;
  ReadoutArgs r{};
  r.mol_offset = db->mol_offset; r.n_struct = db->n_struct; r.max_atoms = db->max_atoms;
  r.gq = db->gq; r.gk = db->gk; r.use_ga_norm = c.use_ga_norm; r.relu_out = c.relu_out;
  r.p = h->head; r.ga_attn = db->ga; r.y = db->y;
  launch_readout(r, s);
  if (tm) tm->mark(3);
  HIPCHK(h, hipGetLastError());
  return SCANN_OK;
}

}  // namespace

extern "C" {

int scann_forward_resident(scann_handle_t* h, scann_dbatch_t* db, int stream_slot) {
  if (!h || !db) return fail(h, SCANN_ERR_INVALID, "scann_forward_resident: null argument");
  HIPCHK(h, hipSetDevice(h->device));
  const int slot = ((stream_slot % h->nstream) + h->nstream) % h->nstream;
  db->last_slot = slot;
  return run_forward(h, db, h->streams[slot], nullptr);
}

int scann_batch_info(scann_handle_t* h, const scann_dbatch_t* db, int32_t* out8) {
  if (!h || !db || !out8) return fail(h, SCANN_ERR_INVALID, "scann_batch_info: null argument");
  out8[0] = db->n_struct; out8[1] = db->n_atom; out8[2] = db->n_edge; out8[3] = db->n_big;
  out8[4] = db->n_slot; out8[5] = db->max_degree; out8[6] = db->n_tile; out8[7] = db->tile_rows;
  return SCANN_OK;
}

// y (and the GlobalAttention scores) of the batch's last forward -> the caller's arrays: one D2H into the slot's pinned block, then
// plain memcpy (two hipMemcpyAsync into pageable numpy arrays were two staged copies: 23 us of a one-batch call's 280;
// polling the stream before the blocking wait changed nothing: hipStreamSynchronize already spins for waits this short)
static int fetch_results(scann_handle_t* h, scann_dbatch_t* db, hipStream_t s, float* y_out, float* ga_attn_out) {
  const char* src = reinterpret_cast<const char*>(ga_attn_out ? db->ga : db->y);
  const size_t y_off = (size_t)(reinterpret_cast<const char*>(db->y) - src);
  const size_t f_off = db->pack_flag ? (size_t)(reinterpret_cast<const char*>(db->pack_flag) - src) : 0;  // (behind y in the arena)
  const size_t bytes = db->pack_flag ? f_off + 4 : y_off + (size_t)db->n_struct * 4;
  scann_handle::DlStage& st = h->dl_stage[db->last_slot];
  if (st.cap < bytes) {
    if (st.p) {
      HIPCHK(h, hipStreamSynchronize(s));
      (void)hipHostFree(st.p);
      st.p = nullptr;
      st.cap = 0;
    }
    const size_t cap = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 16);
    HIPCHK(h, hipHostMalloc((void**)&st.p, cap, hipHostMallocDefault));
    st.cap = cap;
  }
  HIPCHK(h, hipMemcpyAsync(st.p, src, bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  if (db->pack_flag) {  // a batch packed on the device: what the host packer refuses when it packs, the kernel reports here
    const int32_t bad = *reinterpret_cast<const int32_t*>(st.p + f_off);
    if (bad & 1) return fail(h, SCANN_ERR_INVALID, "scann_batch_download: an unmasked neighbour slot points at a padded atom (or outside the structure)");
    if (bad & 2) return fail(h, SCANN_ERR_INVALID, "scann_batch_download: atomic number outside the embedding table (n_atoms)");
  }
  memcpy(y_out, st.p + y_off, (size_t)db->n_struct * 4);
  if (ga_attn_out) memcpy(ga_attn_out, st.p, (size_t)db->n_atom * 4);
  return SCANN_OK;
}

int scann_batch_download(scann_handle_t* h, scann_dbatch_t* db, float* y_out, float* ga_attn_out) {
  if (!h || !db || !y_out) return fail(h, SCANN_ERR_INVALID, "scann_batch_download: null argument");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = h->streams[db->last_slot];
  const int rf = fetch_results(h, db, s, y_out, ga_attn_out);
  if (rf) return rf;
  // The forward's range guard fired: an activation left the range of the split-fp16 projections (sites 1-4).  The reference runs any
  // fp32 values (attention.py:95-113), so the forward is run again on the exact-fp32 instantiations (1/16 of the matrix rate, this
  // batch only) instead of handing an error back -- unless SCANN_STRICT_RANGE=1 asks for the error.
  if (h->range_flag && !h->strict_range && !db->kept) {
    const int32_t code = *reinterpret_cast<volatile int32_t*>(h->range_flag + db->last_slot);
    const int site = code >> 8;
    if (code && site >= 1 && site <= 4) {
      h->range_flag[db->last_slot] = 0;
      const int r = run_forward(h, db, s, nullptr, true);
      if (r) return r;
      h->exact_reruns++;
      const int rf3 = fetch_results(h, db, s, y_out, ga_attn_out);
      if (rf3) return rf3;
    }
  }
  db->idle = true;
  return check_range(h, "scann_batch_download", db->last_slot);
}

int scann_device_memory(scann_handle_t* h, int64_t* free_bytes, int64_t* total_bytes) {
  if (!h || !free_bytes || !total_bytes) return fail(h, SCANN_ERR_INVALID, "scann_device_memory: null argument");
  HIPCHK(h, hipSetDevice(h->device));
  size_t f = 0, t = 0;
  HIPCHK(h, hipMemGetInfo(&f, &t));
  *free_bytes = (int64_t)f;
  *total_bytes = (int64_t)t;
  return SCANN_OK;
}

int64_t scann_exact_reruns(const scann_handle_t* h) { return h ? h->exact_reruns : -1; }

int scann_sync(scann_handle_t* h) {
  if (!h) return SCANN_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  for (int i = 0; i < h->nstream; ++i) HIPCHK(h, hipStreamSynchronize(h->streams[i]));
  return SCANN_OK;
}

int scann_forward(scann_handle_t* h, const scann_batch_t* batch, float* y_out, float* ga_attn_out) {
  // synchronous convenience path: the batch lives in the handle's reusable scratch (no hipMalloc per call)
  scann_dbatch_t* db = nullptr;
  int r = upload_impl(h, batch, &db, true);
  if (r) return r;
  r = scann_forward_resident(h, db, 0);
  if (!r) r = scann_batch_download(h, db, y_out, ga_attn_out);
  return r;
}

int scann_forward_padded(scann_handle_t* h, int32_t B, int32_t M, int32_t N, const int32_t* atomic, const uint8_t* atom_mask,
                         const int32_t* neighbors, const uint8_t* neighbor_mask, const float* neighbor_weight,
                         const float* neighbor_distance, float* y_out, float* ga_out) {
  if (!h || B <= 0 || M <= 0 || N < 0 || !atomic || !atom_mask || !y_out || (N > 0 && (!neighbors || !neighbor_mask || !neighbor_weight || !neighbor_distance)))
    return fail(h, SCANN_ERR_INVALID, "scann_forward_padded: bad argument");
  if (h->cfg.use_ring || h->cfg.feature_cgcnn) return fail(h, SCANN_ERR_UNSUPPORTED, "scann_forward_padded: atomic feature without ring only");
  const size_t BM = (size_t)B * M;
  // the handle's own packing buffers, grown when a call needs more: seven fresh vectors per call were 0.6 MB of mmap + page faults +
  // zero fill at the reference's batch size -- a good part of what a one-batch call spends before the device can start
  scann_handle::PadScratch& ps = h->pad_scratch;
  auto grow_i = [](std::vector<int32_t>& v, size_t n) { if (v.size() < n) v.resize(n + n / 4); };
  auto grow_f = [](std::vector<float>& v, size_t n) { if (v.size() < n) v.resize(n + n / 4); };
  grow_i(ps.gidx, BM); grow_i(ps.mol, (size_t)B + 1); grow_i(ps.eoff, BM + 1);
  std::vector<int32_t>&gidx = ps.gidx, &mol = ps.mol, &eoff = ps.eoff;
  int32_t na = 0, ne = 0;
  // the host reads the MASKS only (real atoms, degrees -> offsets, tile plan); the payload arrays go to the device as they are and
  // are compacted there (pack_padded_kernel) -- a training handle, whose uploads carry the reverse adjacency, packs on the host
  const bool device_pack = !h->t_master;
  scann_batch_t pb{};
  std::vector<float>& ga_packed = ps.ga;
  int r;
  if (device_pack) {
    // The payload (13/14 of the bytes) does not depend on what the masks say: it is staged and its copy ENQUEUED first, on the stream the
    // rest of the call uses, and crosses the bus while this thread reads the masks and plans the tiles -- for one batch of 128 the copy
    // (~25 us) and the mask pass + plan (~20 us) used to run one after the other in front of the first launch.
    HIPCHK(h, hipSetDevice(h->device));
    const size_t BMN = BM * (size_t)N;
    const size_t p_at = 0, p_nbr = align_up(BM * 4), p_mask = p_nbr + align_up(BMN * 4), p_w = p_mask + align_up(BMN), p_d = p_w + align_up(BMN * 4);
    const size_t p_bytes = p_d + align_up(BMN * 4);
    if (p_bytes > h->pp_cap) {
      HIPCHK(h, hipStreamSynchronize(h->streams[0]));
      if (h->pp_dev) (void)hipFree(h->pp_dev);
      if (h->pp_host) (void)hipHostFree(h->pp_host);
      h->pp_dev = h->pp_host = nullptr;
      h->pp_cap = 0;
      const size_t want = p_bytes + p_bytes / 2;
      HIPCHK(h, hipMalloc((void**)&h->pp_dev, want));
      HIPCHK(h, hipHostMalloc((void**)&h->pp_host, want, hipHostMallocDefault));
      h->pp_cap = want;
    }
    memcpy(h->pp_host + p_at, atomic, BM * 4);
    if (BMN) {
      par_memcpy(h->pp_host + p_nbr, neighbors, BMN * 4);
      par_memcpy(h->pp_host + p_mask, neighbor_mask, BMN);
      par_memcpy(h->pp_host + p_w, neighbor_weight, BMN * 4);
      par_memcpy(h->pp_host + p_d, neighbor_distance, BMN * 4);
    }
    HIPCHK(h, hipMemcpyAsync(h->pp_dev, h->pp_host, p_bytes, hipMemcpyHostToDevice, h->streams[0]));
    if (scann_count_padded(B, M, N, atom_mask, 1, neighbor_mask, 1, mol.data(), eoff.data(), gidx.data(), &na, &ne)) {
      (void)hipStreamSynchronize(h->streams[0]);  // (the staging block is about to be reusable again)
      return fail(h, SCANN_ERR_INVALID, std::string("scann_forward_padded: ") + scann_pack_last_error());
    }
    pb.n_struct = B; pb.n_atom = na; pb.n_edge = ne; pb.mol_offset = mol.data(); pb.edge_offset = eoff.data();
    PaddedSrc src{M, N, atomic, neighbors, neighbor_mask, 1, neighbor_weight, neighbor_distance, gidx.data()};
    src.d_atomic = (const int32_t*)(h->pp_dev + p_at); src.d_neighbors = (const int32_t*)(h->pp_dev + p_nbr); src.d_mask = h->pp_dev + p_mask;
    src.d_weight = (const float*)(h->pp_dev + p_w); src.d_dist = (const float*)(h->pp_dev + p_d);
    if (ga_out) grow_f(ps.ga, (size_t)na);
    scann_dbatch_t* db = nullptr;
    r = upload_impl(h, &pb, &db, true, &src);
    if (!r) r = scann_forward_resident(h, db, 0);
    if (!r) r = scann_batch_download(h, db, y_out, ga_out ? ga_packed.data() : nullptr);
  } else {
    grow_i(ps.at, BM); grow_i(ps.col, BM * N + 1); grow_f(ps.dist, BM * N + 1); grow_f(ps.wgt, BM * N + 1);
    if (scann_pack_padded(B, M, N, atomic, nullptr, atom_mask, neighbors, neighbor_mask, neighbor_weight, neighbor_distance,
                          nullptr, ps.at.data(), nullptr, nullptr, mol.data(), eoff.data(), ps.col.data(), ps.dist.data(), ps.wgt.data(),
                          gidx.data(), &na, &ne))
      return fail(h, SCANN_ERR_INVALID, std::string("scann_forward_padded: ") + scann_pack_last_error());
    pb.n_struct = B; pb.n_atom = na; pb.n_edge = ne;
    pb.atomic = ps.at.data(); pb.mol_offset = mol.data(); pb.edge_offset = eoff.data();
    pb.edge_col = ps.col.data(); pb.edge_dist = ps.dist.data(); pb.edge_weight = ps.wgt.data();
    if (ga_out) grow_f(ps.ga, (size_t)na);
    r = scann_forward(h, &pb, y_out, ga_out ? ga_packed.data() : nullptr);
  }
  if (r) return r;
  if (ga_out)
    for (size_t i = 0; i < BM; ++i) ga_out[i] = gidx[i] >= 0 ? ga_packed[gidx[i]] : 0.f;  // softmax of -1e9 -> 0
  return SCANN_OK;
}

int scann_upload_padded(scann_handle_t* h, int32_t B, int32_t M, int32_t N, const int32_t* atomic, const void* atom_mask,
                        int32_t atom_mask_size, const int32_t* neighbors, const void* neighbor_mask, int32_t neighbor_mask_size,
                        const float* neighbor_weight, const float* neighbor_distance, scann_dbatch_t** out, int32_t* n_atom_out,
                        int32_t* n_edge_out) {
  if (!h || !out || B <= 0 || M <= 0 || N < 0 || !atomic || !atom_mask || (N > 0 && (!neighbors || !neighbor_mask || !neighbor_weight || !neighbor_distance)))
    return fail(h, SCANN_ERR_INVALID, "scann_upload_padded: bad argument");
  *out = nullptr;
  const size_t BM = (size_t)B * M;
  std::vector<int32_t> row_of(BM), mol((size_t)B + 1), eoff(BM + 1);  // (per call: the caller may upload from a second thread)
  int32_t na = 0, ne = 0;
  if (scann_count_padded(B, M, N, atom_mask, atom_mask_size, neighbor_mask, neighbor_mask_size, mol.data(), eoff.data(), row_of.data(), &na, &ne))
    return fail(h, SCANN_ERR_INVALID, std::string("scann_upload_padded: ") + scann_pack_last_error());
  scann_batch_t pb{};
  pb.n_struct = B; pb.n_atom = na; pb.n_edge = ne; pb.mol_offset = mol.data(); pb.edge_offset = eoff.data();
  const PaddedSrc src{M, N, atomic, neighbors, neighbor_mask, neighbor_mask_size, neighbor_weight, neighbor_distance, row_of.data()};
  const int r = upload_impl(h, &pb, out, false, &src);
  if (r) return r;
  if (n_atom_out) *n_atom_out = na;
  if (n_edge_out) *n_edge_out = ne;
  return SCANN_OK;
}

// A batch packed on the device (scann_upload_padded) carries what pack_padded_kernel found wrong with the input in a flag word that
// scann_batch_download reads with the results.  The entry points that hand device-side tensors back WITHOUT a download (read_csr,
// forward_profile, debug_read) read the word themselves -- otherwise they would return the kernel's sanitised stand-ins (col = row,
// z = 0) as if they were the caller's data.  Call with the packing finished (upload event or stream synchronised).
static int check_pack_flag(scann_handle_t* h, scann_dbatch_t* db, const char* who) {
  if (!db->pack_flag) return SCANN_OK;
  int32_t bad = 0;
  HIPCHK(h, hipMemcpy(&bad, db->pack_flag, 4, hipMemcpyDeviceToHost));
  if (bad & 1) return fail(h, SCANN_ERR_INVALID, std::string(who) + ": an unmasked neighbour slot points at a padded atom (or outside the structure)");
  if (bad & 2) return fail(h, SCANN_ERR_INVALID, std::string(who) + ": atomic number outside the embedding table (n_atoms)");
  return SCANN_OK;
}

int scann_batch_read_csr(scann_handle_t* h, scann_dbatch_t* db, int32_t* atomic, int32_t* mol_offset, int32_t* edge_offset, int32_t* edge_col,
                         float* edge_dist, float* edge_weight) {
  if (!h || !db) return fail(h, SCANN_ERR_INVALID, "scann_batch_read_csr: null argument");
  HIPCHK(h, hipSetDevice(h->device));
  if (db->upload_ev) HIPCHK(h, hipEventSynchronize(db->upload_ev));
  else HIPCHK(h, hipStreamSynchronize(h->streams[0]));
  if (const int rp = check_pack_flag(h, db, "scann_batch_read_csr")) return rp;
  const size_t A = (size_t)db->n_atom, E = (size_t)db->n_edge;
  if (atomic) HIPCHK(h, hipMemcpy(atomic, db->atomic, A * 4, hipMemcpyDeviceToHost));
  if (mol_offset) HIPCHK(h, hipMemcpy(mol_offset, db->mol_offset, ((size_t)db->n_struct + 1) * 4, hipMemcpyDeviceToHost));
  if (edge_offset) HIPCHK(h, hipMemcpy(edge_offset, db->edge_offset, (A + 1) * 4, hipMemcpyDeviceToHost));
  if (edge_col && E) HIPCHK(h, hipMemcpy(edge_col, db->edge_col, E * 4, hipMemcpyDeviceToHost));
  if (edge_dist && E) HIPCHK(h, hipMemcpy(edge_dist, db->dist, E * 4, hipMemcpyDeviceToHost));
  if (edge_weight && E) HIPCHK(h, hipMemcpy(edge_weight, db->weight, E * 4, hipMemcpyDeviceToHost));
  return SCANN_OK;
}

int scann_forward_profile(scann_handle_t* h, scann_dbatch_t* db, scann_profile_t* prof) {
  if (!h || !db || !prof) return fail(h, SCANN_ERR_INVALID, "scann_forward_profile: null argument");
  HIPCHK(h, hipSetDevice(h->device));
  memset(prof, 0, sizeof(*prof));
  Timer tm{h->streams[0], true, {}, {}};
  db->last_slot = 0;
  const int r = run_forward(h, db, h->streams[0], &tm);
  if (r) return r;
  HIPCHK(h, hipStreamSynchronize(h->streams[0]));
  if (const int rp = check_pack_flag(h, db, "scann_forward_profile")) return rp;
  for (size_t i = 1; i < tm.ev.size(); ++i) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, tm.ev[i - 1], tm.ev[i]);
    switch (tm.kind[i]) {
      case 0: prof->ms_basis += ms; break;
      case 1: prof->ms_atom += ms; prof->n_atom_launch++; break;
      case 2: prof->ms_edge += ms; prof->n_edge_launch++; break;
      default: prof->ms_readout += ms; break;
    }
  }
  if (tm.ev.size() >= 2) (void)hipEventElapsedTime(&prof->ms_total, tm.ev.front(), tm.ev.back());
  for (hipEvent_t e : tm.ev) (void)hipEventDestroy(e);
  return SCANN_OK;
}

int scann_edge_timing(scann_handle_t* h, int every) {
  if (!h) return SCANN_ERR_INVALID;
  h->time_every = every > 0 ? every : 0;
  h->time_count = 0;
  return SCANN_OK;
}

int scann_edge_timing_read(scann_handle_t* h, double* avg_us, int64_t* n_launches, double* avg_edges) {
  if (!h || !avg_us || !n_launches) return SCANN_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  double tot = 0, edges = 0;
  int64_t n = 0;
  for (size_t i = 0; i + 1 < h->time_ev.size(); i += 2) {
    float ms = 0.f;
    if (hipEventSynchronize(h->time_ev[i + 1]) == hipSuccess && hipEventElapsedTime(&ms, h->time_ev[i], h->time_ev[i + 1]) == hipSuccess) {
      tot += ms * 1e3;
      edges += h->time_edges[i / 2];
      ++n;
    }
    (void)hipEventDestroy(h->time_ev[i]);
    (void)hipEventDestroy(h->time_ev[i + 1]);
  }
  h->time_ev.clear();
  h->time_edges.clear();
  *avg_us = n ? tot / n : 0.0;
  *n_launches = n;
  if (avg_edges) *avg_edges = n ? edges / n : 0.0;
  return SCANN_OK;
}

int scann_debug_stamps(scann_handle_t* h, scann_dbatch_t* db, uint64_t* out, int max_tiles) {
  if (!h || !db || !out) return fail(h, SCANN_ERR_INVALID, "scann_debug_stamps: null argument");
#ifdef SCANN_STAMPS
  if (!db->stamps) return fail(h, SCANN_ERR_INVALID, "scann_debug_stamps: no forward has run");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipDeviceSynchronize());
  const int n = std::min(max_tiles, db->n_stamp);
  HIPCHK(h, hipMemcpy(out, db->stamps, (size_t)n * 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return n;
#else
  (void)max_tiles;
  return fail(h, SCANN_ERR_UNSUPPORTED, "scann_debug_stamps: library was not built with -DSCANN_STAMPS");
#endif
}

int scann_debug_read(scann_handle_t* h, scann_dbatch_t* db, int what, int layer, float* out) {
  if (!h || !db || !out) return fail(h, SCANN_ERR_INVALID, "scann_debug_read: null argument");
  const int L = h->cfg.n_attention;
  if (db->dbg_layers != L) return fail(h, SCANN_ERR_INVALID, "scann_debug_read: forward was not run with debug on");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->streams[db->last_slot]));
  if (const int rp = check_pack_flag(h, db, "scann_debug_read")) return rp;
  const size_t rowA = (size_t)db->n_atom * D, rowE = (size_t)db->n_edge * D;
  const float* src = nullptr;
  size_t n = 0;
  if (what == 0 && layer >= 0 && layer <= L) { src = db->dbg_c + layer * rowA; n = rowA; }
  else if (what == 1 && h->cfg.g_update && layer >= 0 && layer <= L) { src = db->dbg_g + layer * rowE; n = rowE; }
  else if (what == 2 && layer >= 1 && layer <= L) { src = db->dbg_ctx + (layer - 1) * rowA; n = rowA; }
  else if (what >= 3 && what <= 7 && db->kept && layer >= 1 && layer <= L) {
    // per-layer tensors kept by the last TRAINING forward (scann_train_forward): 3 = K, 4 = ang, 5 = V, 6 = T [n_edge,128]; 7 = q [n_atom,128]
    const float* base = what == 3 ? db->keep_K : what == 4 ? db->keep_ang : what == 5 ? db->keep_V : what == 6 ? db->keep_T : db->keep_q;
    if (!base)  // the fused backward forms T and the gated rows again instead of reading them: the training forward does not store them
      return fail(h, SCANN_ERR_UNSUPPORTED, "scann_debug_read: this tensor is not kept by the training forward (selectors 4 = ang and 6 = T exist with the "
                                            "modular backward only: SCANN_TRAIN_FUSED=0; the base branch keeps no T)");
    n = what == 7 ? rowA : rowE;
    src = base + (size_t)(layer - 1) * n;
  }
  else return fail(h, SCANN_ERR_INVALID, "scann_debug_read: bad selector");
  if (n) HIPCHK(h, hipMemcpy(out, src, n * 4, hipMemcpyDeviceToHost));
  return SCANN_OK;
}

}  // extern "C"

// =====================================================================================================================
// Training path: training-mode forward, hand-written backward, Adam, RCCL all-reduce.
// =====================================================================================================================

struct scann_train_ws {  // per resident batch, allocated on first use
  char* arena = nullptr;
  std::vector<float*> tA;  // [n_atom,128] temporaries: 5 shared + 5 per layer and readout (operands of that layer's weight gradients)
  float *keep_q = nullptr, *keep_V = nullptr, *keep_T = nullptr, *keep_ang = nullptr, *keep_K = nullptr;  // [L][rows,128] or null
  float *keep_pre1 = nullptr, *keep_H1 = nullptr, *keep_T2 = nullptr;
  float *keep_preA = nullptr, *keep_z = nullptr;
  std::vector<float*> tE;  // [n_edge,128] temporaries: 4 shared + 2 per layer and readout
  float *rep = nullptr, *dpre = nullptr, *dy = nullptr, *targets = nullptr, *dlut = nullptr;
  float* wpart = nullptr;  // per-slab partial sums of every weight gradient of a step (WgradCtx::arena)
  size_t wpart_floats = 0;
  double* sse = nullptr;
  float drop_p = 0.f, attn_p = 0.f;
  unsigned long long seed = 0;
  GenKeep gen;  // generic widths: the training forward's tensors and the backward's temporaries
};

namespace {

std::map<scann_dbatch*, scann_train_ws> g_train_ws;  // keyed by batch; freed with the batch
std::mutex g_train_mu;                               // handles may live on different threads

int ensure_train_ws(scann_handle* h, scann_dbatch* db, scann_train_ws** out) {
  scann_train_ws* wp = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_train_mu);
    wp = &g_train_ws[db];  // std::map nodes are stable: the pointer stays valid after the lock is dropped
  }
  scann_train_ws& w = *wp;
  *out = &w;
  if (w.arena) return SCANN_OK;
  if (h->generic) {  // loss statistics, targets, d loss / d y; the tensors live in w.gen (run_forward_generic, gen_backward)
    const size_t nb = align_up((size_t)db->n_struct * 4);
    HIPCHK(h, cached_malloc((void**)&w.arena, 256 + 2 * nb));
    w.sse = (double*)w.arena;
    w.dy = (float*)(w.arena + 256);
    w.targets = (float*)(w.arena + 256 + nb);
    return SCANN_OK;
  }
  const size_t rowA = align_up((size_t)db->n_atom * D * 4), rowE = align_up((size_t)std::max(db->n_edge, 1) * D * 4);
  const size_t rowB = align_up((size_t)db->n_struct * D * 4);
  // per-layer tensors kept by the training forward (edge_kernel_lean on 64-edge tiles): q [A,128]; V, T, ang, K [E,128]
  // (base branch: geomL in the V slices, T unused)
  const size_t Lk = (size_t)h->cfg.n_attention;
  // weight-gradient partial slots: per layer <= 2 gradients over the edge rows (key, filter_geo geometry third; base: key) and
  // <= 5 over the atom rows (filter_geo centre / neighbour thirds, query, ResidualNorm dense_1 / dense_2), readout 3 over atoms
  // and 1 over structures; each with a bias row per slab
  const size_t Lc = (size_t)h->cfg.n_attention;
  w.wpart_floats = (size_t)(D * D + D) * (Lc * (2 * (size_t)wgrad_slabs(std::max(db->n_edge, 1)) + 5 * (size_t)wgrad_slabs(db->n_atom)) +
                                           3 * (size_t)wgrad_slabs(db->n_atom) + (size_t)wgrad_slabs(db->n_struct)) +
                   // LayerNorm gamma / beta partials: per layer ln_bwd over edges and atoms, attention backward over atoms
                   (size_t)2 * D * Lc * ((size_t)std::max(ln_bwd_slots(std::max(db->n_edge, 1)), tile_slots(std::max(db->n_edge, 1))) +
                                         (size_t)std::max(ln_bwd_slots(db->n_atom), tile_slots(db->n_atom)) +
                                         (size_t)attn_bwd_slots(db->n_atom, db->max_degree)) +
                   (size_t)D * db->n_struct +  // predict_property/kernel: one slot per structure (readout_bwd_kernel)
                   (size_t)4 * D * Lc * (size_t)db->n_tile;  // attention + edge backward in one launch: four vectors, one slot per tile
  // the operands of a layer's weight gradients live until the end of the step (sets of their own per layer): the gradient launches on
  // the side stream never have to be waited for before a buffer is reused
  // the modular backward (SCANN_TRAIN_FUSED=0) reads T and ang as tensors; the fused chains form them again
  const bool keep_all = !h->train_fused;
  const size_t n_keepE = keep_all ? 4 : h->cfg.g_update ? 2 : 3;
  const size_t nTA = 5 + 5 * (Lc + 1), nTE = 4 + 2 * (Lc + 1);
  const size_t total = nTA * rowA + nTE * rowE + 2 * rowB + 2 * align_up((size_t)db->n_struct * 4) +
                       align_up((size_t)h->cfg.n_atoms * D * 4) + 256 + Lk * (4 * rowA + n_keepE * rowE) + 2 * rowA + align_up(w.wpart_floats * 4);
  HIPCHK(h, cached_malloc((void**)&w.arena, total));
  char* p = w.arena;
  w.tA.assign(nTA, nullptr);
  w.tE.assign(nTE, nullptr);
  for (size_t i = 0; i < nTA; ++i) { w.tA[i] = (float*)p; p += rowA; }
  for (size_t i = 0; i < nTE; ++i) { w.tE[i] = (float*)p; p += rowE; }
  w.rep = (float*)p; p += rowB;
  w.dpre = (float*)p; p += rowB;
  w.dy = (float*)p; p += align_up((size_t)db->n_struct * 4);
  w.targets = (float*)p; p += align_up((size_t)db->n_struct * 4);
  w.dlut = (float*)p; p += align_up((size_t)h->cfg.n_atoms * D * 4);
  // (zero from here on between steps: embed_bwd_kernel clears every row it consumes -- one memset command per step less on the
  //  main stream; this one is the workspace's first and only, on the stream every training launch of the handle goes to)
  HIPCHK(h, hipMemsetAsync(w.dlut, 0, (size_t)h->cfg.n_atoms * D * 4, h->streams[0]));
  w.sse = (double*)p; p += 256;
  w.wpart = (float*)p; p += align_up(w.wpart_floats * 4);
  if (Lk) {  // slices are [rows,128] without padding between layers: size them from the un-aligned row counts
    w.keep_q = (float*)p; p += Lk * rowA;
    w.keep_V = (float*)p; p += Lk * rowE;
    if (keep_all) {
      w.keep_T = (float*)p; p += Lk * rowE;
      w.keep_ang = (float*)p; p += Lk * rowE;
    } else if (!h->cfg.g_update) {  // base branch: the gated rows feed edge_dang_kernel as they are
      w.keep_ang = (float*)p; p += Lk * rowE;
    }
    w.keep_K = (float*)p; p += Lk * rowE;
    w.keep_pre1 = (float*)p; p += Lk * rowA;
    w.keep_H1 = (float*)p; p += Lk * rowA;
    w.keep_preA = (float*)p; p += rowA;
    w.keep_z = (float*)p; p += rowA;
    w.keep_T2 = (float*)p; p += Lk * rowA;
  }
  return SCANN_OK;
}

}  // namespace

static void free_train_ws(scann_dbatch* db) {
  std::lock_guard<std::mutex> lk(g_train_mu);
  auto it = g_train_ws.find(db);
  if (it == g_train_ws.end()) return;
  cached_free(it->second.arena);
  cached_free(it->second.gen.arena);
  cached_free(it->second.gen.barena);
  g_train_ws.erase(it);
}

namespace {

int64_t spec_offset(const scann_handle* h, const std::string& name) {
  for (size_t i = 0; i < h->specs.size(); ++i)
    if (h->specs[i].name == name) return h->spec_off[i];
  return -1;
}

}  // namespace

extern "C" {

int64_t scann_param_count(const scann_handle_t* h) {
  if (!h) return SCANN_ERR_INVALID;
  int64_t n = 0;
  for (const WeightSpec& s : h->specs) n += s.numel();
  return n;
}

int scann_train_begin(scann_handle_t* h) {
  if (!h) return SCANN_ERR_INVALID;
  if (!h->loaded) return fail(h, SCANN_ERR_WEIGHTS, "scann_train_begin: weights not loaded");
  if (h->weights_exact)
    return fail(h, SCANN_ERR_UNSUPPORTED, "scann_train_begin: a 128x128 kernel has |w| >= 255.9; the training kernels multiply in split-fp16 "
                                          "form only (inference of such a checkpoint runs on the exact-fp32 kernels)");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = h->host_master.size();
  if (!h->t_master) {
    HIPCHK(h, hipMalloc((void**)&h->t_master, n * 4));
    HIPCHK(h, hipMalloc((void**)&h->t_grad, n * 4));
    HIPCHK(h, hipMalloc((void**)&h->t_m, n * 4));
    HIPCHK(h, hipMalloc((void**)&h->t_v, n * 4));
    HIPCHK(h, hipMalloc((void**)&h->t_l2, n * 4));
    if (!h->generic) HIPCHK(h, hipMalloc((void**)&h->t_descs, h->descs.size() * sizeof(RepackDesc)));
  }
  HIPCHK(h, hipMemcpy(h->t_master, h->host_master.data(), n * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemset(h->t_grad, 0, n * 4));
  HIPCHK(h, hipMemset(h->t_m, 0, n * 4));
  HIPCHK(h, hipMemset(h->t_v, 0, n * 4));
  if (!h->generic) HIPCHK(h, hipMemcpy(h->t_descs, h->descs.data(), h->descs.size() * sizeof(RepackDesc), hipMemcpyHostToDevice));
  // kernel_regularizer=l2(1e-4) mask: LocalAttention query/key/filter_geo, ResidualNorm dense_1/2, GlobalAttention
  // query/key, after_Lc, bf_property (attention.py:27-28,95-109,260-265; scann_model.py:428,441)
  std::vector<float> l2(n, 0.f);
  for (size_t i = 0; i < h->specs.size(); ++i) {
    const std::string& nm = h->specs[i].name;
    const bool is_kernel = nm.size() > 7 && nm.compare(nm.size() - 7, 7, "/kernel") == 0;
    const bool reg = is_kernel && (nm.find("local_attention_") == 0 || nm.find("residual_norm_") == 0 ||
                                   nm.find("global_attention/") == 0 || nm.find("after_Lc/") == 0 || nm.find("bf_property/") == 0);
    if (reg) std::fill(l2.begin() + h->spec_off[i], l2.begin() + h->spec_off[i] + h->specs[i].numel(), 1.0f);
  }
  HIPCHK(h, hipMemcpy(h->t_l2, l2.data(), n * 4, hipMemcpyHostToDevice));
  if (h->generic) {
    // the backward's d x = d z . W^T runs through gen_dense_kernel on transposed images of the kernels: one block per kernel, except
    // that filter_geo of the g_update branch is cut into its centre / geometry / neighbour thirds (attention.py:142-150) and
    // dense_embed with the ring input into its embedding / ring rows (scann_model.py:367-373) -- each third's d x is a tensor of its own
    h->gt_descs.clear();
    h->gt_off.clear();
    int64_t off = 0;
    h->gt_max = 0;
    const int d = h->cfg.local_dim, emb = h->cfg.embedding_dim;
    for (size_t i = 0; i < h->specs.size(); ++i) {
      const WeightSpec& sp = h->specs[i];
      const std::string& nm = sp.name;
      if (!(sp.cols > 0 && nm.size() > 7 && nm.compare(nm.size() - 7, 7, "/kernel") == 0)) continue;
      std::vector<int> cuts{0, (int)sp.rows};
      if (h->cfg.g_update && nm.find("/filter_geo/") != std::string::npos && sp.rows == 3 * d) cuts = {0, d, 2 * d, 3 * d};
      if (nm == "dense_embed/kernel" && h->cfg.use_ring) cuts = {0, emb, emb + 10};
      for (size_t b = 0; b + 1 < cuts.size(); ++b) {
        const int kn = cuts[b + 1] - cuts[b];
        h->gt_descs.push_back(GenTransDesc{h->spec_off[i], off, cuts[b], kn, (int32_t)sp.cols});
        h->gt_off[nm + "#" + std::to_string(b)] = off;
        off += (int64_t)kn * sp.cols;
        h->gt_max = std::max(h->gt_max, kn * (int)sp.cols);
      }
    }
    if (h->g_WT) (void)hipFree(h->g_WT);
    if (h->d_gt_descs) (void)hipFree(h->d_gt_descs);
    h->g_WT = nullptr;
    h->d_gt_descs = nullptr;
    HIPCHK(h, hipMalloc((void**)&h->g_WT, (size_t)std::max<int64_t>(off, 1) * 4));
    HIPCHK(h, hipMalloc((void**)&h->d_gt_descs, h->gt_descs.size() * sizeof(GenTransDesc)));
    HIPCHK(h, hipMemcpy(h->d_gt_descs, h->gt_descs.data(), h->gt_descs.size() * sizeof(GenTransDesc), hipMemcpyHostToDevice));
  } else if (!h->train_aux) {
    // (side streams created with the lowest priority changed nothing: 0.895 vs 0.895 ms per step, profiles/r04_notes.md)
    // (and so did confining them to half / a quarter of the CUs with hipExtStreamCreateWithCUMask: 0.89-0.93 ms either way)
    // A handle with a second forward stream lends it to the backward pass as its side stream instead of creating a fifth stream: HIP
    // deals a process's streams onto 4 hardware queues, and the fifth shares one (training step 0.91-0.92 -> 0.88-0.89 ms with the
    // default two forward streams; validation forwards on that stream never overlap a step).
    if (h->nstream >= 2) {
      h->train_aux = h->streams[1];
      h->train_aux_borrowed = true;
    } else {
      HIPCHK(h, hipStreamCreateWithFlags(&h->train_aux, hipStreamNonBlocking));
    }
    HIPCHK(h, hipStreamCreateWithFlags(&h->train_aux2, hipStreamNonBlocking));
    h->train_ev.resize(128);
    // fork / join events between streams of ONE device: no system-scope fence (the kernels' own end-of-kernel release already makes
    // their results visible device-wide, and nothing the host or a DMA engine wrote is ordered by them)
    const unsigned ev_flags = hipEventDisableTiming | hipEventDisableSystemFence;
    for (hipEvent_t& e : h->train_ev) HIPCHK(h, hipEventCreateWithFlags(&e, ev_flags));
  }
  h->grads_zeroed = false;  // (re)allocated gradient vector: contents unknown
  h->step_begun = h->step_ended = 0;
  {
    const char* e = getenv("SCANN_TRAIN_FUSED");
    h->train_fused = !(e && e[0] == '0');
  }
  h->t_step = 0;
  return SCANN_OK;
}

int scann_set_attention_dropout(scann_handle_t* h, float p) {
  if (!h || !(p >= 0.f && p < 1.f)) return fail(h, SCANN_ERR_INVALID, "scann_set_attention_dropout: rate must be in [0, 1)");
  h->attn_drop_p = p;
  return SCANN_OK;
}

int scann_zero_grads(scann_handle_t* h) {
  if (!h || !h->t_grad) return fail(h, SCANN_ERR_INVALID, "scann_zero_grads: call scann_train_begin first");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemsetAsync(h->t_grad, 0, h->host_master.size() * 4, h->streams[0]));
  h->grads_zeroed = true;
  return SCANN_OK;
}

int scann_get_grads(scann_handle_t* h, float* out) {
  if (!h || !h->t_grad || !out) return fail(h, SCANN_ERR_INVALID, "scann_get_grads: no training state");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->streams[0]));
  HIPCHK(h, hipMemcpy(out, h->t_grad, h->host_master.size() * 4, hipMemcpyDeviceToHost));
  return SCANN_OK;
}

int scann_get_weights(scann_handle_t* h, float* out) {
  if (!h || !out) return SCANN_ERR_INVALID;
  if (!h->t_master) {
    memcpy(out, h->host_master.data(), h->host_master.size() * 4);
    return SCANN_OK;
  }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->streams[0]));
  HIPCHK(h, hipMemcpy(out, h->t_master, h->host_master.size() * 4, hipMemcpyDeviceToHost));
  memcpy(h->host_master.data(), out, h->host_master.size() * 4);
  return SCANN_OK;
}

// the training forward (activations kept for the backward) and the batch's sum of squared errors + count -> w->sse[0..1]; no sync
static int train_forward_impl(scann_handle_t* h, scann_dbatch_t* db, const float* targets, float dropout, uint64_t seed, scann_train_ws** wout,
                              int slot) {  // slot 0 / 1: a scann_train_step in that slot; 2: the synchronous scann_train_forward
  const bool fused_step = slot < 2;
  scann_train_ws* w = nullptr;
  int r = ensure_train_ws(h, db, &w);
  if (r) return r;
  hipStream_t s = h->streams[0];
  db->last_slot = 0;
  w->drop_p = dropout;
  w->seed = seed;
  db->keep_q = w->keep_q; db->keep_V = w->keep_V; db->keep_T = w->keep_T; db->keep_ang = w->keep_ang; db->keep_K = w->keep_K;
  db->keep_pre1 = w->keep_pre1; db->keep_H1 = w->keep_H1; db->keep_T2 = w->keep_T2;
  db->keep_preA = w->keep_preA; db->keep_z = w->keep_z;
  db->kept = false;
  const bool dbg = h->debug;
  if (h->generic) {  // run_forward_generic keeps its tensors in w->gen
    w->gen.drop_p = dropout;
    w->gen.attn_p = h->attn_drop_p;
    w->gen.seed = seed;
    h->gen_keep = &w->gen;
  }
  h->debug = !h->generic;  // keep centres / geometry / context of every layer (and, with edge_kernel_lean, q / V / T / ang / K)
  h->train_drop_p = dropout;
  h->train_seed = seed;
  h->in_train_forward = true;
  w->attn_p = h->attn_drop_p;
  db->last_slot = 0;  // training runs on stream 0 (its range-guard word is slot 0's)
  r = run_forward(h, db, s, nullptr);
  h->in_train_forward = false;
  h->train_drop_p = 0.f;
  h->debug = dbg;
  h->gen_keep = nullptr;
  if (r) return r;
  if (h->generic) {
    db->kept = true;
    db->dbg_layers = h->cfg.n_attention;
  }
  // targets: staged in pinned memory that the loss kernel reads directly (it leaves the device copy the backward uses): no copy operation
  if (h->h_targets_cap[slot] < (size_t)db->n_struct) {
    if (h->h_targets[slot]) {
      HIPCHK(h, hipStreamSynchronize(s));  // an earlier step may still be reading the buffer that is about to be replaced
      (void)hipHostFree(h->h_targets[slot]);
    }
    h->h_targets[slot] = nullptr;
    h->h_targets_cap[slot] = 0;
    HIPCHK(h, hipHostMalloc((void**)&h->h_targets[slot], (size_t)db->n_struct * 4));
    h->h_targets_cap[slot] = (size_t)db->n_struct;
  }
  memcpy(h->h_targets[slot], targets, (size_t)db->n_struct * 4);
  const bool single = !(h->comm && h->comm_world > 1);
  if (fused_step && !h->h_stat) HIPCHK(h, hipHostMalloc((void**)&h->h_stat, 2 * 4 * sizeof(double)));
  // single-rank fused step: the loss kernel also forms d rmse / d y and posts {sse, count, sum |y - t|} to the slot's pinned triple
  launch_sse(db->y, h->h_targets[slot], db->n_struct, w->sse, w->targets, fused_step && single ? w->dy : nullptr,
             fused_step && single ? h->h_stat + 4 * slot : nullptr, s);
  *wout = w;
  return SCANN_OK;
}

int scann_train_forward(scann_handle_t* h, scann_dbatch_t* db, const float* targets, float dropout, uint64_t seed, double* sse_out) {
  if (!h || !db || !targets || !sse_out) return fail(h, SCANN_ERR_INVALID, "scann_train_forward: null argument");
  if (!h->t_master) return fail(h, SCANN_ERR_INVALID, "scann_train_forward: call scann_train_begin first");
  HIPCHK(h, hipSetDevice(h->device));
  scann_train_ws* w = nullptr;
  const int r = train_forward_impl(h, db, targets, dropout, seed, &w, 2);
  if (r) return r;
  hipStream_t s = h->streams[0];
  HIPCHK(h, hipMemcpyAsync(sse_out, w->sse, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  return check_range(h, "scann_train_forward");
}

static int backward_impl(scann_handle_t* h, scann_dbatch_t* db, scann_train_ws& w, float scale, const double* d_stat, bool dy_done);

int scann_train_backward(scann_handle_t* h, scann_dbatch_t* db, double sse_global, int64_t count_global) {
  if (!h || !db) return fail(h, SCANN_ERR_INVALID, "scann_train_backward: null argument");
  scann_train_ws* wp = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_train_mu);
    auto it = g_train_ws.find(db);
    if (it != g_train_ws.end()) wp = &it->second;
  }
  if (!h->t_grad || !wp || db->dbg_layers != h->cfg.n_attention || !db->kept)
    return fail(h, SCANN_ERR_INVALID, "scann_train_backward: run scann_train_forward on this batch first");
  HIPCHK(h, hipSetDevice(h->device));
  const double rmse = std::sqrt(sse_global / (double)count_global);
  const float scale = rmse > 0 ? (float)(1.0 / ((double)count_global * rmse)) : 0.f;
  h->grads_zeroed = false;
  return backward_impl(h, db, *wp, scale, nullptr, false);
}

// Reverse adjacency of a batch that was uploaded while the handle was not in training mode (scann_batch_upload skips it then):
// the neighbour indices come back from the device, the counting sort runs on the host as in upload_impl.  Synchronous; once per batch.
static int ensure_reverse(scann_handle_t* h, scann_dbatch_t* db) {
  if (db->has_rev) return SCANN_OK;
  const int A = db->n_atom, E = db->n_edge;
  hipStream_t s = h->streams[0];
  HIPCHK(h, wait_upload(db, s));
  std::vector<int32_t> col((size_t)std::max(E, 1)), in_off((size_t)A + 1, 0), in_edge((size_t)std::max(E, 1));
  if (E > 0) HIPCHK(h, hipMemcpyAsync(col.data(), db->edge_col, (size_t)E * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  for (int e = 0; e < E; ++e) ++in_off[(size_t)col[(size_t)e] + 1];
  for (int a = 0; a < A; ++a) in_off[(size_t)a + 1] += in_off[(size_t)a];
  std::vector<int32_t> fill(in_off.begin(), in_off.begin() + A);
  for (int e = 0; e < E; ++e) in_edge[(size_t)fill[(size_t)col[(size_t)e]]++] = e;
  HIPCHK(h, hipMemcpyAsync(db->in_off, in_off.data(), (size_t)(A + 1) * 4, hipMemcpyHostToDevice, s));
  if (E > 0) HIPCHK(h, hipMemcpyAsync(db->in_edge, in_edge.data(), (size_t)E * 4, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipStreamSynchronize(s));
  db->has_rev = true;
  return SCANN_OK;
}

// d_stat (device, {global sse, global count}) non-null: the loss scale is formed on the device (scann_train_step: no host round trip)
static int gen_backward(scann_handle_t* h, scann_dbatch_t* db, scann_train_ws& w, float scale, const double* d_stat, bool dy_done);

static int backward_impl(scann_handle_t* h, scann_dbatch_t* db, scann_train_ws& w, float scale, const double* d_stat, bool dy_done) {
  if (const int r = ensure_reverse(h, db)) return r;
  if (h->generic) return gen_backward(h, db, w, scale, d_stat, dy_done);
  hipStream_t s = h->streams[0];
  const scann_config_t& c = h->cfg;
  const int L = c.n_attention, A = db->n_atom, E = db->n_edge, B = db->n_struct;
  const size_t nA = (size_t)A * D, nE = (size_t)E * D;
  float* const G = h->t_grad;
  auto g = [&](const std::string& name) { return G + spec_offset(h, name); };
  if (!dy_done) launch_dy(db->y, w.targets, B, scale, d_stat, w.dy, s);
  // Weight-gradient GEMMs are off the critical path (only the final reduce needs them): with the kept-activation forward
  // their operands are never overwritten inside a layer, so they run on a side stream beside the data-gradient chain, which
  // alone does not fill the chip at batch 128.  fork(): side stream waits for everything enqueued so far; join(): main waits
  // for the side stream (start of every layer: the previous layer's temporaries are about to be overwritten).
  hipStream_t aux = h->train_aux;
  const bool side = aux != nullptr;
  size_t ev_i = 0;
  auto fork = [&]() -> hipStream_t {
    if (!side) return s;
    hipEvent_t e = h->train_ev[ev_i++ % h->train_ev.size()];
    (void)hipEventRecord(e, s);
    (void)hipStreamWaitEvent(aux, e, 0);
    return aux;
  };
  // a fork whose producer was launched with the event as its own completion signal (launch_atom_gather3 / launch_attn_edge_bwd, `done`):
  // the side stream only has to wait -- no marker packet on the main stream (4.0 instead of 6.7 us per fork, tools/fork_probe.hip)
  auto next_ev = [&]() -> hipEvent_t { return side ? h->train_ev[ev_i++ % h->train_ev.size()] : nullptr; };
  auto fork_after = [&](hipEvent_t e) -> hipStream_t {
    if (!side) return s;
    (void)hipStreamWaitEvent(aux, e, 0);
    return aux;
  };
  auto join = [&]() {
    if (!side) return;
    hipEvent_t e = h->train_ev[ev_i++ % h->train_ev.size()];
    (void)hipEventRecord(e, aux);
    (void)hipStreamWaitEvent(s, e, 0);
  };
  // The side stream's chain per layer is a gradient launch (~32 us) and two reductions of its partial slots (~34 us): as long as the
  // main stream's chain per layer (~70 us), so the step ended when the SIDE stream did, ~65 us after the main one.  The reductions
  // go to the second side stream (idle but for the basis leaf): gradient launch of layer l - 1 beside the reductions of layer l.
  bool aux2_used = false;
  auto flush_side = [&](hipStream_t ws, WgradCtx& ctx, bool last) {
    if (!side) return;
    hipStream_t fs = ws;
    // (the LAST layer's reductions stay behind their gradient launch: the second side stream is busy with the basis leaf, 46 us, by then)
    if (h->train_aux2 && !last) {
      hipEvent_t e = h->train_ev[ev_i++ % h->train_ev.size()];
      (void)hipEventRecord(e, ws);
      (void)hipStreamWaitEvent(h->train_aux2, e, 0);
      fs = h->train_aux2;
      aux2_used = true;
    }
    wgrad_flush(ctx, fs);
  };
  WgradCtx wg;
  wg.arena = w.wpart;
  // a fork costs the main stream ~7 us (tools/fork_probe.hip): the layers' gradient launches may share one (their operand sets live
  // to the end of the step)
  const int fork_every = 1;  // (2 / 3 / 4 / 7 layers per fork measured slower: profiles/r04_notes.md)

  // named temporaries
  float *dC = w.tA[0], *dCtx = w.tA[1], *t0 = w.tA[2], *t1 = w.tA[3], *t2 = w.tA[4];
  float *edAng = w.tE[0], *eT = w.tE[1], *edGa = w.tE[2], *edGb = w.tE[3];
  // The operands of a layer's weight gradients (t3, t4, dQ, dP1, dP3, edK, eU) exist once per layer (set L = the readout): the
  // gradient launch of layer l runs on the side stream beside the data-gradient chains of the layers below and nothing it reads
  // is overwritten before the end of the step.
  auto setA = [&](int l, int k) { return w.tA[5 + 5 * (size_t)l + k]; };
  auto setE = [&](int l, int k) { return w.tE[4 + 2 * (size_t)l + k]; };
  float *t3 = setA(L, 0), *t4 = setA(L, 1), *dQ = setA(L, 2), *dP1 = setA(L, 3), *dP3 = setA(L, 4), *edK = setE(L, 0), *eU = setE(L, 1);
  const float* cL = db->dbg_c + (size_t)L * nA;  // centres entering after_Lc

  // ---- readout (scann_model.py:424-447, attention.py:267-318) ----
  float* const rdgk = t3;  // the readout is "layer L" of the operand-set scheme
  // the training forward kept preA = cL.Wa + ba and z = swish(preA); gq, gk, ga, y are still in the batch workspace
  t0 = db->keep_preA;
  t1 = db->keep_z;
  ReadoutBwdArgs ra{};
  ra.mol_offset = db->mol_offset; ra.n_struct = B; ra.max_atoms = db->max_atoms; ra.use_ga_norm = c.use_ga_norm;
  ra.gq = db->gq; ra.gk = db->gk; ra.ga = db->ga; ra.dy = w.dy;
  ra.Wb = h->head.Wb; ra.bb = h->head.bb; ra.wo = h->head.wo;
  ra.dgq = t2; ra.dgk = rdgk; ra.rep_out = w.rep; ra.dpre_out = w.dpre;
  // (one slot per structure, summed in structure order with the layer's other vectors: 128 workgroups adding to the same 128
  // addresses was a queue of 16 k atomics)
  ra.dwo = reserve_vec(wg, g("predict_property/kernel"), B); ra.dbo = g("predict_property/bias");
  launch_readout_bwd(ra, s);
  // the readout's four weight gradients ride with the first layer's launch on the side stream (their operands -- rep, dpre, z = t1,
  // dgq = t2, dgk and dpreA in the readout's operand set -- are not written again before the end of the step): no fork of their own,
  // each of which costs the main stream ~7 us (tools/fork_probe.hip)
  wgrad_add(wg, w.rep, w.dpre, g("bf_property/kernel"), g("bf_property/bias"), B);
  wgrad_add(wg, t1, t2, g("global_attention/query/kernel"), g("global_attention/query/bias"), A);
  wgrad_add(wg, t1, rdgk, g("global_attention/key/kernel"), g("global_attention/key/bias"), A);
  float* const dpreA = dQ;
  launch_linear_sum(t2, h->WgqT, rdgk, h->WgkT, nullptr, nullptr, dpreA, A, 0, s, t0);  // dpreA = (dgq.Wgq^T + dgk.Wgk^T) * swish'(preA)
  wgrad_add(wg, cL, dpreA, g("after_Lc/kernel"), g("after_Lc/bias"), A);

  const float* dG_in = nullptr;  // gradient w.r.t. the geometry leaving layer l (none for the last layer)
  // fused chains (scann_train_fused.hip); SCANN_TRAIN_FUSED=0 selects the modular one-kernel-per-operation backward
  const bool fused = h->train_fused;
  struct Pend {  // projections of layer l + 1 that still have to be added to dC (d loss / d centres_{l+1})
    int n = 0;
    const float* X[3];
    const _Float16* Wh[3];
    const float* W[3];
    bool fresh = false;  // dC holds nothing yet: the terms ARE d loss / d centres (first use: dpreA.Wa^T of the readout)
  } pend;
  hipStream_t tail_s = s;  // where the embedding chain goes (below)
  auto flush_pend = [&]() {
    if (pend.n)
      launch_linear_sum(pend.X[0], pend.W[0], pend.n > 1 ? pend.X[1] : nullptr, pend.n > 1 ? pend.W[1] : nullptr,
                        pend.n > 2 ? pend.X[2] : nullptr, pend.n > 2 ? pend.W[2] : nullptr, dC, A, pend.fresh ? 0 : 1, tail_s);
    pend.n = 0;
    pend.fresh = false;
  };
  // d loss / d centres_L = dpreA.Wa^T: folded into the first rn_bwd_kernel (or launched by flush_pend)
  pend.n = 1;
  pend.X[0] = dpreA; pend.Wh[0] = h->WaTh; pend.W[0] = h->WaT;
  pend.fresh = true;
  // basis MLP (scann_model.py:378-389): a leaf (parameter gradients only) on a stream of its own, started as soon as the geometry
  // gradient entering layer 0 exists -- at 45 us it is the longest thing between there and the optimiser
  hipEvent_t ev_basis = nullptr;
  bool basis_done = false;
  auto basis_leaf = [&](const float* dG, hipEvent_t produced) {  // `produced`: completion event of the kernel that wrote dG, or null
    hipStream_t bs = s;
    if (side && h->train_aux2) {
      hipEvent_t e = produced;
      if (!e) {
        e = h->train_ev[ev_i++ % h->train_ev.size()];
        (void)hipEventRecord(e, s);
      }
      (void)hipStreamWaitEvent(h->train_aux2, e, 0);
      bs = h->train_aux2;
    }
    launch_basis_bwd(h->basis, db->dist, db->weight, dG, E, g("neighbor_d/kernel"), g("neighbor_d/bias"),
                     g("neighbor_w/kernel"), g("neighbor_w/bias"), bs);
    if (bs != s) {
      ev_basis = h->train_ev[ev_i++ % h->train_ev.size()];
      (void)hipEventRecord(ev_basis, bs);
    }
    basis_done = true;
  };
  for (int l = L - 1; l >= 0; --l) {
    t3 = setA(l, 0); t4 = setA(l, 1); dQ = setA(l, 2); dP1 = setA(l, 3); dP3 = setA(l, 4);
    edK = setE(l, 0); eU = setE(l, 1);
    const LayerParams& p = h->layers[l];
    const scann_handle::LayerT& pt = h->layersT[l];
    const std::string la = "local_attention_" + std::to_string(l) + "/", rn = "residual_norm_" + std::to_string(l) + "/";
    const float* c_in = db->dbg_c + (size_t)l * nA;        // centres entering LocalAttention l
    const float* ctx = db->dbg_ctx + (size_t)l * nA;       // LocalAttention output (after layer_norm)
    const float* Gin = c.g_update ? db->dbg_g + (size_t)l * nE : nullptr;         // geometry entering layer l
    const float* Gout = c.g_update ? db->dbg_g + (size_t)(l + 1) * nE : nullptr;  // geometry leaving layer l (= layer_norm_g output)
    // tensors the training forward kept (nothing is recomputed): q [A,128]; K, ang, V (base branch: geomL), T [E,128]
    const float* qL = db->keep_q + (size_t)l * nA;
    const float* angL = db->keep_ang ? db->keep_ang + (size_t)l * nE : nullptr;  // null: formed again from c[j] and G
    const float* KL = db->keep_K + (size_t)l * nE;
    const float* VL = db->keep_V + (size_t)l * nE;
    const float* TL = db->keep_T ? db->keep_T + (size_t)l * nE : nullptr;        // null: formed again from V and G

    if (pend.n && !(fused && c.use_attn_norm)) flush_pend();  // nobody below folds the projections of the layer above in
    // ---- ResidualNorm backward (attention.py:37-40): c_{l+1} = LN(x + drop(W2 swish(W1 x + b1) + b2)), x = ctx ----
    if (c.use_attn_norm) {
      const float* pre1 = db->keep_pre1 + (size_t)l * nA;
      const float* H1 = db->keep_H1 + (size_t)l * nA;
      const float* T2 = db->keep_T2 + (size_t)l * nA;
      if (fused) {
        // one kernel: [dC += the projections of the layer above] -> LayerNorm backward -> Dropout mask -> dense_2^T, swish' -> dense_1^T
        RnBwdArgs ra{};
        ra.dC = pend.fresh ? nullptr : dC; ra.T2 = T2; ra.pre1 = pre1; ra.gamma = p.lnr_g; ra.Wf2Th = pt.Wf2Th; ra.Wf1Th = pt.Wf1Th;
        ra.dY = t3; ra.dpre1 = t4; ra.dCtx = dCtx; ra.n_atom = A;
        ra.drop_p = w.drop_p; ra.drop_seed = w.seed; ra.drop_tag = (unsigned)l;
        ra.n_pre = pend.n;
        for (int t = 0; t < pend.n; ++t) { ra.X[t] = pend.X[t]; ra.Wh[t] = pend.Wh[t]; }
        pend.n = 0;
        pend.fresh = false;
        launch_rn_bwd(wg, ra, g(rn + "layer_norm/gamma"), g(rn + "layer_norm/beta"), s);
      } else {
        launch_ln_bwd(wg, T2, p.lnr_g, dC, dCtx, g(rn + "layer_norm/gamma"), g(rn + "layer_norm/beta"), A, 0, s);  // dT2 -> dCtx
        // gradient of the Dense_2 output = dT2 through the Dropout mask, in a buffer of its own: dT2 (dCtx) is the residual path and
        // is accumulated into below, while the queued weight gradient reads its operand at the end of the layer
        if (w.drop_p > 0.f) launch_dropout_copy(t3, dCtx, nA, w.seed, (unsigned)l, w.drop_p, s);
        else HIPCHK(h, hipMemcpyAsync(t3, dCtx, nA * 4, hipMemcpyDeviceToDevice, s));
        launch_linear(t3, pt.Wf2T, nullptr, t4, const_cast<float*>(pre1), A, 4, s);  // dpre1 = (dY.W2^T) * swish'(pre1)
        launch_linear(t4, pt.Wf1T, nullptr, dCtx, nullptr, A, 1, s);                 // dctx = dT2 + dpre1.W1^T
      }
      wgrad_add(wg, H1, t3, g(rn + "dense_2/kernel"), g(rn + "dense_2/bias"), A);
      wgrad_add(wg, ctx, t4, g(rn + "dense_1/kernel"), g(rn + "dense_1/bias"), A);
    } else {
      HIPCHK(h, hipMemcpyAsync(dCtx, dC, nA * 4, hipMemcpyDeviceToDevice, s));
    }

    // ---- LocalAttention backward (attention.py:118-216) ----
    // On the forward's 32-row tile plan (whole atoms per tile, every degree <= 16) the softmax / LayerNorm backward of a tile's atoms
    // runs at the head of the tile's edge_bwd workgroup: one launch less per layer.
    const bool fuse_attn = fused && c.g_update && db->tile_rows == 32 && db->n_big == 0 && db->max_degree <= 16 && E > 0;
    if (!fuse_attn)
      launch_attn_bwd(wg, qL, KL, db->edge_offset, dCtx, p.ln_g, dQ, edK, g(la + "layer_norm/gamma"), g(la + "layer_norm/beta"), A,
                      db->max_degree, w.attn_p, DROP_TAG_ATTN + (unsigned)l, w.seed, s);
    if (angL) wgrad_add(wg, angL, edK, g(la + "key/kernel"), g(la + "key/bias"), E);
    else wgrad_add(wg, c_in, edK, g(la + "key/kernel"), g(la + "key/bias"), E, db->edge_col, Gout);  // ang = c[j] * G'
    wgrad_add(wg, c_in, dQ, g(la + "query/kernel"), g(la + "query/bias"), A);
    if (!c.g_update) {
      // base SCANN (attention.py:155): geomL = swish(gd.Wf + bf) * weight from the raw basis (kept in the V slices), no geometry threading
      launch_linear(edK, pt.WkT, nullptr, edAng, nullptr, E, 0, s);            // dang
      launch_edge_dang(c_in, db->edge_col, VL, edAng, nullptr, eT, eU, E, s);  // eT = dang * geomL ; eU = dgeomL = dang * c[j]
      launch_gather_sum(eT, db->in_off, db->in_edge, dC, A, 0, s);             // dC[j] = sum over the edges that point at j
      // the layer's weight gradients, their reduction and the filter_geo leaf beside the chain of the layers below
      hipStream_t ws = fork();
      wgrad_launch(wg, ws);
      flush_side(ws, wg, l == 0);
      launch_base_geom_bwd(db->gd, p.Wfg, p.bfg, db->weight, eU, E, g(la + "filter_geo/kernel"), g(la + "filter_geo/bias"), ws);
      pend.n = 1;  // dC += dq.Wq^T: folded into the next rn_bwd_kernel (or launched by flush_pend)
      pend.X[0] = dQ; pend.Wh[0] = pt.WqTh; pend.W[0] = pt.WqT;
      continue;
    }
    // geometry update: G' = LN_g(swish(V) + G), V = G.W2 + P1[i] + P3[j]; gate ang = c[j] * G'
    float* dGnext = (dG_in == edGa) ? edGb : edGa;  // d loss / d geometry entering layer l
    hipEvent_t ev_sums = nullptr;
    if (fused) {
      // one kernel: dang = dK.Wk^T -> dG'tot = dang * c[j] + dG'(next layer) -> LayerNorm_g backward -> dV = dT * swish'(V) -> dG = dT + dV.W2^T
      EdgeBwdArgs ea{};
      ea.dK = edK; ea.c = c_in; ea.dG_in = dG_in; ea.T = TL; ea.G = Gin; ea.V = VL; ea.gamma = p.lng_g; ea.nb = db->edge_col;
      ea.WkTh = pt.WkTh; ea.W2Th = pt.W2Th; ea.dang = edAng; ea.dV = eU; ea.dG = dGnext; ea.n_edge = E;
      hipEvent_t ev_dg = nullptr;  // layer 0: the basis leaf waits for the geometry gradient this launch leaves
      if (fuse_attn) {
        AttnPart ab{};
        ab.q = qL; ab.K = KL; ab.dctx = dCtx; ab.gamma = p.ln_g; ab.edge_offset = db->edge_offset; ab.tiles = db->tiles;
        ab.dq = dQ; ab.dK = edK; ab.drop_p = w.attn_p; ab.drop_tag = DROP_TAG_ATTN + (unsigned)l; ab.drop_seed = w.seed;
        if (l == 0 && h->train_aux2) ev_dg = next_ev();
        launch_attn_edge_bwd(wg, ea, ab, db->n_tile, g(la + "layer_norm_g/gamma"), g(la + "layer_norm_g/beta"), g(la + "layer_norm/gamma"),
                             g(la + "layer_norm/beta"), s, ev_dg);
      } else {
        launch_edge_bwd(wg, ea, g(la + "layer_norm_g/gamma"), g(la + "layer_norm_g/beta"), s);
      }
      if (l == 0) basis_leaf(dGnext, ev_dg);  // (before the atom sums below: they do not touch the geometry gradient)
      // dC[j] = sum over the edges that point at j of dang * G' (gate), dP3[j] = the same sum of dV, dP1[i] = sum of dV over i's own edges
      ev_sums = next_ev();  // ... and this launch's completion is what the layer's weight-gradient launch on the side stream waits for
      launch_atom_gather3(edAng, Gout, eU, db->edge_offset, db->in_off, db->in_edge, dC, dP1, dP3, A, s, ev_sums);
    } else {
      launch_linear(edK, pt.WkT, nullptr, edAng, nullptr, E, 0, s);  // dang
      launch_gather_prod_sum(edAng, Gout, db->in_off, db->in_edge, dC, A, 0, s);
      // LayerNorm_g backward with its neighbours fused: in  dG'tot = dang * c[j] + dG'(next layer), out  dT (residual path, -> dGnext)
      // and dV = dT * swish'(V) (-> eU)
      launch_ln_bwd_edge(wg, TL, p.lng_g, edAng, c_in, db->edge_col, dG_in, VL, dGnext, eU, g(la + "layer_norm_g/gamma"),
                         g(la + "layer_norm_g/beta"), E, s);
      launch_atom_sums(eU, db->edge_offset, db->in_off, db->in_edge, dP1, dP3, A, s);  // dP1[i]: the atom's own edges; dP3[j]: the edges that point at j
      launch_linear(eU, pt.W2T, nullptr, dGnext, nullptr, E, 1, s);                    // dG += dV.W2^T
    }
    float* fgk = g(la + "filter_geo/kernel");
    wgrad_add(wg, Gin, eU, fgk + (size_t)D * D, nullptr, E);  // dW2
    wgrad_add(wg, c_in, dP1, fgk, g(la + "filter_geo/bias"), A);
    wgrad_add(wg, c_in, dP3, fgk + (size_t)2 * D * D, nullptr, A);
    if (l == 0 && side && ev_sums) {
      // The FIRST layer's gradient launch and its reductions are the longest thing left (the main stream only has the embedding chain,
      // ~40 us): they stay on the main stream, with no hand-over in front of them, and the embedding chain goes to the side stream
      // instead (0.830 -> 0.819 ms per step, eight alternations on one box: profiles/r05_notes.md)
      wgrad_launch(wg, s);
      wgrad_flush(wg, s);
      tail_s = fork_after(ev_sums);
    } else if ((L - 1 - l) % fork_every == fork_every - 1 || l == 0) {
      // every weight gradient of this layer (ResidualNorm 2, key, query, filter_geo 3) in ONE launch, then the fixed-order sum of its
      // partial slots (and of the layer's LayerNorm gamma / beta slots): both beside the chains of the layers below
      hipStream_t ws = ev_sums ? fork_after(ev_sums) : fork();
      wgrad_launch(wg, ws);
      flush_side(ws, wg, l == 0);
    }
    // dC += dP1.W1^T + dP3.W3^T + dq.Wq^T: folded into the next layer's rn_bwd_kernel (or launched by flush_pend)
    pend.n = 3;
    pend.X[0] = dP1; pend.X[1] = dP3; pend.X[2] = dQ;
    pend.Wh[0] = pt.W1Th; pend.Wh[1] = pt.W3Th; pend.Wh[2] = pt.WqTh;
    pend.W[0] = pt.W1T; pend.W[1] = pt.W3T; pend.W[2] = pt.WqT;
    dG_in = dGnext;
  }
  // ---- basis MLP and embedding (scann_model.py:362-389) ----
  // (the basis leaf first: it needs only the geometry gradient the last edge_bwd_kernel left, and at 43 us on its own stream it is
  // the longest thing between here and the optimiser -- started behind the embedding chain it ended 30 us after it)
  if (dG_in && !basis_done) basis_leaf(dG_in, nullptr);
  flush_pend();
  if (c.use_ring || c.feature_cgcnn) {
    launch_dropout(dC, nA, w.seed, DROP_TAG_EMBED, w.drop_p, tail_s);
    EmbedArgs e = h->embed;
    e.n_atom = A; e.atomic = db->atomic; e.c0 = db->c0;
    e.ring = c.use_ring ? db->ring : nullptr;
    e.cgcnn = c.feature_cgcnn ? db->cgcnn : nullptr;
    launch_embed_general_bwd(e, dC, c.feature_cgcnn ? nullptr : g("embed_atom/embeddings"),
                             c.feature_cgcnn ? g("embed_atom/kernel") : nullptr, c.feature_cgcnn ? g("embed_atom/bias") : nullptr,
                             c.use_ring ? g("extra_embed/kernel") : nullptr, c.use_ring ? g("extra_embed/bias") : nullptr,
                             g("dense_embed/kernel"), g("dense_embed/bias"), tail_s);
  } else {
    launch_embed_bwd(dC, db->atomic, A, h->d_weights + h->o_emb, h->d_weights + h->o_Wde, h->d_weights + h->o_bde, w.dlut,
                     c.n_atoms, c.embedding_dim, g("embed_atom/embeddings"), g("dense_embed/kernel"), g("dense_embed/bias"), w.seed,
                     DROP_TAG_EMBED, w.drop_p, tail_s);
  }
  if (wg.off > w.wpart_floats)
    return fail(h, SCANN_ERR_HIP, "scann_train_backward: weight-gradient partial arena overrun");
  if (!wg.jobs.empty()) wgrad_launch(wg, s);  // a model without LocalAttention layers: the readout's gradients were never launched
  join();
  if (aux2_used) {  // everything the second side stream was given (the basis leaf included)
    hipEvent_t e = h->train_ev[ev_i++ % h->train_ev.size()];
    (void)hipEventRecord(e, h->train_aux2);
    (void)hipStreamWaitEvent(s, e, 0);
  } else if (ev_basis) {
    (void)hipStreamWaitEvent(s, ev_basis, 0);
  }
  wgrad_flush(wg, s);  // ONE launch adds the per-slab partials of every weight gradient, in slab order
  HIPCHK(h, hipGetLastError());
  return SCANN_OK;
}

// The backward pass of create_model (scann_model.py:362-447) for a generic-width handle: the formulas of backward_impl above, one plain
// kernel each (scann_generic_train.hip), on the tensors the training forward kept (GenKeep).  One stream; gradients are ACCUMULATED
// into the flat gradient vector (two backward calls give the gradient of the sum, as on the 128-wide path).
static int gen_backward(scann_handle_t* h, scann_dbatch_t* db, scann_train_ws& w, float scale, const double* d_stat, bool dy_done) {
  hipStream_t s = h->streams[0];
  const scann_config_t& c = h->cfg;
  GenKeep& kp = w.gen;
  const int L = c.n_attention, A = db->n_atom, E = db->n_edge, B = db->n_struct;
  const int d = c.local_dim, dg = c.global_dim, dout = c.dense_out, H = c.num_head, emb = c.embedding_dim;
  const int cin = emb + (c.use_ring ? 10 : 0);
  if (!kp.arena || (int)kp.layer.size() != L) return fail(h, SCANN_ERR_INVALID, "scann_train_backward: run scann_train_forward on this batch first");
  if ((size_t)std::max(std::max(dg, dout), std::max(3 * d, std::max(cin, 92))) * 4 * 4 > 60000)
    return fail(h, SCANN_ERR_UNSUPPORTED, "backward (generic widths): a layer's rows exceed one workgroup's LDS");
  // launch limits of two kernels, checked BEFORE anything is launched (a refused launch would otherwise surface as a bare hipGetLastError at
  // the end, after earlier kernels have added into the gradient vector): gen_table_part_kernel's grid.y = 64-atom chunks,
  // gen_attn_bwd_kernel's dynamic LDS = 3 x max_degree x heads floats
  if (!c.feature_cgcnn && (A + 63) / 64 > 65535)
    return fail(h, SCANN_ERR_UNSUPPORTED, "backward (generic widths): more than 4,194,240 atoms in one batch (Embedding gradient: 65,535 chunks of 64 atoms)");
  if ((size_t)3 * std::max(1, db->max_degree) * H * sizeof(float) > 65536)
    return fail(h, SCANN_ERR_UNSUPPORTED, "backward (generic widths): an atom's neighbours x heads exceed one workgroup's LDS (3 x max_degree x num_head floats <= 64 KiB)");
  if (((size_t)3 * db->max_atoms + 4) * sizeof(double) > 65536)
    return fail(h, SCANN_ERR_UNSUPPORTED, "backward (generic widths): a structure's atoms exceed one workgroup's LDS (GlobalAttention pooling: 3 x atoms doubles <= 64 KiB)");
  // ---- temporaries ----
  const size_t fA = (size_t)A, fE = (size_t)std::max(E, 1), fB = (size_t)B;
  const size_t dmax = (size_t)std::max(d, std::max(dg, dout));
  const size_t need0 = 4 * (fA * (8 * (size_t)d + 4 * (size_t)dg + (size_t)emb + 10) + fE * 9 * (size_t)d + fB * (2 * (size_t)dout + (size_t)dg) +
                           2 * std::max(fA, fE) + 2 * 512 * dmax) + 256 * 48;
  // per-slab partial tiles of a weight gradient (gen_dense_dw_kernel): slabs x tiles <= 1024 + tiles (gen_dw_slabs), 1024 floats a tile
  const size_t Kmax = (size_t)std::max(std::max(3 * d, dg), std::max(std::max(dout, cin), 92));
  const size_t wpart = (1024 + ((Kmax + 31) / 32) * ((dmax + 31) / 32)) * 1024 + 64 * dmax;
  const size_t tpart = c.feature_cgcnn ? 0 : ((fA + 63) / 64) * (size_t)c.n_atoms * (size_t)emb;  // Embedding gradient: per-chunk sums
  const size_t need = need0 + 4 * (wpart + tpart);
  if (kp.bbytes < need) {
    HIPCHK(h, hipStreamSynchronize(s));
    cached_free(kp.barena);
    kp.barena = nullptr;
    kp.bbytes = 0;
    HIPCHK(h, cached_malloc((void**)&kp.barena, need));
    kp.bbytes = need;
  }
  float* p = reinterpret_cast<float*>(kp.barena);
  auto take = [&](size_t n) { float* q = p; p += (n + 63) & ~(size_t)63; return q; };
  float *dCa = take(fA * d), *dCb = take(fA * d), *dXr = take(fA * d), *tA1 = take(fA * d), *tA2 = take(fA * d), *dT1 = take(fA * d), *dQ = take(fA * d);
  float *dz = take(fA * dg), *dgq = take(fA * dg), *dgk = take(fA * dg), *dv = take(fA * (emb + 10));
  float *dK = take(fE * d), *dang = take(fE * d), *dGt = take(fE * d), *dT = take(fE * d), *dZ = take(fE * d), *dXi = take(fE * d), *dXj = take(fE * d);
  float *dGa = take(fE * d), *dGb = take(fE * d);
  float *dhid = take(fB * dout), *drep = take(fB * dg);
  float *stats = take(2 * std::max(fA, fE)), *part = take(2 * 512 * dmax), *wp = take(wpart), *tp_ = take(tpart);
  if (reinterpret_cast<char*>(p) > kp.barena + kp.bbytes) return fail(h, SCANN_ERR_HIP, "backward (generic widths): workspace overrun");
  // ---- helpers ----
  launch_gen_transpose(h->d_gt_descs, (int)h->gt_descs.size(), h->gt_max, h->g_weights, h->g_WT, s);
  auto Wp = [&](const std::string& name) -> const float* { return h->g_weights + h->g_off.at(name); };
  auto WT = [&](const std::string& name, int blk = 0) -> const float* { return h->g_WT + h->gt_off.at(name + "/kernel#" + std::to_string(blk)); };
  auto G = [&](const std::string& name) -> float* { return h->t_grad + h->g_off.at(name); };
  const GenSeg none{nullptr, nullptr, 0};
  // d x [rows, n_in] = d z [rows, n_out] . W^T (+ res)
  auto dx = [&](const float* dZ_, int rows, int n_out, int n_in, const float* wt, const float* res, float* out) {
    GenDenseArgs a{};
    a.seg[0] = GenSeg{dZ_, nullptr, n_out}; a.seg[1] = none; a.seg[2] = none; a.n_seg = 1;
    a.W = wt; a.b = nullptr; a.K = n_out; a.N = n_in; a.rows = rows; a.res = res; a.Y = out;
    launch_gen_dense(a, s);
  };
  bool part_overrun = false;
  auto dw = [&](GenSeg s0, GenSeg s1, GenSeg s2, int n_seg, int prod, const float* dZ_, int K_, int N_, int rows, const std::string& name) {
    GenDwArgs a{};
    a.seg[0] = s0; a.seg[1] = s1; a.seg[2] = s2; a.n_seg = n_seg; a.prod = prod;
    a.dZ = dZ_; a.K = K_; a.N = N_; a.rows = rows; a.dW = G(name + "/kernel"); a.db = G(name + "/bias"); a.part = wp;
    if (rows > 0 && gen_dw_part_floats(rows, K_, N_) > wpart) { part_overrun = true; return; }
    launch_gen_dense_dw(a, s);
  };
  auto lnb = [&](const float* X, const float* res, const std::string& name, const float* dY, int rows, float* dX) {
    launch_gen_layernorm_bwd(X, res, Wp(name + "/gamma"), dY, rows, d, dX, stats, part, G(name + "/gamma"), G(name + "/beta"), s);
  };
  if (!dy_done) launch_dy(db->y, w.targets, B, scale, d_stat, w.dy, s);
  // ---- property head (scann_model.py:437-447; mrelu's gradient is the identity, custom_layers.py:6-15) ----
  dw(GenSeg{kp.hid, nullptr, dout}, none, none, 1, 0, w.dy, dout, 1, B, "predict_property");
  dx(w.dy, B, 1, dout, WT("predict_property"), nullptr, dhid);
  launch_gen_act_bwd(dhid, kp.hid_pre, nullptr, B, dout, 0.f, 0, 0, dhid, s);
  dw(GenSeg{kp.rep, nullptr, dg}, none, none, 1, 0, dhid, dg, dout, B, "bf_property");
  dx(dhid, B, dout, dg, WT("bf_property"), nullptr, drep);
  // ---- GlobalAttention pooling, its projections, after_Lc (attention.py:279-316; scann_model.py:424-434) ----
  launch_gen_pool_bwd(db->mol_offset, B, db->max_atoms, kp.gq, kp.gk, dg, c.use_ga_norm, drep, dgq, dgk, s);
  kp.dbg = {{"gq", {kp.gq, fA * dg}}, {"gk", {kp.gk, fA * dg}}, {"z", {kp.z, fA * dg}}, {"rep", {kp.rep, fB * dg}}, {"drep", {drep, fB * dg}},
            {"dgq", {dgq, fA * dg}}, {"dgk", {dgk, fA * dg}}, {"dz", {dz, fA * dg}}};
  dw(GenSeg{kp.z, nullptr, dg}, none, none, 1, 0, dgq, dg, dg, A, "global_attention/query");
  dw(GenSeg{kp.z, nullptr, dg}, none, none, 1, 0, dgk, dg, dg, A, "global_attention/key");
  dx(dgq, A, dg, dg, WT("global_attention/query"), nullptr, dz);
  dx(dgk, A, dg, dg, WT("global_attention/key"), dz, dz);
  launch_gen_act_bwd(dz, kp.z_pre, nullptr, A, dg, 0.f, 0, 0, dz, s);
  dw(GenSeg{kp.cc_L, nullptr, d}, none, none, 1, 0, dz, d, dg, A, "after_Lc");
  float *dC = dCa, *dC_other = dCb;
  dx(dz, A, dg, d, WT("after_Lc"), nullptr, dC);
  const float* dGn = nullptr;  // gradient of the geometry leaving layer l (nothing reads the last layer's)
  float *dG_next = dGa, *dG_spare = dGb;
  // ---- LocalAttention + ResidualNorm iterations, last to first (attention.py:118-216, :37-40) ----
  for (int l = L - 1; l >= 0; --l) {
    const GenLayerKeep& b = kp.layer[(size_t)l];
    const std::string la = "local_attention_" + std::to_string(l), rn = "residual_norm_" + std::to_string(l);
    const float* dCtx = dC;
    if (c.use_attn_norm) {  // c' = LayerNorm(ctx + Dropout(dense_2(swish(dense_1 ctx))))
      lnb(b.ctx, b.t2, rn + "/layer_norm", dC, A, dXr);
      launch_gen_act_bwd(dXr, nullptr, nullptr, A, d, kp.drop_p, (unsigned)l, kp.seed, tA1, s);  // through the Dropout mask
      dw(GenSeg{b.h1, nullptr, d}, none, none, 1, 0, tA1, d, d, A, rn + "/dense_2");
      dx(tA1, A, d, d, WT(rn + "/dense_2"), nullptr, tA2);
      launch_gen_act_bwd(tA2, b.pre1, nullptr, A, d, 0.f, 0, 0, tA2, s);
      dw(GenSeg{b.ctx, nullptr, d}, none, none, 1, 0, tA2, d, d, A, rn + "/dense_1");
      dx(tA2, A, d, d, WT(rn + "/dense_1"), dXr, dXr);  // + the residual branch
      dCtx = dXr;
    }
    lnb(b.t1, nullptr, la + "/layer_norm", dCtx, A, dT1);
    launch_gen_attn_bwd(b.q, b.K, db->edge_offset, A, d, H, db->max_degree, dT1, kp.attn_p, DROP_TAG_ATTN + (unsigned)l, kp.seed, dQ, dK, s);
    dw(GenSeg{b.cc_in, nullptr, d}, none, none, 1, 0, dQ, d, d, A, la + "/query");
    dw(GenSeg{b.cc_in, db->edge_col, d}, GenSeg{b.Gn, nullptr, d}, none, 2, 1, dK, d, d, E, la + "/key");
    dx(dQ, A, d, d, WT(la + "/query"), nullptr, dC_other);
    dx(dK, E, d, d, WT(la + "/key"), nullptr, dang);  // gradient of the gated rows c[j] * g (attention.py:157)
    if (c.g_update) {
      launch_gen_mul_gather(dang, b.cc_in, db->edge_col, dGn, E, d, dGt, s);  // d g' = dang * c[j] + what the layer above left
      lnb(b.T, nullptr, la + "/layer_norm_g", dGt, E, dT);
      launch_gen_act_bwd(dT, b.Z, nullptr, E, d, 0.f, 0, 0, dZ, s);
      dw(GenSeg{b.cc_in, db->edge_row, d}, GenSeg{b.G_in, nullptr, d}, GenSeg{b.cc_in, db->edge_col, d}, 3, 0, dZ, 3 * d, d, E, la + "/filter_geo");
      dx(dZ, E, d, d, WT(la + "/filter_geo", 0), nullptr, dXi);
      dx(dZ, E, d, d, WT(la + "/filter_geo", 1), dT, dG_next);  // + the residual geometry (attention.py:152)
      dx(dZ, E, d, d, WT(la + "/filter_geo", 2), nullptr, dXj);
      launch_gen_edge_to_atom(db->edge_offset, db->in_off, db->in_edge, dXi, dXj, dang, b.Gn, dC_other, A, d, dC_other, s);
      dGn = dG_next;
      std::swap(dG_next, dG_spare);
    } else {  // g = swish(basis . Wf + bf) * Voronoi weight (attention.py:159-163)
      launch_gen_mul_gather(dang, b.cc_in, db->edge_col, nullptr, E, d, dGt, s);
      launch_gen_act_bwd(dGt, b.Z, db->weight, E, d, 0.f, 0, 0, dZ, s);
      dw(GenSeg{kp.gd, nullptr, NG}, none, none, 1, 0, dZ, NG, d, E, la + "/filter_geo");
      launch_gen_edge_to_atom(db->edge_offset, db->in_off, db->in_edge, nullptr, nullptr, dang, b.Gn, dC_other, A, d, dC_other, s);
    }
    std::swap(dC, dC_other);
  }
  // ---- basis MLP of the initial geometry (scann_model.py:386-391) ----
  if (c.g_update && dGn) {
    launch_gen_mul_gather(dGn, kp.Tw, nullptr, nullptr, E, d, dT, s);
    launch_gen_act_bwd(dT, kp.pre_d, nullptr, E, d, 0.f, 0, 0, dT, s);
    dw(GenSeg{kp.gd, nullptr, NG}, none, none, 1, 0, dT, NG, d, E, "neighbor_d");
    launch_gen_mul_gather(dGn, kp.Td, nullptr, nullptr, E, d, dZ, s);
    launch_gen_act_bwd(dZ, kp.pre_w, nullptr, E, d, 0.f, 0, 0, dZ, s);
    dw(GenSeg{kp.gw, nullptr, NG}, none, none, 1, 0, dZ, NG, d, E, "neighbor_w");
  }
  // ---- embedding (scann_model.py:362-374) ----
  launch_gen_act_bwd(dC, kp.pre_e, nullptr, A, d, kp.drop_p, DROP_TAG_EMBED, kp.seed, tA1, s);
  const GenSeg e0 = c.feature_cgcnn ? GenSeg{kp.embE, nullptr, emb} : GenSeg{Wp("embed_atom/embeddings"), db->atomic, emb};
  if (c.use_ring) dw(e0, GenSeg{kp.ring10, nullptr, 10}, none, 2, 0, tA1, cin, d, A, "dense_embed");
  else dw(e0, none, none, 1, 0, tA1, emb, d, A, "dense_embed");
  dx(tA1, A, d, emb, WT("dense_embed", 0), nullptr, dv);
  if (c.feature_cgcnn) dw(GenSeg{db->cgcnn, nullptr, 92}, none, none, 1, 0, dv, 92, emb, A, "embed_atom");
  else launch_gen_table_grad(db->atomic, A, dv, emb, c.n_atoms, tp_, G("embed_atom/embeddings"), s);
  if (c.use_ring) {
    dx(tA1, A, d, 10, WT("dense_embed", 1), nullptr, dv);
    dw(GenSeg{db->ring, nullptr, 2}, none, none, 1, 0, dv, 2, 10, A, "extra_embed");
  }
  if (part_overrun) return fail(h, SCANN_ERR_HIP, "backward (generic widths): a weight gradient's partial tiles exceed their scratch");
  HIPCHK(h, hipGetLastError());
  return SCANN_OK;
}

int64_t scann_train_debug_read(scann_handle_t* h, scann_dbatch_t* db, const char* name, float* out, int64_t cap) {
  if (!h || !db || !name || !out) return fail(h, SCANN_ERR_INVALID, "scann_train_debug_read: null argument");
  if (!h->generic) return fail(h, SCANN_ERR_UNSUPPORTED, "scann_train_debug_read: plain-fp32 (generic-width) training handles only");
  scann_train_ws* w = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_train_mu);
    auto it = g_train_ws.find(db);
    if (it != g_train_ws.end()) w = &it->second;
  }
  if (!w || w->gen.dbg.empty()) return fail(h, SCANN_ERR_INVALID, "scann_train_debug_read: run scann_train_backward on this batch first");
  auto it = w->gen.dbg.find(name);
  if (it == w->gen.dbg.end()) return fail(h, SCANN_ERR_INVALID, std::string("scann_train_debug_read: no tensor named ") + name);
  if ((int64_t)it->second.second > cap) return fail(h, SCANN_ERR_INVALID, "scann_train_debug_read: output buffer too small");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->streams[0]));
  HIPCHK(h, hipMemcpy(out, it->second.first, it->second.second * 4, hipMemcpyDeviceToHost));
  return (int64_t)it->second.second;
}

static int adam_impl(scann_handle_t* h, float lr_t, float beta1, float beta2, float eps, float l2, int zero_g) {
  hipStream_t s = h->streams[0];
  h->t_step += 1;
  const double t = (double)h->t_step;
  const float lr_hat = (float)((double)lr_t * std::sqrt(1.0 - std::pow((double)beta2, t)) / (1.0 - std::pow((double)beta1, t)));
  const size_t n = h->host_master.size();
  // t_l2 holds a 0/1 mask; fold the coefficient in by scaling through the kernel argument
  launch_adam(h->t_master, h->t_grad, h->t_m, h->t_v, h->t_l2, n, lr_hat, beta1, beta2, eps, l2, zero_g, s);
  if (h->generic) {  // the plain kernels read the Keras tensors as they are: the forward's copy is the master vector
    HIPCHK(h, hipMemcpyAsync(h->g_weights, h->t_master, n * 4, hipMemcpyDeviceToDevice, s));
    return SCANN_OK;
  }
  launch_repack(h->t_descs, (int)h->descs.size(), h->t_master, h->d_weights, h->range_flag, s);
  h->sp_dirty = true;
  if (!h->cfg.use_ring && !h->cfg.feature_cgcnn)
    launch_embed_lut(h->d_weights + h->o_emb, h->d_weights + h->o_Wde, h->d_weights + h->o_bde, h->cfg.n_atoms,
                     h->cfg.embedding_dim, h->d_weights + h->o_lut, s);
  HIPCHK(h, hipGetLastError());
  return SCANN_OK;
}

int scann_adam_step(scann_handle_t* h, float lr_t, float beta1, float beta2, float eps, float l2) {
  if (!h || !h->t_master) return fail(h, SCANN_ERR_INVALID, "scann_adam_step: call scann_train_begin first");
  HIPCHK(h, hipSetDevice(h->device));
  const int r = adam_impl(h, lr_t, beta1, beta2, eps, l2, 0);
  if (r) return r;
  HIPCHK(h, hipStreamSynchronize(h->streams[0]));
  return check_range(h, "scann_adam_step");
}

// One optimisation step without a host round trip in the middle: forward, [all-reduce of {sse, count}], backward with the loss scale
// formed on the device, [all-reduce of the gradients], Adam + weight-image refresh; ONE synchronisation at the end.  Same results as
// scann_train_forward / scann_allreduce_sse / scann_zero_grads / scann_train_backward / scann_allreduce_grads / scann_adam_step.
int scann_train_step_begin(scann_handle_t* h, scann_dbatch_t* db, const float* targets, float dropout, uint64_t seed, float lr_t, float beta1,
                           float beta2, float eps, float l2) {
  if (!h || !db || !targets) return fail(h, SCANN_ERR_INVALID, "scann_train_step: null argument");
  if (h->step_begun - h->step_ended >= 2) return fail(h, SCANN_ERR_INVALID, "scann_train_step_begin: two steps are already in flight; end one first");
  const int slot = (int)(h->step_begun & 1);
  if (!h->t_master) return fail(h, SCANN_ERR_INVALID, "scann_train_step: call scann_train_begin first");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = h->streams[0];
  scann_train_ws* w = nullptr;
  int r = train_forward_impl(h, db, targets, dropout, seed, &w, slot);
  if (r) return r;
  // From here on kernels of this step are queued: a failure below must not leave the slot's pinned buffers (targets, statistics)
  // looking free while that work is still running -- drain the stream before the error goes back (the step is not counted).
  auto drained = [&](int code) {
    (void)hipStreamSynchronize(s);
    if (h->train_aux) (void)hipStreamSynchronize(h->train_aux);
    if (h->train_aux2) (void)hipStreamSynchronize(h->train_aux2);
    h->grads_zeroed = false;
    return code;
  };
  const bool single = !(h->comm && h->comm_world > 1);
  if (!single) {  // losses.py:5-6 is the RMSE of the GLOBAL batch
    const ncclResult_t nr = ncclAllReduce(w->sse, w->sse, 3, ncclDouble, ncclSum, h->comm, s);
    if (nr != ncclSuccess) return drained(fail(h, SCANN_ERR_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(nr)));
    if (hipMemcpyAsync(h->h_stat + 4 * slot, w->sse, 3 * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess)
      return drained(fail(h, SCANN_ERR_HIP, "scann_train_step_begin: hipMemcpyAsync(statistics) failed"));
  }
  if (!h->grads_zeroed && hipMemsetAsync(h->t_grad, 0, h->host_master.size() * 4, s) != hipSuccess)
    return drained(fail(h, SCANN_ERR_HIP, "scann_train_step_begin: hipMemsetAsync(gradients) failed"));
  h->grads_zeroed = false;
  r = backward_impl(h, db, *w, 0.f, w->sse, /*dy_done=*/single);
  if (r) return drained(r);
  r = scann_allreduce_grads(h);
  if (r) return drained(r);
  r = adam_impl(h, lr_t, beta1, beta2, eps, l2, /*zero_g=*/1);  // leaves the gradient vector zeroed for the next step
  if (r) return drained(r);
  h->grads_zeroed = true;
  if (!h->step_ev[slot]) HIPCHK(h, hipEventCreateWithFlags(&h->step_ev[slot], hipEventDisableTiming));
  HIPCHK(h, hipEventRecord(h->step_ev[slot], s));
  db->busy_ev = h->step_ev[slot];
  h->step_begun += 1;
  return SCANN_OK;
}

int scann_train_step_end(scann_handle_t* h, double* sse_out, int64_t* count_out, double* abs_err_out) {
  if (!h || !sse_out || !count_out) return fail(h, SCANN_ERR_INVALID, "scann_train_step_end: null argument");
  if (h->step_begun == h->step_ended) return fail(h, SCANN_ERR_INVALID, "scann_train_step_end: no step in flight");
  HIPCHK(h, hipSetDevice(h->device));
  const int slot = (int)(h->step_ended & 1);  // the OLDEST step in flight
  HIPCHK(h, hipEventSynchronize(h->step_ev[slot]));
  h->step_ended += 1;  // only now: after a failed wait the slot still counts as in flight (its buffers are not reused)
  *sse_out = h->h_stat[4 * slot];
  *count_out = (int64_t)(h->h_stat[4 * slot + 1] + 0.5);
  if (abs_err_out) *abs_err_out = h->h_stat[4 * slot + 2];
  return check_range(h, "scann_train_step_end");
}

int scann_train_step(scann_handle_t* h, scann_dbatch_t* db, const float* targets, float dropout, uint64_t seed, float lr_t, float beta1,
                     float beta2, float eps, float l2, double* sse_out, int64_t* count_out) {
  if (!sse_out || !count_out) return fail(h, SCANN_ERR_INVALID, "scann_train_step: null argument");
  const int r = scann_train_step_begin(h, db, targets, dropout, seed, lr_t, beta1, beta2, eps, l2);
  return r ? r : scann_train_step_end(h, sse_out, count_out, nullptr);
}

int scann_comm_unique_id(char* out128) {
  if (!out128) return SCANN_ERR_INVALID;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return SCANN_ERR_HIP;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, 128);
  return SCANN_OK;
}

int scann_comm_init(scann_handle_t* h, const char* id128, int rank, int world) {
  if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return fail(h, SCANN_ERR_INVALID, "scann_comm_init: bad argument");
  HIPCHK(h, hipSetDevice(h->device));
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
  const ncclResult_t r = ncclCommInitRank(&h->comm, world, id, rank);
  if (r != ncclSuccess) return fail(h, SCANN_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  h->comm_world = world;
  return SCANN_OK;
}

int scann_comm_ranks(scann_handle_t* h) {
  if (!h) return SCANN_ERR_INVALID;
  if (!h->comm) return 0;
  int n = 0;
  const ncclResult_t r = ncclCommCount(h->comm, &n);  // what RCCL itself says the communicator spans
  if (r != ncclSuccess) return fail(h, SCANN_ERR_HIP, std::string("ncclCommCount: ") + ncclGetErrorString(r));
  return n;
}

int scann_broadcast_weights(scann_handle_t* h, int root) {
  if (!h || !h->t_master) return fail(h, SCANN_ERR_INVALID, "scann_broadcast_weights: call scann_train_begin first");
  if (!h->comm || h->comm_world == 1) return SCANN_OK;
  if (root < 0 || root >= h->comm_world) return fail(h, SCANN_ERR_INVALID, "scann_broadcast_weights: bad root");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = h->streams[0];
  const ncclResult_t r = ncclBroadcast(h->t_master, h->t_master, h->host_master.size(), ncclFloat, root, h->comm, s);
  if (r != ncclSuccess) return fail(h, SCANN_ERR_HIP, std::string("ncclBroadcast: ") + ncclGetErrorString(r));
  if (h->generic) {
    HIPCHK(h, hipMemcpyAsync(h->g_weights, h->t_master, h->host_master.size() * 4, hipMemcpyDeviceToDevice, s));
  } else {
    launch_repack(h->t_descs, (int)h->descs.size(), h->t_master, h->d_weights, h->range_flag, s);
    h->sp_dirty = true;
    if (!h->cfg.use_ring && !h->cfg.feature_cgcnn)
      launch_embed_lut(h->d_weights + h->o_emb, h->d_weights + h->o_Wde, h->d_weights + h->o_bde, h->cfg.n_atoms,
                       h->cfg.embedding_dim, h->d_weights + h->o_lut, s);
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(h->host_master.data(), h->t_master, h->host_master.size() * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  return SCANN_OK;
}

int scann_allreduce_grads(scann_handle_t* h) {
  if (!h || !h->t_grad) return fail(h, SCANN_ERR_INVALID, "scann_allreduce_grads: no training state");
  if (!h->comm || h->comm_world == 1) return SCANN_OK;
  HIPCHK(h, hipSetDevice(h->device));
  // one fused flat all-reduce (890,977 floats = 3.56 MB at the QM9 config): latency-bound on xGMI, so a single call
  const ncclResult_t r = ncclAllReduce(h->t_grad, h->t_grad, h->host_master.size(), ncclFloat, ncclSum, h->comm, h->streams[0]);
  if (r != ncclSuccess) return fail(h, SCANN_ERR_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
  return SCANN_OK;
}

int scann_allreduce_sse(scann_handle_t* h, double* sse, int64_t* count) {
  if (!h || !sse || !count) return SCANN_ERR_INVALID;
  if (!h->comm || h->comm_world == 1) return SCANN_OK;
  HIPCHK(h, hipSetDevice(h->device));
  double* d = nullptr;
  HIPCHK(h, cached_malloc((void**)&d, 2 * sizeof(double)));
  const double v[2] = {*sse, (double)*count};
  HIPCHK(h, hipMemcpy(d, v, sizeof(v), hipMemcpyHostToDevice));
  const ncclResult_t r = ncclAllReduce(d, d, 2, ncclDouble, ncclSum, h->comm, h->streams[0]);
  if (r != ncclSuccess) { cached_free(d); return fail(h, SCANN_ERR_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r)); }
  double o[2];
  HIPCHK(h, hipStreamSynchronize(h->streams[0]));
  HIPCHK(h, hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost));
  cached_free(d);
  *sse = o[0];
  *count = (int64_t)(o[1] + 0.5);
  return SCANN_OK;
}

}  // extern "C"
