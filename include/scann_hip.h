/*
 * scann_hip.h -- C ABI of libscann_hip.so: the MI355X (gfx950) implementation of the SCANN / SCANN+
 * forward hot path.
 *
 * The reference (sinhvt3421/scann--material) has no FFI: its hot path is the Keras graph built by
 * scann/models/scann_model.py:329-453 (create_model) and executed by `model.predict(inputs)`
 * (scann_model.py:266,316).  This header is the boundary that replaces that graph execution; the
 * Python facade (`scann--material_amd/scann`, class SCANN, `.model.predict`) binds it with ctypes.
 * Each entry point names the reference interface it stands in for.
 *
 * Conventions: every function returns 0 on success or a negative scann_status code and never
 * throws; all host buffers are caller-owned plain pointers; the library owns device memory.
 * One handle per GPU; a handle is not thread-safe, distinct handles are.
 */
#ifndef SCANN_HIP_H
#define SCANN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCANN_ABI_VERSION 1

typedef enum scann_status {
  SCANN_OK = 0,
  SCANN_ERR_INVALID = -1,     /* bad argument / malformed batch (index out of range, ...) */
  SCANN_ERR_UNSUPPORTED = -2, /* configuration outside what the kernels implement */
  SCANN_ERR_NO_DEVICE = -3,   /* no HIP device / device_id out of range */
  SCANN_ERR_HIP = -4,         /* a HIP runtime call failed (see scann_last_error) */
  SCANN_ERR_WEIGHTS = -5,     /* weights missing / wrong shape / not loaded */
  SCANN_ERR_OOM = -6,
  SCANN_ERR_RANGE = -7        /* an activation or a weight left the range of the split-fp16 projections (|x| < 65504, |w| < 255.9):
                                 the results of the call are not valid; scann_last_error names the layer and the site */
} scann_status;

/* Model hyper-parameters: the `model:` section of configs/ *.yaml as read by create_model
 * (scann_model.py:330-447).  Keys not listed there (e.g. `scale`) are never read by the reference. */
typedef struct scann_config {
  int32_t n_atoms;        /* Embedding vocabulary, scann_model.py:362 */
  int32_t embedding_dim;  /* :362 */
  int32_t local_dim;      /* :373  (128 with num_head 8, global_dim 128, dense_out 128 = every shipped yaml: the MFMA kernels; */
  int32_t num_head;       /* :399   any other widths: the plain-fp32 kernels of scann_generic*.hip, ~10 x slower) */
  int32_t n_attention;    /* :413 */
  int32_t global_dim;     /* :425 */
  int32_t dense_out;      /* :438 */
  int32_t n_gauss;        /* 20, scann_model.py:378 */
  float gaussian_d;       /* :378 */
  int32_t g_update;       /* :380  SCANN+ geometry update */
  int32_t use_attn_norm;  /* :404  ResidualNorm after each LocalAttention */
  int32_t use_ga_norm;    /* :433  GlobalAttention(norm=...) */
  int32_t use_ring;       /* :356  extra ring/aromatic input */
  int32_t feature_cgcnn;  /* :334  92-d CGCNN features instead of the Embedding */
  int32_t relu_out;       /* :446  mrelu on the output iff hyper.target == "e_b" */
} scann_config_t;

/* One named fp32 tensor inside a flat weight blob (the library's weight container; replaces the
 * Keras HDF5 checkpoint read by load_model, scann_model.py:79,87,323).  Names are listed by
 * scann_weight_name(); kernels are stored [in, out] like Keras Dense kernels. */
typedef struct scann_tensor_desc {
  const char* name;
  int64_t offset; /* element offset into the blob */
  int64_t numel;
} scann_tensor_desc_t;

/* A batch in packed (CSR) form.  The padded Keras input dict (scann_model.py:338-357; produced by
 * DataIterator.__getitem__, datagenerator.py:123-133) maps to it as: real atoms (atom_mask) of all
 * structures concatenated; per atom its unmasked neighbour slots in slot order; neighbour ids made
 * global (structure offset + neighbors[b, a, n], the job of gather_shape, custom_layers.py:18-28). */
typedef struct scann_batch {
  int32_t n_struct;           /* B */
  int32_t n_atom;             /* sum of real atoms */
  int32_t n_edge;             /* sum of unmasked neighbour slots */
  const int32_t* atomic;      /* [n_atom]  atomic number / embedding row ("atomic") */
  const int32_t* mol_offset;  /* [n_struct + 1] first atom of each structure */
  const int32_t* edge_offset; /* [n_atom + 1]  CSR row pointers */
  const int32_t* edge_col;    /* [n_edge] global atom row of the neighbour */
  const float* edge_dist;     /* [n_edge] "neighbor_distance" */
  const float* edge_weight;   /* [n_edge] "neighbor_weight" */
  const float* ring;          /* [n_atom, 2] "ring_aromatic" or NULL */
  const float* cgcnn;         /* [n_atom, 92] CGCNN features or NULL */
} scann_batch_t;

typedef struct scann_handle scann_handle_t;
typedef struct scann_dbatch scann_dbatch_t; /* a batch resident in HBM with its own workspace */

/* Per-kernel device timing of one forward (HIP events on the handle's stream). */
typedef struct scann_profile {
  float ms_basis;     /* embed + edge basis MLP            (scann_model.py:362-389) */
  float ms_atom;      /* sum over layers: atom-tile kernel (attention.py:37-40,160 + split filter_geo) */
  float ms_edge;      /* sum over layers: edge-tile kernel (attention.py:136-216) */
  float ms_readout;   /* after_Lc + GlobalAttention + head (scann_model.py:424-447) */
  float ms_total;
  int32_t n_edge_launch; /* number of edge-kernel launches timed (= n_attention) */
  int32_t n_atom_launch;
  int32_t reserved;
} scann_profile_t;

int scann_abi_version(void);
int scann_device_count(void);

/* Replaces create_model(config) (scann_model.py:329).  local_dim = global_dim = dense_out = 128 with num_head = 8 (every shipped
 * reference yaml) runs on the split-fp16 MFMA kernels; any other widths the reference accepts (scann_model.py:330-434; here: each
 * <= 1024, local_dim a multiple of num_head) evaluate AND train on the plain-fp32 kernels of csrc/scann_generic.hip /
 * csrc/scann_generic_train.hip -- same entry points, same packed batch, about ten times slower at equal width (a training step
 * about seven times).  Env SCANN_GENERIC=1 forces that path for a 128 / 8 handle (the cross-check of tests/test_gpu_parity.py and
 * tests/test_gpu_training.py). */
int scann_create(const scann_config_t* cfg, int device_id, scann_handle_t** out);
void scann_destroy(scann_handle_t* h);
const char* scann_last_error(const scann_handle_t* h); /* h may be NULL: last create error */

/* Canonical tensor list for this configuration (index 0..n-1); shape as up to 2 dims. */
int scann_weight_count(const scann_handle_t* h);
int scann_weight_name(const scann_handle_t* h, int index, const char** name, int64_t* rows, int64_t* cols);

/* Replaces load_model / model.set_weights (scann_model.py:79,323). */
int scann_load_weights(scann_handle_t* h, const float* blob, const scann_tensor_desc_t* manifest, int n);

/* Replaces model.predict(inputs) (scann_model.py:266,316) on host buffers:
 * y_out[n_struct]; ga_attn_out[n_atom] (packed GlobalAttention scores, attention.py:302) or NULL. */
int scann_forward(scann_handle_t* h, const scann_batch_t* batch, float* y_out, float* ga_attn_out);

/* The same on the PADDED Keras input dict itself (scann_model.py:338-357; DataIterator.__getitem__,
 * datagenerator.py:123-133): atomic[B,M] int32, atom_mask[B,M] bytes (bool), neighbors[B,M,N] int32,
 * neighbor_mask[B,M,N] bytes, neighbor_weight / neighbor_distance [B,M,N] float.  Packing to CSR (what gather_shape +
 * the masks express, custom_layers.py:18-28) is done natively; ga_out[B*M] (or NULL) receives the GlobalAttention scores
 * re-padded with exact zeros for padded atoms.  feature="atomic" without ring features only. */
int scann_forward_padded(scann_handle_t* h, int32_t B, int32_t M, int32_t N, const int32_t* atomic,
                         const uint8_t* atom_mask, const int32_t* neighbors, const uint8_t* neighbor_mask,
                         const float* neighbor_weight, const float* neighbor_distance, float* y_out, float* ga_out);

/* Resident-batch path (inputs already in HBM; used for pipelined inference and by bench.py).  scann_batch_upload validates the
 * CSR arrays, plans the edge tiles, copies the inputs into a pinned staging buffer of the handle and returns when the copy to the
 * device is ENQUEUED on a stream of its own: the caller's arrays may be reused at once, and whatever is launched on the batch
 * afterwards waits for the copy through the batch's event.  May be called from a second thread while the handle's owner thread
 * launches and fetches other batches (HipModel.predict_dataset, trainer.fit do). */
int scann_batch_upload(scann_handle_t* h, const scann_batch_t* batch, scann_dbatch_t** out);
/* The same, from the PADDED Keras input arrays (scann_model.py:338-357; shapes as scann_forward_padded; masks of 1-byte bool / uint8
 * or 4-byte float32 / int32 elements): the host reads the masks only (real atoms, neighbour counts -> offsets -> edge-tile plan), the
 * payload arrays are copied to the device as they are and packed to CSR there (pack_padded_kernel; replaces the host loop
 * DataIterator.__getitem__ + gather_shape amount to, datagenerator.py:69-135, custom_layers.py:18-28).  What the host packer refuses at
 * packing time -- an unmasked slot that points at a padded atom, an atomic number outside the embedding table -- is reported by
 * scann_batch_download of this batch (SCANN_ERR_INVALID).  feature="atomic" without ring features, inference handles.
 * n_atom_out / n_edge_out (or NULL) receive the packed counts.  scann_batch_read_csr copies the packed arrays of a resident batch back
 * (any pointer may be NULL): the test hook behind "device packing == scann_pack_padded, byte for byte". */
int scann_upload_padded(scann_handle_t* h, int32_t B, int32_t M, int32_t N, const int32_t* atomic, const void* atom_mask,
                        int32_t atom_mask_size, const int32_t* neighbors, const void* neighbor_mask, int32_t neighbor_mask_size,
                        const float* neighbor_weight, const float* neighbor_distance, scann_dbatch_t** out, int32_t* n_atom_out,
                        int32_t* n_edge_out);
int scann_batch_read_csr(scann_handle_t* h, scann_dbatch_t* db, int32_t* atomic, int32_t* mol_offset, int32_t* edge_offset,
                         int32_t* edge_col, float* edge_dist, float* edge_weight);
void scann_batch_free(scann_handle_t* h, scann_dbatch_t* db);
void scann_batch_release(scann_handle_t* h, scann_dbatch_t* db); /* see scann_train_step_begin */
int scann_forward_resident(scann_handle_t* h, scann_dbatch_t* db, int stream_slot); /* async */
int scann_batch_download(scann_handle_t* h, scann_dbatch_t* db, float* y_out, float* ga_attn_out); /* syncs that batch */
/* out8 = { structures, atoms, edges, atoms with more than 64 neighbours, their softmax-merge slots, largest neighbour count, edge tiles
 * and rows per tile (32 | 64) of the batch's tile plan }. */
int scann_batch_info(scann_handle_t* h, const scann_dbatch_t* db, int32_t* out8);
int scann_sync(scann_handle_t* h); /* all streams of the handle */
/* hipMemGetInfo of the handle's device: what is left of the 288 GB for resident batches (the library keeps freed blocks in a
 * per-device cache, so `free` does not rise when a batch is released; it must not FALL across repeated calls of one shape). */
int scann_device_memory(scann_handle_t* h, int64_t* free_bytes, int64_t* total_bytes);
/* Inference forwards whose range guard fired -- an activation left the range of the split-fp16 projections (|x| < 65504) -- are run
 * again by scann_batch_download / scann_forward on exact-fp32 matrix instructions (v_mfma_f32_32x32x2_f32, the arithmetic of the
 * reference's fp32 Dense layers, attention.py:95-113) instead of returning SCANN_ERR_RANGE; this counts them.  Env
 * SCANN_STRICT_RANGE=1 turns the re-run off (the error is returned).  Training entry points always return the error. */
int64_t scann_exact_reruns(const scann_handle_t* h);
int scann_num_streams(const scann_handle_t* h);

/* Timed forward of a resident batch: HIP events around every kernel on its stream. */
int scann_forward_profile(scann_handle_t* h, scann_dbatch_t* db, scann_profile_t* prof);

/* Live kernel timing inside a pipelined run: while enabled, every `every`-th forward brackets each of its edge-kernel
 * launches (not the first layer's launch of an inference forward, which has the basis MLP fused in: a different kernel) with
 * HIP events on the launch stream.  scann_edge_timing_read (after scann_sync) returns the average launch
 * duration in microseconds and the number of launches sampled, and clears the samples. */
int scann_edge_timing(scann_handle_t* h, int every);
int scann_edge_timing_read(scann_handle_t* h, double* avg_us, int64_t* n_launches, double* avg_edges);

/* Test hook: copy an intermediate of the last forward of `db` to host.
 * what: 0 = centers after layer `layer` (0 = after dense_embed) [n_atom,128];
 *       1 = geometry features after layer `layer` [n_edge,128];
 *       2 = context (LocalAttention output incl. layer_norm) of layer `layer`>=1 [n_atom,128];
 *       3, 5, 7 (after scann_train_forward only) = K, V [n_edge,128] and q [n_atom,128] kept for the backward by LocalAttention
 *       `layer`>=1;  4, 6 = ang, T [n_edge,128]: kept only when the modular backward runs (env SCANN_TRAIN_FUSED=0) -- the fused
 *       backward forms them again from (c, geometry) and (V, geometry) -- otherwise SCANN_ERR_UNSUPPORTED.
 * Only valid when the forward was run with scann_set_debug(h, 1) (keeps per-layer copies). */
int scann_set_debug(scann_handle_t* h, int on);
int scann_debug_read(scann_handle_t* h, scann_dbatch_t* db, int what, int layer, float* out);
/* Test hook, plain-fp32 (generic-width) training only: copy a tensor of the readout's backward of the last scann_train_backward on
 * `db` to host -- name one of "gq", "gk", "z" [n_atom, global_dim] (kept by the forward), "rep" [n_struct, global_dim],
 * "drep" [n_struct, global_dim], "dgq", "dgk", "dz" [n_atom, global_dim] (dz: gradient of after_Lc's pre-activation).  `cap` = floats `out` holds; returns the number of floats copied or a negative status. */
int64_t scann_train_debug_read(scann_handle_t* h, scann_dbatch_t* db, const char* name, float* out, int64_t cap);

/* Diagnostic builds (-DSCANN_STAMPS) only: per-tile phase clocks [n_tile, 16] of the last edge-kernel launch of
 * `db`; returns the number of tiles copied.  The shipped library returns SCANN_ERR_UNSUPPORTED. */
int scann_debug_stamps(scann_handle_t* h, scann_dbatch_t* db, uint64_t* out, int max_tiles);

/* ---- training step: replaces model.compile(loss=rmse, Adam(lr, decay=1e-5)) + model.fit (scann_model.py:199-241) ----
 * Gradients are hand-written derivatives of the forward graph; parameters, gradients and Adam moments are flat fp32
 * vectors in scann_weight_name() order.  Every architecture switch of create_model is covered (g_update on/off,
 * use_attn_norm, use_ga_norm, use_ring, feature="cgcnn", target "e_b"), at the shipped widths (128 / 8: MFMA kernels) and at any
 * other widths scann_create accepts (plain-fp32 kernels, every sum in a fixed order: a step is bit-reproducible on both).
 * Data-parallel use: every rank calls forward on its shard, the SSE / count are summed over ranks (host side or
 * scann_allreduce_sse), then backward, scann_allreduce_grads (one flat RCCL all-reduce), scann_adam_step. */
int64_t scann_param_count(const scann_handle_t* h);
int scann_train_begin(scann_handle_t* h);
/* training-mode forward on stream 0: keeps per-layer activations in `db`; `dropout` = rate of the two Dropout(0.1)
 * layers (scann_model.py:374, attention.py:29), 0 disables; targets[n_struct]; *sse_out = sum (y - target)^2 of this batch */
int scann_train_forward(scann_handle_t* h, scann_dbatch_t* db, const float* targets, float dropout, uint64_t seed, double* sse_out);
/* accumulates d(rmse)/d(params) into the gradient vector; rmse = sqrt(sse_global / count_global) (losses.py:5-6) */
int scann_train_backward(scann_handle_t* h, scann_dbatch_t* db, double sse_global, int64_t count_global);
/* model.use_drop (train.py --use_drop): Dropout(0.05) on the local-attention weights during training forwards
 * (attention.py:116,191); 0 disables.  Both branches (g_update on / off), MFMA and plain-fp32 kernels alike. */
int scann_set_attention_dropout(scann_handle_t* h, float p);
int scann_zero_grads(scann_handle_t* h);
int scann_allreduce_grads(scann_handle_t* h);                 /* RCCL sum over the communicator; no-op without one */
int scann_allreduce_sse(scann_handle_t* h, double* sse, int64_t* count); /* in-place sum over ranks */
/* g += 2*l2*w on the regularised kernels, then Adam (epsilon outside the sqrt, as tf.keras); lr_t already includes the
 * schedule and the legacy decay 1/(1 + 1e-5*iterations); refreshes the packed device weights */
int scann_adam_step(scann_handle_t* h, float lr_t, float beta1, float beta2, float eps, float l2);
/* One whole optimisation step, asynchronous until its end: scann_train_forward, [sum of {sse, count} over the communicator],
 * scann_zero_grads, scann_train_backward with the loss scale formed on the device, scann_allreduce_grads, scann_adam_step -- the
 * same kernels and results, without the host round trips between them (one fit step of model.fit, scann_model.py:225-241).
 * *sse_out / *count_out = the GLOBAL batch's sum of squared errors and size.  Adam leaves the gradient vector zeroed (the next step
 * needs no scann_zero_grads; scann_get_grads after a step returns zeros). */
int scann_train_step(scann_handle_t* h, scann_dbatch_t* db, const float* targets, float dropout, uint64_t seed, float lr_t, float beta1,
                     float beta2, float eps, float l2, double* sse_out, int64_t* count_out);
/* The same step in two halves: _begin enqueues everything and returns; _end waits for the OLDEST step in flight and returns its
 * {sse, count, sum |y - target|} (global over the communicator).  Up to TWO steps may be in flight: the host assembles, uploads
 * (scann_batch_upload copies on a stream of its own; the step waits for the batch's copy event) and begins step k + 1 while the
 * device still runs step k, so the device never waits for the host.  A batch must not be freed or downloaded before the _end of
 * its step; scann_batch_release then frees it without the device-wide synchronisation of scann_batch_free.  The gradient vector
 * is left zeroed. */
int scann_train_step_begin(scann_handle_t* h, scann_dbatch_t* db, const float* targets, float dropout, uint64_t seed, float lr_t, float beta1,
                           float beta2, float eps, float l2);
int scann_train_step_end(scann_handle_t* h, double* sse_out, int64_t* count_out, double* abs_err_out /* sum |y - target|, or NULL */);
int scann_get_grads(scann_handle_t* h, float* out);           /* [scann_param_count] */
int scann_get_weights(scann_handle_t* h, float* out);         /* current master parameters, same order */
int scann_comm_unique_id(char* out128);                       /* ncclGetUniqueId on rank 0; broadcast by the caller */
int scann_comm_init(scann_handle_t* h, const char* id128, int rank, int world);
/* ranks of the handle's RCCL communicator as RCCL reports them (ncclCommCount); 0 without a communicator (single rank, or the
 * collective-free inference path); negative status on error.  bench.py --train prints it as `rccl_ranks`. */
int scann_comm_ranks(scann_handle_t* h);
/* Data-parallel start-up: every rank's master parameters become rank `root`'s (one flat ncclBroadcast) and the packed
 * device images are regenerated from them, so that replicas created with different initialiser draws train ONE model
 * (the reference is single-process: create_model runs once, scann_model.py:77).  No-op without a communicator.
 * Needs scann_train_begin. */
int scann_broadcast_weights(scann_handle_t* h, int root);

/* ---- host batch packers (no GPU work) --------------------------------------------------------------
 * Replace DataIterator.__getitem__ + pad_sequence / pad_nested_sequences (datagenerator.py:69-135, general.py:14-50).
 * All arrays are caller-allocated; on error the return is SCANN_ERR_INVALID and scann_pack_last_error() has the text
 * (thread-local). */
const char* scann_pack_last_error(void);
/* Padded Keras input dict (scann_model.py:338-357) -> packed CSR.  atomic[B,M] or NULL with cgcnn[B,M,92];
 * ring[B,M,2] or NULL; masks as bytes.  Outputs sized for the worst case: out_atomic[B*M], out_cgcnn[B*M*92],
 * out_ring[B*M*2], out_mol_offset[B+1], out_edge_offset[B*M+1], out_edge_col/dist/weight[B*M*N];
 * out_row_of[B*M] = packed row of (b, m) or -1 for padded atoms (used to re-pad the GlobalAttention scores). */
int scann_pack_padded(int32_t B, int32_t M, int32_t N, const int32_t* atomic, const float* cgcnn,
                      const uint8_t* atom_mask, const int32_t* neighbors, const uint8_t* neighbor_mask,
                      const float* neighbor_weight, const float* neighbor_distance, const float* ring,
                      int32_t* out_atomic, float* out_cgcnn, float* out_ring, int32_t* out_mol_offset,
                      int32_t* out_edge_offset, int32_t* out_edge_col, float* out_edge_dist,
                      float* out_edge_weight, int32_t* out_row_of, int32_t* n_atom, int32_t* n_edge);
/* A whole dataset kept in CSR form (ds_mol_offset[n_struct_total+1] atoms per structure, ds_edge_offset[atoms+1],
 * ds_edge_local = neighbour index INSIDE its structure as stored by the preprocessing, voronoi_neighbor.py:38-47):
 * the batch made of structures sel[0..n_sel) in that order (the structures DataIterator.__getitem__(idx) would hold).
 * scann_slice_count gives the output sizes first. */
int scann_slice_count(const int64_t* ds_mol_offset, const int64_t* ds_edge_offset, const int64_t* sel, int32_t n_sel,
                      int64_t n_struct_total, int64_t* n_atom, int64_t* n_edge);
int scann_slice_batch(const int64_t* ds_mol_offset, const int64_t* ds_edge_offset, const int32_t* ds_atomic,
                      const float* ds_ring, const int32_t* ds_edge_local, const float* ds_edge_dist,
                      const float* ds_edge_weight, const int64_t* sel, int32_t n_sel, int64_t n_struct_total,
                      int32_t* out_atomic, float* out_ring, int32_t* out_mol_offset, int32_t* out_edge_offset,
                      int32_t* out_edge_col, float* out_edge_dist, float* out_edge_weight);
/* The host half of scann_upload_padded (host only; exposed so that it can be checked without a GPU): the MASKS of a padded Keras input
 * (1-byte bool / uint8 or 4-byte float32 / int32 elements) -> mol_offset[B+1], edge_offset[n_atom+1] (capacity B*M+1) and the packed row
 * of every padded atom slot, row_of[B*M] (-1: padded).  Same arrays as scann_pack_padded gives. */
int scann_count_padded(int32_t B, int32_t M, int32_t N, const void* atom_mask, int32_t atom_mask_size, const void* neighbor_mask,
                       int32_t neighbor_mask_size, int32_t* out_mol_offset, int32_t* out_edge_offset, int32_t* out_row_of,
                       int32_t* n_atom, int32_t* n_edge);
/* The staging copy of the padded path (host only): memcpy on up to 8 threads for blocks of >= 8 MiB (disjoint ranges, joined before the
 * return); what scann_upload_padded / scann_forward_padded move the payload arrays into pinned memory with.  Exposed so that it can be
 * checked without a GPU and under ThreadSanitizer (tests/test_tsan.py). */
int scann_host_copy(void* dst, const void* src, int64_t bytes);
/* The edge-tile plan scann_batch_upload builds for a packed batch (host only; exposed so that it can be checked without a
 * GPU): whole atoms per tile, <= tile_rows (32 | 64) edges and <= tile_atoms (<= 32) atoms; with allow_chunks an atom with
 * more than 64 neighbours becomes ceil(deg/64) single-atom chunk tiles, part_out[tile] = its softmax-merge slot (-1 for
 * ordinary tiles), otherwise such a batch is SCANN_ERR_UNSUPPORTED.  tiles_out[cap][4] = atom_begin, atom_end, edge_begin,
 * edge_end.  Returns the planned edge rows per tile (32 | 64) or a negative status (text: scann_pack_last_error). */
int scann_plan_tiles(const scann_batch_t* batch, int32_t tile_rows, int32_t tile_atoms, int32_t allow_chunks, int32_t cap,
                     int32_t* tiles_out, int32_t* part_out, int32_t* n_tiles, int32_t* n_slots);

#ifdef __cplusplus
}
#endif
#endif /* SCANN_HIP_H */
