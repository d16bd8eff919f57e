// Host batch packers behind the C ABI.  No GPU work in this file.
//
// The reference builds every batch from nested Python lists in DataIterator.__getitem__
// (datagenerator.py:69-135) with pad_sequence / pad_nested_sequences (general.py:14-50); at the
// rates of the HIP forward that host work is the bottleneck.  Two entry points replace it:
//   scann_pack_padded   padded Keras input dict (scann_model.py:338-357)  ->  packed CSR batch
//   scann_slice_batch   structures `sel` of a dataset held in CSR form    ->  packed CSR batch
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/scann_hip.h"

namespace {
thread_local std::string t_pack_error;
int pack_fail(const char* msg) {
  t_pack_error = msg;
  return SCANN_ERR_INVALID;
}
}  // namespace

extern "C" {

const char* scann_pack_last_error(void) { return t_pack_error.c_str(); }

int scann_pack_padded(int32_t B, int32_t M, int32_t N, const int32_t* atomic, const float* cgcnn,
                      const uint8_t* atom_mask, const int32_t* neighbors, const uint8_t* neighbor_mask,
                      const float* neighbor_weight, const float* neighbor_distance, const float* ring,
                      int32_t* out_atomic, float* out_cgcnn, float* out_ring, int32_t* out_mol_offset,
                      int32_t* out_edge_offset, int32_t* out_edge_col, float* out_edge_dist,
                      float* out_edge_weight, int32_t* out_row_of, int32_t* n_atom, int32_t* n_edge) {
  if (B < 0 || M < 0 || N < 0 || !atom_mask ||
      (N > 0 && (!neighbors || !neighbor_mask || !neighbor_weight || !neighbor_distance)) || !out_mol_offset || !out_edge_offset || !out_edge_col || !out_edge_dist ||
      !out_edge_weight || !out_row_of || !n_atom || !n_edge)
    return pack_fail("scann_pack_padded: null argument or negative shape");
  if ((atomic && !out_atomic) || (cgcnn && !out_cgcnn) || (ring && !out_ring))
    return pack_fail("scann_pack_padded: an input is given without its output array");
  // pass 1: packed row of every real atom (the job of gather_shape, custom_layers.py:18-28, moved to the host)
  int64_t na = 0;
  out_mol_offset[0] = 0;
  for (int32_t b = 0; b < B; ++b) {
    const uint8_t* am = atom_mask + (int64_t)b * M;
    int32_t* ro = out_row_of + (int64_t)b * M;
    const int64_t first = na;
    for (int32_t a = 0; a < M; ++a) ro[a] = am[a] ? (int32_t)na++ : -1;
    if (na == first) return pack_fail("a structure in the batch has no atoms");
    if (na > INT32_MAX) return pack_fail("batch too large for int32 atom rows");
    out_mol_offset[b + 1] = (int32_t)na;
  }
  // pass 2: unmasked neighbour slots of real atoms, in slot order
  int64_t ne = 0, row = 0;
  out_edge_offset[0] = 0;
  for (int32_t b = 0; b < B; ++b) {
    const int32_t* ro = out_row_of + (int64_t)b * M;
    for (int32_t a = 0; a < M; ++a) {
      if (ro[a] < 0) continue;
      const int64_t base = ((int64_t)b * M + a) * N;
      for (int32_t n = 0; n < N; ++n) {
        if (!neighbor_mask[base + n]) continue;
        const int32_t t = neighbors[base + n];
        if (t < 0 || t >= M || ro[t] < 0) return pack_fail("an unmasked neighbour slot points at a padded atom");
        if (ne >= INT32_MAX) return pack_fail("batch too large for int32 edge rows");
        out_edge_col[ne] = ro[t];
        out_edge_dist[ne] = neighbor_distance[base + n];
        out_edge_weight[ne] = neighbor_weight[base + n];
        ++ne;
      }
      if (atomic) out_atomic[row] = atomic[(int64_t)b * M + a];
      if (cgcnn) memcpy(out_cgcnn + row * 92, cgcnn + ((int64_t)b * M + a) * 92, 92 * sizeof(float));
      if (ring) memcpy(out_ring + row * 2, ring + ((int64_t)b * M + a) * 2, 2 * sizeof(float));
      out_edge_offset[++row] = (int32_t)ne;
    }
  }
  *n_atom = (int32_t)na;
  *n_edge = (int32_t)ne;
  return SCANN_OK;
}

int scann_slice_count(const int64_t* ds_mol_offset, const int64_t* ds_edge_offset, const int64_t* sel, int32_t n_sel,
                      int64_t n_struct_total, int64_t* n_atom, int64_t* n_edge) {
  if (!ds_mol_offset || !ds_edge_offset || (!sel && n_sel) || n_sel < 0 || !n_atom || !n_edge)
    return pack_fail("scann_slice_count: null argument");
  int64_t na = 0, ne = 0;
  for (int32_t i = 0; i < n_sel; ++i) {
    const int64_t s = sel[i];
    if (s < 0 || s >= n_struct_total) return pack_fail("scann_slice_count: structure index out of range");
    const int64_t a0 = ds_mol_offset[s], a1 = ds_mol_offset[s + 1];
    na += a1 - a0;
    ne += ds_edge_offset[a1] - ds_edge_offset[a0];
  }
  *n_atom = na;
  *n_edge = ne;
  return SCANN_OK;
}

int scann_slice_batch(const int64_t* ds_mol_offset, const int64_t* ds_edge_offset, const int32_t* ds_atomic,
                      const float* ds_ring, const int32_t* ds_edge_local, const float* ds_edge_dist,
                      const float* ds_edge_weight, const int64_t* sel, int32_t n_sel, int64_t n_struct_total,
                      int32_t* out_atomic, float* out_ring, int32_t* out_mol_offset, int32_t* out_edge_offset,
                      int32_t* out_edge_col, float* out_edge_dist, float* out_edge_weight) {
  if (!ds_mol_offset || !ds_edge_offset || !ds_atomic || !ds_edge_local || !ds_edge_dist || !ds_edge_weight ||
      (!sel && n_sel) || n_sel < 0 || !out_atomic || !out_mol_offset || !out_edge_offset || !out_edge_col ||
      !out_edge_dist || !out_edge_weight || (ds_ring && !out_ring))
    return pack_fail("scann_slice_batch: null argument");
  int64_t na = 0, ne = 0;
  out_mol_offset[0] = 0;
  out_edge_offset[0] = 0;
  for (int32_t i = 0; i < n_sel; ++i) {
    const int64_t s = sel[i];
    if (s < 0 || s >= n_struct_total) return pack_fail("scann_slice_batch: structure index out of range");
    const int64_t a0 = ds_mol_offset[s], a1 = ds_mol_offset[s + 1];
    const int64_t e0 = ds_edge_offset[a0], e1 = ds_edge_offset[a1];
    const int64_t cnt = a1 - a0, ecnt = e1 - e0;
    if (cnt <= 0) return pack_fail("a structure in the batch has no atoms");
    if (na + cnt > INT32_MAX || ne + ecnt > INT32_MAX) return pack_fail("batch too large for int32 rows");
    memcpy(out_atomic + na, ds_atomic + a0, cnt * sizeof(int32_t));
    if (ds_ring) memcpy(out_ring + na * 2, ds_ring + a0 * 2, cnt * 2 * sizeof(float));
    const int64_t shift = ne - e0;  // rebase the structure's CSR row pointers
    for (int64_t a = 0; a < cnt; ++a) out_edge_offset[na + a + 1] = (int32_t)(ds_edge_offset[a0 + a + 1] + shift);
    const int32_t base = (int32_t)na;  // neighbour ids inside the structure -> packed atom rows
    for (int64_t e = 0; e < ecnt; ++e) {
      const int32_t l = ds_edge_local[e0 + e];
      if (l < 0 || l >= cnt) return pack_fail("a neighbour index lies outside its structure");
      out_edge_col[ne + e] = base + l;
    }
    memcpy(out_edge_dist + ne, ds_edge_dist + e0, ecnt * sizeof(float));
    memcpy(out_edge_weight + ne, ds_edge_weight + e0, ecnt * sizeof(float));
    na += cnt;
    ne += ecnt;
    out_mol_offset[i + 1] = (int32_t)na;
  }
  return SCANN_OK;
}

}  // extern "C"
