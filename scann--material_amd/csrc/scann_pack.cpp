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

#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/scann_hip.h"
#include "scann_internal.h"

namespace {
thread_local std::string t_pack_error;
int pack_fail(const char* msg) {
  t_pack_error = msg;
  return SCANN_ERR_INVALID;
}
}  // namespace

namespace scann {

// Greedy tiling of atoms [a0, a1): whole atoms, <= want edges and <= tile_atoms atoms per tile; with allow_chunks an atom with more
// than `want` neighbours closes the open tile and becomes ceil(deg / want) single-atom chunk tiles (tile_part = softmax-merge slot).
static void tile_range(const int32_t* edge_offset, int32_t a0, int32_t a1, int want, int tile_atoms, bool allow_chunks,
                       std::vector<EdgeTile>& tiles, std::vector<int32_t>& tile_part, std::vector<int32_t>& big_tab, int32_t& n_slot) {
  EdgeTile cur{a0, a0, edge_offset[a0], edge_offset[a0]};
  for (int a = a0; a < a1; ++a) {
    const int32_t e0 = edge_offset[a], e1 = edge_offset[a + 1];
    if (allow_chunks && e1 - e0 > want) {  // big atom: close the open tile, then one chunk tile per <= tile_rows of its edges
      if (a > cur.atom_begin) {
        cur.atom_end = a;
        cur.edge_end = e0;
        tiles.push_back(cur);
        tile_part.push_back(-1);
      }
      big_tab.push_back(a); big_tab.push_back(n_slot); big_tab.push_back((e1 - e0 + want - 1) / want);
      for (int c0 = e0; c0 < e1; c0 += want) {
        tiles.push_back(EdgeTile{a, a + 1, c0, std::min(c0 + want, e1)});
        tile_part.push_back(n_slot++);
      }
      cur = EdgeTile{a + 1, a + 1, e1, e1};
      continue;
    }
    // greedy tiling: whole atoms, <= tile_rows edges and <= tile_atoms atoms per tile
    if ((e1 - cur.edge_begin) > want || (a - cur.atom_begin) >= tile_atoms) {
      cur.atom_end = a;
      cur.edge_end = e0;
      tiles.push_back(cur);
      tile_part.push_back(-1);
      cur = EdgeTile{a, a, e0, e0};
    }
  }
  if (cur.atom_begin < a1) {
    cur.atom_end = a1;
    cur.edge_end = edge_offset[a1];
    tiles.push_back(cur);
    tile_part.push_back(-1);
  }
}

int plan_tiles(const int32_t* mol_offset, int32_t B, const int32_t* edge_offset, const int32_t* edge_col, int32_t A, int32_t E,
               int tile_rows_req, int tile_atoms, bool allow_chunks, std::vector<EdgeTile>& tiles, std::vector<int32_t>& tile_part,
               std::vector<int32_t>& big_tab, std::vector<int32_t>& edge_row, int* tile_rows_out, int32_t* max_degree,
               int32_t* n_slot_out, std::string& err, bool fill_edge_row) {
  // Validation and the centre atom of every edge in flat passes (long loops the compiler vectorises), the greedy tiling in a loop
  // over the atoms alone: at 1.8 M molecules/s this function runs ~14,000 atoms per millisecond of device time, and the per-edge
  // branches of the first version (one fused loop) were half of scann_batch_upload's host time.
  int32_t n_slot = 0, maxdeg = 0, mindeg = 0;
  for (int a = 0; a < A; ++a) {
    const int32_t deg = edge_offset[a + 1] - edge_offset[a];
    maxdeg = std::max(maxdeg, deg);
    mindeg = std::min(mindeg, deg);
  }
  if (mindeg < 0) { err = "edge_offset not monotone"; return SCANN_ERR_INVALID; }
  if (maxdeg > TE_MAX && !allow_chunks) {
    err = "an atom has more than 64 neighbours (edge-tile limit of the selected edge kernel; edge_kernel_lean, the default on "
          "the g_update path, has none)";
    return SCANN_ERR_UNSUPPORTED;
  }
  for (int s = 0; s < B && edge_col; ++s) {  // every neighbour index inside its own structure (device packing: checked by pack_padded_kernel)
    const int32_t a0 = mol_offset[s], a1 = mol_offset[s + 1];
    int32_t mn = a0, mx = a0;
    for (int e = edge_offset[a0]; e < edge_offset[a1]; ++e) {
      mn = std::min(mn, edge_col[e]);
      mx = std::max(mx, edge_col[e]);
    }
    if (mn < a0 || mx >= a1) { err = "neighbour index outside its structure"; return SCANN_ERR_INVALID; }
  }
  if (fill_edge_row) {  // (scann_batch_upload leaves this to a kernel behind the input copy: launch_edge_row)
    edge_row.resize((size_t)E);
    for (int a = 0; a < A; ++a)
      for (int e = edge_offset[a]; e < edge_offset[a + 1]; ++e) edge_row[(size_t)e] = a;
  }
  int tile_rows = tile_rows_req;
  if (maxdeg > tile_rows && !allow_chunks) tile_rows = TE_MAX;  // a 32-row tile cannot hold the largest atom: 64-row tiles
  tiles.clear(); tile_part.clear(); big_tab.clear(); n_slot = 0;
  tile_range(edge_offset, 0, A, tile_rows, tile_atoms, allow_chunks, tiles, tile_part, big_tab, n_slot);
  if (tiles.empty()) {
    tiles.push_back(EdgeTile{0, A, 0, E});
    tile_part.push_back(-1);
  }
  *tile_rows_out = tile_rows;
  *max_degree = maxdeg;
  *n_slot_out = n_slot;
  return SCANN_OK;
}


}  // namespace scann

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
  // Structures are independent: ranges of them are counted (pass 1: real atoms, unmasked neighbour slots of real atoms), the counts
  // are turned into each range's first atom row / first edge, and the ranges are filled (pass 2) -- by one thread for a batch, by up
  // to 8 for the reference's `model.predict(whole padded dataset)` (130 k structures: 100 ms -> 15-20 ms of host time).  The output
  // does not depend on the number of threads.
  const int64_t BM = (int64_t)B * M;
  int n_thr = 1;
  if (B >= 2048) {  // (below ~1,000 structures per thread the threads' start-up costs what they save: measured at 256 per thread)
    const unsigned hw = std::thread::hardware_concurrency();
    n_thr = (int)std::min<int64_t>(std::min<unsigned>(hw ? hw : 1u, 8u), B / 1024);
    if (const char* e = getenv("SCANN_PACK_THREADS")) n_thr = std::max(1, std::min(64, atoi(e)));
    n_thr = std::max(1, std::min<int>(n_thr, B));
  }
  std::vector<int64_t> cnt_a((size_t)n_thr + 1, 0), cnt_e((size_t)n_thr + 1, 0);
  std::vector<const char*> err((size_t)n_thr, nullptr);
  auto range = [&](int t) { return std::make_pair((int32_t)((int64_t)B * t / n_thr), (int32_t)((int64_t)B * (t + 1) / n_thr)); };
  auto run = [&](auto&& fn) {
    if (n_thr == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < n_thr; ++t) th.emplace_back(fn, t);
    fn(0);
    for (std::thread& x : th) x.join();
  };
  // pass 1: counts per range
  run([&](int t) {
    const auto [b0, b1] = range(t);
    int64_t na_t = 0, ne_t = 0;
    for (int32_t b = b0; b < b1; ++b) {
      const uint8_t* am = atom_mask + (int64_t)b * M;
      int64_t here = 0;
      for (int32_t a = 0; a < M; ++a) {
        if (!am[a]) continue;
        ++here;
        const uint8_t* nm = neighbor_mask + ((int64_t)b * M + a) * N;
        for (int32_t n = 0; n < N; ++n) ne_t += nm[n] != 0;
      }
      if (here == 0) { err[t] = "a structure in the batch has no atoms"; return; }
      na_t += here;
    }
    cnt_a[t + 1] = na_t;
    cnt_e[t + 1] = ne_t;
  });
  for (int t = 0; t < n_thr; ++t) {
    if (err[t]) return pack_fail(err[t]);
    cnt_a[t + 1] += cnt_a[t];
    cnt_e[t + 1] += cnt_e[t];
  }
  const int64_t na = cnt_a[n_thr], ne = cnt_e[n_thr];
  if (na > INT32_MAX) return pack_fail("batch too large for int32 atom rows");
  if (ne > INT32_MAX) return pack_fail("batch too large for int32 edge rows");
  (void)BM;
  // pass 2: packed row of every real atom (the job of gather_shape, custom_layers.py:18-28, moved to the host), then the unmasked
  // neighbour slots of real atoms, in slot order
  out_mol_offset[0] = 0;
  out_edge_offset[0] = 0;
  run([&](int t) {
    const auto [b0, b1] = range(t);
    int64_t row = cnt_a[t], e = cnt_e[t];
    for (int32_t b = b0; b < b1; ++b) {
      const uint8_t* am = atom_mask + (int64_t)b * M;
      int32_t* ro = out_row_of + (int64_t)b * M;
      int64_t r = row;
      for (int32_t a = 0; a < M; ++a) ro[a] = am[a] ? (int32_t)r++ : -1;
      for (int32_t a = 0; a < M; ++a) {
        if (ro[a] < 0) continue;
        const int64_t base = ((int64_t)b * M + a) * N;
        for (int32_t n = 0; n < N; ++n) {
          if (!neighbor_mask[base + n]) continue;
          const int32_t tgt = neighbors[base + n];
          if (tgt < 0 || tgt >= M || ro[tgt] < 0) { err[t] = "an unmasked neighbour slot points at a padded atom"; return; }
          out_edge_col[e] = ro[tgt];
          out_edge_dist[e] = neighbor_distance[base + n];
          out_edge_weight[e] = neighbor_weight[base + n];
          ++e;
        }
        if (atomic) out_atomic[row] = atomic[(int64_t)b * M + a];
        if (cgcnn) memcpy(out_cgcnn + row * 92, cgcnn + ((int64_t)b * M + a) * 92, 92 * sizeof(float));
        if (ring) memcpy(out_ring + row * 2, ring + ((int64_t)b * M + a) * 2, 2 * sizeof(float));
        out_edge_offset[++row] = (int32_t)e;
      }
      out_mol_offset[b + 1] = (int32_t)row;
    }
  });
  for (int t = 0; t < n_thr; ++t)
    if (err[t]) return pack_fail(err[t]);
  *n_atom = (int32_t)na;
  *n_edge = (int32_t)ne;
  return SCANN_OK;
}

// The host half of the DEVICE packing (scann_upload_padded): only the 1-byte (or 4-byte) MASKS are read here -- which atoms are real, how
// many unmasked neighbour slots each has -- and turned into mol_offset / edge_offset / the packed row of every padded atom slot.  The
// 4-byte payload arrays (neighbour indices, distances, weights, atomic numbers: 13/14 of the bytes) are never touched by the host: they
// are copied to the device as they are and compacted there (pack_padded_kernel).  Same semantics as scann_pack_padded (datagenerator.py:
// 69-135, custom_layers.py:18-28): a real atom = atom_mask != 0; an edge = an unmasked slot of a real atom, in slot order.
// Masks of 1 byte (bool / uint8) or 4 bytes (the float32 masks of the Keras input dict, int32; -0.0f counts as 0).
namespace {
inline bool mask_at(const void* m, int size, int64_t i) {
  return size == 1 ? static_cast<const uint8_t*>(m)[i] != 0 : (static_cast<const uint32_t*>(m)[i] & 0x7fffffffu) != 0;
}
}  // namespace

// memcpy of a large block on up to 8 threads (the padded payload of a 2,048-structure chunk is ~10 MB, of a whole dataset ~100: one
// thread moves ~7-10 GB/s, which would make the staging copy the longest host step of the padded path).  Disjoint 64-byte-aligned
// ranges, joined before the return.
int scann_host_copy(void* dst, const void* src, int64_t bytes_) {
  if (bytes_ < 0 || ((!dst || !src) && bytes_ > 0)) return pack_fail("scann_host_copy: null argument or negative size");
  const size_t bytes = (size_t)bytes_;
  const unsigned hw = std::thread::hardware_concurrency();
  size_t n_thr = std::min<size_t>(std::min<unsigned>(hw ? hw : 1u, 8u), bytes >> 22);  // >= 4 MiB per thread
  if (const char* e = getenv("SCANN_COPY_THREADS")) n_thr = (size_t)std::max(1, std::min(64, atoi(e)));  // (tests: threads on a small block)
  if (n_thr <= 1 || bytes < 64 * n_thr) { memcpy(dst, src, bytes); return SCANN_OK; }
  std::vector<std::thread> th;
  for (size_t t = 1; t < n_thr; ++t) {
    const size_t o0 = (bytes * t / n_thr) & ~(size_t)63, o1 = t + 1 == n_thr ? bytes : (bytes * (t + 1) / n_thr) & ~(size_t)63;
    th.emplace_back([=] { memcpy(static_cast<char*>(dst) + o0, static_cast<const char*>(src) + o0, o1 - o0); });
  }
  memcpy(dst, src, (bytes / n_thr) & ~(size_t)63);
  for (std::thread& x : th) x.join();
  return SCANN_OK;
}

int scann_count_padded(int32_t B, int32_t M, int32_t N, const void* atom_mask, int32_t atom_mask_size, const void* neighbor_mask,
                       int32_t neighbor_mask_size, int32_t* out_mol_offset, int32_t* out_edge_offset, int32_t* out_row_of, int32_t* n_atom,
                       int32_t* n_edge) {
  if (B < 0 || M < 0 || N < 0 || !atom_mask || (N > 0 && !neighbor_mask) || !out_mol_offset || !out_edge_offset || !out_row_of || !n_atom || !n_edge)
    return pack_fail("scann_count_padded: null argument or negative shape");
  if ((atom_mask_size != 1 && atom_mask_size != 4) || (neighbor_mask_size != 1 && neighbor_mask_size != 4))
    return pack_fail("scann_count_padded: masks must have 1-byte or 4-byte elements");
  int n_thr = 1;
  if (B >= 2048) {
    const unsigned hw = std::thread::hardware_concurrency();
    n_thr = (int)std::min<int64_t>(std::min<unsigned>(hw ? hw : 1u, 8u), B / 1024);
    if (const char* e = getenv("SCANN_PACK_THREADS")) n_thr = std::max(1, std::min(64, atoi(e)));
    n_thr = std::max(1, std::min<int>(n_thr, B));
  }
  std::vector<int64_t> cnt_a((size_t)n_thr + 1, 0), cnt_e((size_t)n_thr + 1, 0);
  std::vector<const char*> err((size_t)n_thr, nullptr);
  auto range = [&](int t) { return std::make_pair((int32_t)((int64_t)B * t / n_thr), (int32_t)((int64_t)B * (t + 1) / n_thr)); };
  auto run = [&](auto&& fn) {
    if (n_thr == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < n_thr; ++t) th.emplace_back(fn, t);
    fn(0);
    for (std::thread& x : th) x.join();
  };
  // degree of every padded atom slot (0 for a padded atom), kept in out_row_of until the second pass turns it into the row
  auto degree = [&](int64_t bm) {
    int32_t d = 0;
    if (neighbor_mask_size == 1) {
      const uint8_t* nm = static_cast<const uint8_t*>(neighbor_mask) + bm * N;
      for (int32_t n = 0; n < N; ++n) d += nm[n] != 0;
    } else {
      const uint32_t* nm = static_cast<const uint32_t*>(neighbor_mask) + bm * N;
      for (int32_t n = 0; n < N; ++n) d += (nm[n] & 0x7fffffffu) != 0;
    }
    return d;
  };
  run([&](int t) {
    const auto [b0, b1] = range(t);
    int64_t na_t = 0, ne_t = 0;
    for (int32_t b = b0; b < b1; ++b) {
      int64_t here = 0;
      for (int32_t a = 0; a < M; ++a) {
        const int64_t bm = (int64_t)b * M + a;
        if (!mask_at(atom_mask, atom_mask_size, bm)) { out_row_of[bm] = -1; continue; }
        ++here;
        const int32_t d = degree(bm);
        out_row_of[bm] = d;
        ne_t += d;
      }
      if (here == 0) { err[t] = "a structure in the batch has no atoms"; return; }
      na_t += here;
    }
    cnt_a[t + 1] = na_t;
    cnt_e[t + 1] = ne_t;
  });
  for (int t = 0; t < n_thr; ++t) {
    if (err[t]) return pack_fail(err[t]);
    cnt_a[t + 1] += cnt_a[t];
    cnt_e[t + 1] += cnt_e[t];
  }
  if (cnt_a[n_thr] > INT32_MAX) return pack_fail("batch too large for int32 atom rows");
  if (cnt_e[n_thr] > INT32_MAX) return pack_fail("batch too large for int32 edge rows");
  out_mol_offset[0] = 0;
  out_edge_offset[0] = 0;
  run([&](int t) {
    const auto [b0, b1] = range(t);
    int64_t row = cnt_a[t], e = cnt_e[t];
    for (int32_t b = b0; b < b1; ++b) {
      for (int32_t a = 0; a < M; ++a) {
        const int64_t bm = (int64_t)b * M + a;
        const int32_t d = out_row_of[bm];
        if (d < 0) continue;
        out_row_of[bm] = (int32_t)row;
        e += d;
        out_edge_offset[++row] = (int32_t)e;
      }
      out_mol_offset[b + 1] = (int32_t)row;
    }
  });
  *n_atom = (int32_t)cnt_a[n_thr];
  *n_edge = (int32_t)cnt_e[n_thr];
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

int scann_plan_tiles(const scann_batch_t* b, int32_t tile_rows, int32_t tile_atoms, int32_t allow_chunks, int32_t cap,
                     int32_t* tiles_out, int32_t* part_out, int32_t* n_tiles, int32_t* n_slots) {
  if (!b || !n_tiles || !n_slots || !b->mol_offset || !b->edge_offset || (b->n_edge > 0 && !b->edge_col) || b->n_struct <= 0 ||
      b->n_atom <= 0 || b->n_edge < 0 || (tile_rows != 32 && tile_rows != 64) || tile_atoms <= 0 || tile_atoms > 32)
    return pack_fail("scann_plan_tiles: bad argument");
  if (b->mol_offset[0] != 0 || b->mol_offset[b->n_struct] != b->n_atom || b->edge_offset[0] != 0 || b->edge_offset[b->n_atom] != b->n_edge)
    return pack_fail("scann_plan_tiles: offsets do not cover the batch");
  for (int s = 0; s < b->n_struct; ++s)
    if (b->mol_offset[s + 1] <= b->mol_offset[s]) return pack_fail("scann_plan_tiles: structure without atoms");
  std::vector<scann::EdgeTile> tiles;
  std::vector<int32_t> part, big, row;
  int rows = 0;
  int32_t maxdeg = 0, nslot = 0;
  std::string err;
  const int r = scann::plan_tiles(b->mol_offset, b->n_struct, b->edge_offset, b->edge_col, b->n_atom, b->n_edge, tile_rows, tile_atoms,
                                  allow_chunks != 0, tiles, part, big, row, &rows, &maxdeg, &nslot, err);
  if (r) {
    t_pack_error = "scann_plan_tiles: " + err;
    return r;
  }
  *n_tiles = (int32_t)tiles.size();
  *n_slots = nslot;
  if ((int32_t)tiles.size() > cap || !tiles_out || !part_out) return tiles_out ? pack_fail("scann_plan_tiles: output capacity too small") : SCANN_OK;
  for (size_t i = 0; i < tiles.size(); ++i) {
    tiles_out[4 * i] = tiles[i].atom_begin; tiles_out[4 * i + 1] = tiles[i].atom_end;
    tiles_out[4 * i + 2] = tiles[i].edge_begin; tiles_out[4 * i + 3] = tiles[i].edge_end;
    part_out[i] = part[i];
  }
  return rows;  // 32 or 64: the edge rows per tile actually planned
}

}  // extern "C"

