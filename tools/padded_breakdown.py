#!/usr/bin/env python3
"""Where the host time of `model.predict(padded arrays)` goes, per chunk of 2,048 structures (the chunked pipeline's unit):
upload_padded (masks -> offsets -> tile plan -> staging copy -> H2D + pack kernel enqueued), forward_resident (launches enqueued),
download (waits for the device).  Host-side medians; the device needs ~1.05 ms per chunk at 1.95 M molecules/s."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
import bench
from scann import _hip
from scann.models.scann_model import HipModel, normalize_config

cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
rng = np.random.default_rng(1000)
batches = [bench.synth_packed_batch(rng, 128) for _ in range(16)]
pk = _hip.concat_packed(batches)
mol, eoff = np.asarray(pk.mol_offset, np.int64), np.asarray(pk.edge_offset, np.int64)
B, A, E = pk.n_struct, pk.n_atom, pk.n_edge
s_of_a = np.repeat(np.arange(B), np.diff(mol)); a_loc = np.arange(A) - mol[s_of_a]
a_of_e = np.repeat(np.arange(A), np.diff(eoff)); n_loc = np.arange(E) - eoff[a_of_e]
M, N = int(np.diff(mol).max()), int(max(1, np.diff(eoff).max()))
atomic = np.zeros((B, M), np.int32); atomic[s_of_a, a_loc] = pk.atomic
amask = np.zeros((B, M, 1), np.float32); amask[s_of_a, a_loc, 0] = 1.0
nbr, nmask = np.zeros((B, M, N), np.int32), np.zeros((B, M, N), np.float32)
dist, wgt = np.zeros((B, M, N), np.float32), np.zeros((B, M, N), np.float32)
idx = (s_of_a[a_of_e], a_loc[a_of_e], n_loc)
nbr[idx] = np.asarray(pk.edge_col, np.int64) - mol[s_of_a[a_of_e]]; nmask[idx] = 1.0
dist[idx], wgt[idx] = pk.edge_dist, pk.edge_weight
inputs = {"atomic": atomic, "atom_mask": amask, "neighbors": nbr, "neighbor_mask": nmask, "neighbor_weight": wgt, "neighbor_distance": dist}
model = HipModel(cfg, device=0, seed=1234)
eng = model.engine
t = {k: [] for k in ("count_padded (host only)", "upload_padded", "forward_resident", "download", "pack_inputs + upload (host packer)")}
for rep in range(30):
    t0 = time.perf_counter(); _hip.count_padded(inputs); t1 = time.perf_counter()
    rb = eng.upload_padded(inputs); t2 = time.perf_counter()
    eng.forward_resident(rb, 0); t3 = time.perf_counter()
    eng.download(rb, want_ga=False); t4 = time.perf_counter()
    rb.free()
    t5 = time.perf_counter(); rb = eng.upload(_hip.pack_inputs(inputs)); t6 = time.perf_counter()
    eng.forward_resident(rb, 0); eng.download(rb, want_ga=False); rb.free()
    for k, v in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t6 - t5)):
        t[k].append(v)
print("chunk: %d structures [%d, %d, %d], %d atoms, %d edges, padded payload %.1f MB" % (B, B, M, N, A, E, (nbr.nbytes * 3 + nmask.nbytes + atomic.nbytes) / 1e6))
for k, v in t.items():
    print("  %-40s median %7.3f ms" % (k, 1e3 * float(np.median(v[5:]))))
