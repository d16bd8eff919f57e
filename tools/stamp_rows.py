#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of edge_rows_kernel from in-kernel s_memtime sums (one row per wave).
Run with SCANN_HIP_LIB=scann--material_amd/lib/libscann_hip_stamps.so (make -C scann--material_amd/csrc stamps)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd")); sys.path.insert(0, ROOT)
from scann.models.scann_model import HipModel, normalize_config
import bench

cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234)
eng = model.engine
rng = np.random.default_rng(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rb = eng.upload(bench.synth_packed_batch(rng, nb))
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    eng.forward_resident(rb, 0)
eng.sync()
st = eng.debug_stamps(rb, 4096).astype(np.int64)
names = ["queue / index wait", "row loads + accumulator init", "GEMM 1", "ticket + next indices", "swish, LN_g, geom' store, ang",
         "GEMM 2 (+ q requests)", "logits", "softmax scans", "context scan", "residual, LayerNorm, ctx store"]
tiles = st[:, 10]
print("waves", st.shape[0], "tiles per wave mean %.2f min %d max %d" % (tiles.mean(), tiles.min(), tiles.max()))
tot = st[:, :10].sum() + st[:, 12:14].sum()
print("cycles per tile (all phases): %.0f ; wave life mean %.0f max %.0f cycles" % (tot / tiles.sum(), st[:, 11].mean(), st[:, 11].max()))
for i, n in enumerate(names):
    print("%-40s %8.0f cycles/tile  %5.1f %%" % (n, st[:, i].sum() / tiles.sum(), 100.0 * st[:, i].sum() / tot))
print("   of the row-load phase: issue of the 48 loads %.0f, ticket wait %.0f cycles/tile (the rest: index requests, load wait, accumulator init)"
      % (st[:, 12].sum() / tiles.sum(), st[:, 13].sum() / tiles.sum()))
