#!/usr/bin/env python3
"""Diagnostic: phase clocks of atom_kernel<true, 0> (ResidualNorm + P1/P3/q projections of a 64-atom tile).
Run with SCANN_HIP_LIB=.../libscann_hip_stamps.so SCANN_STAMP_ATOM=1 (make -C scann--material_amd/csrc stamps)."""
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
for _ in range(5):
    eng.forward_resident(rb, 0)
eng.sync()
st = eng.debug_stamps(rb).astype(np.int64)
order = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12]
names = ["stage x (loads -> planes) + barrier", "GEMM ffn1", "barrier + swish -> planes + barrier", "GEMM ffn2", "residual + statistics + barrier",
         "LayerNorm -> c store + planes + barrier", "GEMM W1", "store P1 + GEMM W3", "store P3", "GEMM Wq", "store q"]
print("tiles", st.shape[0], "total cycles/tile mean", (st[:, 12] - st[:, 0]).mean())
for n, (i, j) in zip(names, zip(order[:-1], order[1:])):
    col = st[:, j] - st[:, i]
    print("%-22s mean %8.0f  median %8.0f  max %8.0f" % (n, col.mean(), np.median(col), col.max()))
# launch-level picture: when the tiles start and end relative to the first start (all tiles of a <= 1,024-tile launch are resident at once)
t0 = st[:, 0].min()
start, end = st[:, 0] - t0, st[:, 12] - t0
print("tile starts: median %d  p90 %d  max %d cycles after the first;  ends: median %d  p90 %d  max %d" %
      (np.median(start), np.percentile(start, 90), start.max(), np.median(end), np.percentile(end, 90), end.max()))
