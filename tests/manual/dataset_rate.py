#!/usr/bin/env python3
"""End-to-end rate of the dataset path behind SCANN.evaluate / predict_model.py: PackedDataset slicing on the host +
upload + forward + download, pipelined over the handle's streams (model.predict_dataset)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle"), ROOT]
import scann_oracle as so
from scann.models.scann_model import HipModel, normalize_config
from scann.utils import PackedDataset
import bench
os.environ.setdefault("SCANN_STREAMS", "4")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234)
de, dn = so.synth_dataset(N, 5)
t0 = time.perf_counter(); ds = PackedDataset(de, dn, batch_size=128, g_update=True); t_conv = time.perf_counter() - t0
model.predict_dataset(ds, group=8)
for g in (4, 8, 16):
    t0 = time.perf_counter(); y, _, t = model.predict_dataset(ds, group=g); dt = time.perf_counter() - t0
    print("predict_dataset group=%2d: %d molecules in %.3f s = %.0f molecules/s (one-time CSR conversion %.2f s)" % (g, N, dt, N / dt, t_conv))
