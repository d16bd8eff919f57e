#!/usr/bin/env python3
"""`model.predict(whole padded dataset)`: wall time of the literal call on S134k-sized padded arrays ([131072, 29, 12] neighbour
slots), chunked pipeline (default above HipModel.BIG_PREDICT structures) against one giant launch sequence.
  python3 tools/predict_big.py [structures = 131072]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd")); sys.path.insert(0, ROOT)
from scann.models.scann_model import HipModel, normalize_config
import bench

rng = np.random.default_rng(0)
B, M, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 131072), 29, 12
na = np.clip(np.round(rng.normal(18, 2.9, B)), 3, 29).astype(int)
amask = np.arange(M)[None, :] < na[:, None]
deg = rng.integers(4, 13, size=(B, M))
nmask = (np.arange(N)[None, None, :] < deg[:, :, None]) & amask[:, :, None]
inputs = {"atomic": np.where(amask, rng.choice([1, 6, 7, 8, 9], size=(B, M)), 0).astype(np.int32), "atom_mask": amask[..., None].astype(np.float32),
          "neighbors": np.where(nmask, rng.integers(0, 1 << 30, size=(B, M, N)) % na[:, None, None], 0).astype(np.int32),
          "neighbor_mask": nmask.astype(np.float32), "neighbor_weight": rng.uniform(0.1, 1.0, size=(B, M, N)).astype(np.float32),
          "neighbor_distance": rng.uniform(0.9, 4.0, size=(B, M, N)).astype(np.float32)}
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234)
for label, big in (("chunked pipeline", HipModel.BIG_PREDICT), ("one launch sequence", 1 << 40)):
    HipModel.BIG_PREDICT = big
    model.predict(inputs)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); y = model.predict(inputs); best = min(best, time.perf_counter() - t0)
    print("%-22s %7.1f ms  %.0f molecules/s  (%d structures, %d edges)" % (label, 1e3 * best, B / best, B, int(nmask.sum())))
