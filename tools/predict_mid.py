#!/usr/bin/env python3
"""`model.predict(padded arrays)` at mid sizes: chunked pipeline against one launch sequence (see tools/predict_big.py)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd")); sys.path.insert(0, ROOT)
from scann.models.scann_model import HipModel, normalize_config
import bench
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234)
rng = np.random.default_rng(0)
M, N = 29, 12
default_big = HipModel.BIG_PREDICT
for B in (1024, 2048, 4096, 8192, 16384):
    na = np.clip(np.round(rng.normal(18, 2.9, B)), 3, 29).astype(int)
    amask = np.arange(M)[None, :] < na[:, None]
    deg = rng.integers(4, 13, size=(B, M))
    nmask = (np.arange(N)[None, None, :] < deg[:, :, None]) & amask[:, :, None]
    inputs = {"atomic": np.where(amask, rng.choice([1, 6, 7, 8, 9], size=(B, M)), 0).astype(np.int32), "atom_mask": amask[..., None].astype(np.float32),
              "neighbors": np.where(nmask, rng.integers(0, 1 << 30, size=(B, M, N)) % na[:, None, None], 0).astype(np.int32),
              "neighbor_mask": nmask.astype(np.float32), "neighbor_weight": rng.uniform(0.1, 1.0, size=(B, M, N)).astype(np.float32),
              "neighbor_distance": rng.uniform(0.9, 4.0, size=(B, M, N)).astype(np.float32)}
    out = []
    for big in (1, 1 << 40):  # chunked, one launch sequence
        HipModel.BIG_PREDICT = big
        model.predict(inputs)
        t = []
        for _ in range(7):
            t0 = time.perf_counter(); model.predict(inputs); t.append(time.perf_counter() - t0)
        out.append(float(np.median(t)))
    print("B %6d: chunked %8.2f ms (%.2f M molecules/s)   one launch sequence %8.2f ms (%.2f M)" % (B, 1e3 * out[0], B / out[0] / 1e6, 1e3 * out[1], B / out[1] / 1e6), flush=True)
HipModel.BIG_PREDICT = default_big
