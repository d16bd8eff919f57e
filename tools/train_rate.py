#!/usr/bin/env python3
"""Training-step time of an architecture variant on resident synthetic QM9-shaped batches:
    python tools/train_rate.py [batch] [key=value ...]      e.g.  g_update=False   use_attn_norm=False   n_attention=8
(env DROP=<rate>: the rate of the Dropout(0.1) layers, 0 switches them off)"""
import os, sys, time, ast
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
import bench
from scann.models.scann_model import HipModel, normalize_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
over = {}
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    over[k] = ast.literal_eval(v)
cfg = normalize_config({"model": dict(bench.QM9_MODEL, **over), "hyper": {"target": "homo"}})
DROP = float(os.environ.get("DROP", "0.1"))
eng = HipModel(cfg, device=0, seed=1234).engine
eng.train_begin()
rng = np.random.default_rng(0)
pool = [eng.upload(bench.synth_packed_batch(rng, B)) for _ in range(8)]
tg = [rng.normal(size=B).astype(np.float32) for _ in pool]
for i in range(20):
    eng.train_step(pool[i % 8], tg[i % 8], 5e-4, dropout=DROP, seed=i)
n = 300
t0 = time.perf_counter()
for i in range(n):
    eng.train_step(pool[i % 8], tg[i % 8], 5e-4, dropout=DROP, seed=20 + i)
dt = time.perf_counter() - t0
print("batch %d %s dropout %g: %.3f ms per step = %.0f molecules/s" % (B, over or "SCANN+", DROP, dt / n * 1e3, n * B / dt))
