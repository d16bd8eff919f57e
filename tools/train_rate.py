#!/usr/bin/env python3
"""Training-step rate (config 3 shape: QM9, batch 128, L=7) on one GPU: forward(train) + backward + Adam per step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
from scann.models.scann_model import HipModel, normalize_config
import bench
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234)
eng = model.engine
eng.train_begin()
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pks = [bench.synth_packed_batch(rng, B) for _ in range(4)]
rbs = [eng.upload(p) for p in pks]
tg = [rng.normal(size=B).astype(np.float32) for _ in pks]
def step(i):
    rb, t = rbs[i % 4], tg[i % 4]
    sse = eng.train_forward(rb, t, dropout=0.1, seed=i)
    eng.zero_grads(); eng.train_backward(rb, sse, B); eng.allreduce_grads(); eng.adam_step(5e-4)
for i in range(5): step(i)
n = 50
t0 = time.perf_counter()
for i in range(n): step(i)
eng.sync()
dt = (time.perf_counter() - t0) / n
print("train step (batch %d, L=7): %.3f ms -> %.0f molecules/s" % (B, dt * 1e3, B / dt))
