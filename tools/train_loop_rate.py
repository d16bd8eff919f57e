#!/usr/bin/env python3
"""Training-step rate as the real loop runs it: a NEW batch is uploaded every step (no resident reuse)."""
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
B = 128
pks = [bench.synth_packed_batch(rng, B) for _ in range(16)]
tg = [rng.normal(size=B).astype(np.float32) for _ in pks]
def step(i):
    rb = eng.upload(pks[i % 16])
    sse = eng.train_forward(rb, tg[i % 16], dropout=0.1, seed=i)
    eng.zero_grads(); eng.train_backward(rb, sse, B); eng.allreduce_grads(); eng.adam_step(5e-4)
    rb.free()
for i in range(5): step(i)
n = 50
t0 = time.perf_counter()
for i in range(n): step(i)
eng.sync()
dt = (time.perf_counter() - t0) / n
print("train step incl. upload/free of a fresh batch (batch %d, L=7): %.3f ms -> %.0f molecules/s" % (B, dt * 1e3, B / dt))
