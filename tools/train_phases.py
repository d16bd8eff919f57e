#!/usr/bin/env python3
"""Diagnostic: where a training step's wall time goes -- forward (synchronous), host issue time of the backward (asynchronous),
and what is left to wait for in the Adam step's sync.  backward issue ~= backward total means the step is host-issue-bound."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
from scann.models.scann_model import HipModel, normalize_config
import bench
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
eng = HipModel(cfg, device=0, seed=1234).engine
eng.train_begin()
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rbs = [eng.upload(bench.synth_packed_batch(rng, B)) for _ in range(4)]
tg = [rng.normal(size=B).astype(np.float32) for _ in rbs]
acc = np.zeros(5)
n = 60
for i in range(n + 5):
    rb, t = rbs[i % 4], tg[i % 4]
    t0 = time.perf_counter(); sse = eng.train_forward(rb, t, dropout=0.1, seed=i)
    t1 = time.perf_counter(); eng.zero_grads(); eng.train_backward(rb, sse, B)
    t2 = time.perf_counter(); eng.sync()
    t3 = time.perf_counter(); eng.adam_step(5e-4)
    t4 = time.perf_counter()
    if i >= 5:
        acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0]
acc *= 1e3 / n
print("batch %d: forward (sync) %.3f ms | backward host issue %.3f ms | backward GPU tail after issue %.3f ms | adam (sync) %.3f ms | step %.3f ms"
      % (B, *acc))
