#!/usr/bin/env python3
"""Forward rate of the base SCANN branch (g_update=False: geometry from the raw Gaussian basis in every layer,
attention.py:154-155) on QM9-shaped batches; same engine, 16 batches per launch sequence."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
from scann import _hip
from scann.models.scann_model import HipModel, normalize_config
import bench
m = dict(bench.QM9_MODEL, g_update=False)
cfg = normalize_config({"model": m, "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234)
eng = model.engine
rng = np.random.default_rng(0)
groups = [eng.upload(_hip.concat_packed([bench.synth_packed_batch(rng, 128) for _ in range(16)])) for _ in range(4)]
for i in range(8): eng.forward_resident(groups[i % 4], 0)
eng.sync()
n = 60
t0 = time.perf_counter()
for i in range(n): eng.forward_resident(groups[i % 4], 0)
eng.sync()
dt = time.perf_counter() - t0
print("base SCANN forward: %.0f molecules/s (%.3f ms per 16-batch forward)" % (n * 16 * 128 / dt, dt / n * 1e3))
pr = eng.profile(groups[0])
print({k: round(pr[k], 4) for k in ("ms_basis", "ms_atom", "ms_edge", "ms_readout", "ms_total")})
