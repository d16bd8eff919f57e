#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in call: model.predict(padded dict) = pack (NumPy) + upload + forward + download."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle"), ROOT]
import scann_oracle as so   # only to build a padded dict the way DataIterator does
from scann import _hip
from scann.models.scann_model import HipModel, normalize_config
import bench
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234)
de, dn = so.synth_dataset(128, 0)
inputs, _ = so.pad_batch(de, dn, True)
for _ in range(3): model.predict(inputs)
n = 50
t0 = time.perf_counter()
for _ in range(n): model.predict(inputs)
t1 = time.perf_counter() - t0
pk = _hip.pack_inputs(inputs)
t0 = time.perf_counter()
for _ in range(n): _hip.pack_inputs(inputs)
t2 = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(n): model.engine.forward(pk, want_ga=False)
t3 = time.perf_counter() - t0
print("predict(padded dict): %.3f ms/batch = %.0f molecules/s | pack_inputs %.3f ms | scann_forward(host buffers) %.3f ms = %.0f molecules/s"
      % (t1 / n * 1e3, 128 * n / t1, t2 / n * 1e3, t3 / n * 1e3, 128 * n / t3))
