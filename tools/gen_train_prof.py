#!/usr/bin/env python3
"""Training steps of ONE generic-width handle on a resident 128-molecule batch, for a kernel trace:
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gen -- python3 tools/gen_train_prof.py [local_dim=64] [heads=4] [steps=20]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd")); sys.path.insert(0, ROOT)
from scann.models.scann_model import HipModel, normalize_config
import bench

d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = dict(bench.QM9_MODEL)
m.update(local_dim=d, num_head=H, global_dim=d, dense_out=d)
if d == 128 and H == 8:
    os.environ["SCANN_GENERIC"] = "1"
model = HipModel(normalize_config({"model": m, "hyper": {"target": "homo"}}), device=0, seed=1234)
eng = model.engine
eng.train_begin()
pk = bench.synth_packed_batch(np.random.default_rng(1), 128)
rb = eng.upload(pk)
tgt = np.random.default_rng(2).normal(size=pk.n_struct).astype(np.float32)
for i in range(steps):
    eng.train_step(rb, tgt, 1e-3, dropout=0.1, seed=i)
print("done", steps)
