#!/usr/bin/env python3
"""Diagnostic: per-role phase work time of edge_kernel_persistent (SCANN_STAMPS build, SCANN_PERSIST_MIN=1)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd")); sys.path.insert(0, ROOT)
from scann import _hip
from scann.models.scann_model import HipModel, normalize_config
import bench
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234)
eng = model.engine
rng = np.random.default_rng(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rb = eng.upload(_hip.concat_packed([bench.synth_packed_batch(rng, 128) for _ in range(g)]))
for _ in range(3):
    eng.forward_resident(rb, 0)
eng.sync()
st = eng.debug_stamps(rb, 256).astype(np.int64)
st = st[st[:, 5] > 0]
nk = st[:, 6]
print("workgroups", st.shape[0], "tiles/wg mean %.1f" % nk.mean(), "steps mean %.1f" % st[:, 5].mean())
print("loop cycles mean", st[:, 4].mean(), "-> per tile", (st[:, 4] / nk).mean())
names_m = ["M ph0 G1(slot0)", "M ph1 G1(slot1)", "M ph2 G2(slot0)", "M ph3 G2(slot1)"]
names_v = ["V ph0 AT(1)+stage(1)", "V ph1 RP(0)", "V ph2 RP(1)", "V ph3 AT(0)+stage(0)"]
for i in range(4):
    print("%-22s work cycles per tile-pair %8.0f" % (names_m[i], (st[:, i] / (nk / 2)).mean()))
for i in range(4):
    print("%-22s work cycles per tile-pair %8.0f" % (names_v[i], (st[:, 8 + i] / (nk / 2)).mean()))
