#!/usr/bin/env python3
"""Diagnostic: where the host-inclusive dataset path (HipModel.predict_dataset) spends its wall time per launch group -- the native
slice, the upload call (tile plan + reverse adjacency + H2D), the forward enqueue, and the download (the only call that waits for the
device).  If slice + upload + forward exceed the device's time per group, the path is host-bound."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
os.environ.setdefault("SCANN_STREAMS", "2")
import bench
from scann.models.scann_model import HipModel, normalize_config
from scann.utils import PackedDataset

cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
rng = np.random.default_rng(0)
batches = [bench.synth_packed_batch(rng, 128) for _ in range(64)] * 8
mol, eoff, atomic, local, dist, wgt = [0], [0], [], [], [], []
for b in batches:
    base = np.repeat(b.mol_offset[:-1], np.diff(b.mol_offset)); deg = np.diff(b.edge_offset)
    local.append(b.edge_col - np.repeat(base, deg))
    mol.extend((b.mol_offset[1:].astype(np.int64) + mol[-1]).tolist()); eoff.extend((b.edge_offset[1:].astype(np.int64) + eoff[-1]).tolist())
    atomic.append(b.atomic); dist.append(b.edge_dist); wgt.append(b.edge_weight)
n = len(mol) - 1
ds = PackedDataset.from_arrays(mol, np.concatenate(atomic), eoff, np.concatenate(local), np.concatenate(dist), np.concatenate(wgt),
                               np.zeros(n, np.float32), batch_size=128)
group = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = HipModel(cfg, device=0, seed=1234)
eng = model.engine
ns = eng.num_streams()
model.predict_dataset(ds, group=group)
t = dict(slice=0.0, upload=0.0, forward=0.0, download=0.0)
pending, k, ng = [], 0, 0
t_all = time.perf_counter()
for g0 in range(0, len(ds), group):
    t0 = time.perf_counter(); pk, tgt = ds.batches(g0, min(len(ds), g0 + group))
    t1 = time.perf_counter(); rb = eng.upload(pk)
    t2 = time.perf_counter()
    if len(pending) >= ns:
        o = pending.pop(0); eng.download(o); o.release()
    t3 = time.perf_counter(); eng.forward_resident(rb, k); k += 1; pending.append(rb)
    t4 = time.perf_counter()
    t["slice"] += t1 - t0; t["upload"] += t2 - t1; t["download"] += t3 - t2; t["forward"] += t4 - t3; ng += 1
while pending:
    o = pending.pop(0); eng.download(o); o.release()
wall = time.perf_counter() - t_all
print("group %d, %d streams: %d molecules in %.1f ms = %.0f molecules/s; per group of %d molecules: wall %.3f ms | slice %.3f | upload %.3f | forward (enqueue) %.3f | download (wait + copy) %.3f"
      % (group, ns, n, wall * 1e3, n / wall, group * 128, wall / ng * 1e3, t["slice"] / ng * 1e3, t["upload"] / ng * 1e3, t["forward"] / ng * 1e3, t["download"] / ng * 1e3))
# the same groups resident (uploaded ahead): what the device alone does with them, each group's arrays cold in the caches
rbs = []
for g0 in range(0, len(ds), group):
    pk, _ = ds.batches(g0, min(len(ds), g0 + group))
    rbs.append(eng.upload(pk))
for rep in range(3):
    eng.sync()
    t0 = time.perf_counter()
    for i, rb in enumerate(rbs):
        eng.forward_resident(rb, i)
    eng.sync()
    dt = time.perf_counter() - t0
print("resident, forward only: %.0f molecules/s (%.3f ms per group)" % (n / dt, dt / len(rbs) * 1e3))
# upload only (no forward): the host side of the copy
t0 = time.perf_counter()
tmp = []
for g0 in range(0, len(ds), group):
    pk, _ = ds.batches(g0, min(len(ds), g0 + group))
    tmp.append(eng.upload(pk))
eng.sync()
dt = time.perf_counter() - t0
print("slice + upload only: %.3f ms per group" % (dt / len(tmp) * 1e3))
