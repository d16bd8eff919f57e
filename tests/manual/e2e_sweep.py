#!/usr/bin/env python3
"""Diagnostic: host-inclusive predict_dataset rate against group size and dataset length (what bench.py's end_to_end leg measures)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
os.environ["SCANN_STREAMS"] = "4"
import bench
from scann.models.scann_model import HipModel, normalize_config
from scann.utils import PackedDataset
rng = np.random.default_rng(1)
pool = [bench.synth_packed_batch(rng, 128) for _ in range(128)]
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234)
for reps in (1, 4):
    batches = pool * reps
    mol, eoff, atomic, local, dist, wgt = [0], [0], [], [], [], []
    for b in batches:
        base = np.repeat(b.mol_offset[:-1], np.diff(b.mol_offset))
        local.append(b.edge_col - np.repeat(base, np.diff(b.edge_offset)))
        mol.extend((b.mol_offset[1:].astype(np.int64) + mol[-1]).tolist())
        eoff.extend((b.edge_offset[1:].astype(np.int64) + eoff[-1]).tolist())
        atomic.append(b.atomic); dist.append(b.edge_dist); wgt.append(b.edge_weight)
    n = len(mol) - 1
    ds = PackedDataset.from_arrays(mol, np.concatenate(atomic), eoff, np.concatenate(local), np.concatenate(dist), np.concatenate(wgt),
                                   np.zeros(n, np.float32), batch_size=128)
    for group in (4, 8, 12, 16):
        model.predict_dataset(ds, group=group)
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.0:
            model.predict_dataset(ds, group=group); k += 1
        dt = time.perf_counter() - t0
        print("%6d molecules, group %2d: %.0f molecules/s" % (n, group, k * n / dt), flush=True)
