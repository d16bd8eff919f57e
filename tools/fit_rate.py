#!/usr/bin/env python3
"""Host-inclusive training rate: the inner loop of trainer.fit (prefetch thread -> upload -> scann_train_step -> download y) over a
host PackedDataset of synthetic QM9-shaped molecules, against the resident-batch step of `bench.py --train`.
    python tools/fit_rate.py [n_batches] [batch]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
import bench
from scann.models.scann_model import HipModel, normalize_config
from scann.models.trainer import _Prefetch, Communicator
from scann.utils import PackedDataset

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(0)
batches = [bench.synth_packed_batch(rng, B) for _ in range(nb)]
mol, eoff, atomic, local, dist, wgt = [0], [0], [], [], [], []
for b in batches:
    base = np.repeat(b.mol_offset[:-1], np.diff(b.mol_offset))
    local.append(b.edge_col - np.repeat(base, np.diff(b.edge_offset)))
    mol.extend((b.mol_offset[1:].astype(np.int64) + mol[-1]).tolist())
    eoff.extend((b.edge_offset[1:].astype(np.int64) + eoff[-1]).tolist())
    atomic.append(b.atomic); dist.append(b.edge_dist); wgt.append(b.edge_weight)
n = len(mol) - 1
ds = PackedDataset.from_arrays(mol, np.concatenate(atomic), eoff, np.concatenate(local), np.concatenate(dist), np.concatenate(wgt),
                               rng.normal(size=n).astype(np.float32), batch_size=B)
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
eng = HipModel(cfg, device=0, seed=1234).engine
eng.train_begin()
comm = Communicator(eng)

def epoch():
    it = 0
    t_up = t_issue = t_end = 0.0
    pending = []
    for shard, tgt in _Prefetch(ds, comm):
        t0 = time.perf_counter(); rb = eng.upload(shard)          # overlaps the step(s) in flight
        t1 = time.perf_counter(); eng.train_step_begin(rb, tgt, 5e-4, dropout=0.1, seed=it)
        t2 = time.perf_counter()
        pending.append(rb)
        if len(pending) == 2:
            eng.train_step_end(); pending.pop(0).release()
        t3 = time.perf_counter()
        t_up += t1 - t0; t_issue += t2 - t1; t_end += t3 - t2
        it += 1
    while pending:
        eng.train_step_end(); pending.pop(0).release()
    return it, t_up, t_issue, t_end

epoch()  # warm
t0 = time.perf_counter(); it, t_up, t_step, t_dl = epoch(); dt = time.perf_counter() - t0
print("fit inner loop: %d steps of %d molecules in %.3f s = %.0f molecules/s (%.3f ms/step: upload %.3f, issuing the step %.3f, waiting for the "
      "oldest step in flight + release %.3f, waiting for the prefetch thread %.3f)"
      % (it, B, dt, it * B / dt, dt / it * 1e3, t_up / it * 1e3, t_step / it * 1e3, t_dl / it * 1e3, (dt - t_up - t_step - t_dl) / it * 1e3))
