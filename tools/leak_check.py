import os, sys, time, resource
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
import bench
from scann.models.scann_model import HipModel, normalize_config
from scann.utils import PackedDataset
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
rng = np.random.default_rng(0)
batches = [bench.synth_packed_batch(rng, 128) for _ in range(64)] * 4
mol, eoff, atomic, local, dist, wgt = [0], [0], [], [], [], []
for b in batches:
    bs = np.repeat(b.mol_offset[:-1], np.diff(b.mol_offset)); deg = np.diff(b.edge_offset)
    local.append(b.edge_col - np.repeat(bs, deg))
    mol.extend((b.mol_offset[1:].astype(np.int64) + mol[-1]).tolist()); eoff.extend((b.edge_offset[1:].astype(np.int64) + eoff[-1]).tolist())
    atomic.append(b.atomic); dist.append(b.edge_dist); wgt.append(b.edge_weight)
n = len(mol) - 1
ds = PackedDataset.from_arrays(mol, np.concatenate(atomic), eoff, np.concatenate(local), np.concatenate(dist), np.concatenate(wgt), np.zeros(n, np.float32), batch_size=128)
model = HipModel(cfg, device=0, seed=1234)
def rss(): return int(open("/proc/self/statm").read().split()[1]) * 4096 / 1e6
import threading
for i in range(241):
    model.predict_dataset(ds)
    if i % 60 == 0:
        print("pass", i, "rss MB %.0f" % rss(), "threads", threading.active_count(), "fds", len(os.listdir("/proc/self/fd")), flush=True)
