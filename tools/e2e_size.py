import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
import bench
from scann.models.scann_model import HipModel, normalize_config
from scann.utils import PackedDataset
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
rng = np.random.default_rng(0)
base = [bench.synth_packed_batch(rng, 128) for _ in range(64)]
GROUPS = [int(a) for a in sys.argv[1:]] or [8]
for mult in (8, 32):
    batches = base * mult
    mol, eoff, atomic, local, dist, wgt = [0], [0], [], [], [], []
    for b in batches:
        bs = np.repeat(b.mol_offset[:-1], np.diff(b.mol_offset)); deg = np.diff(b.edge_offset)
        local.append(b.edge_col - np.repeat(bs, deg))
        mol.extend((b.mol_offset[1:].astype(np.int64) + mol[-1]).tolist()); eoff.extend((b.edge_offset[1:].astype(np.int64) + eoff[-1]).tolist())
        atomic.append(b.atomic); dist.append(b.edge_dist); wgt.append(b.edge_weight)
    n = len(mol) - 1
    ds = PackedDataset.from_arrays(mol, np.concatenate(atomic), eoff, np.concatenate(local), np.concatenate(dist), np.concatenate(wgt), np.zeros(n, np.float32), batch_size=128)
    for streams in (2, 4):
        os.environ["SCANN_STREAMS"] = str(streams)
        model = HipModel(cfg, device=0, seed=1234)
        for group in GROUPS:
            model.predict_dataset(ds, group=group)
            best = 0
            for rep in range(3):
                t0 = time.perf_counter(); model.predict_dataset(ds, group=group); dt = time.perf_counter() - t0
                best = max(best, n / dt)
            print("molecules", n, "streams", streams, "group", group, "best %.0f molecules/s" % best, flush=True)
        model.engine.close()
