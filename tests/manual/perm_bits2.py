#!/usr/bin/env python3
"""Diagnostic: layer-1 K / ang / V / T / q under a permutation of the batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle"), ROOT]
import scann_oracle as so
from scann import _hip
from scann.models.scann_model import HipModel

cfg = so.default_config("qm9")
w = so.init_weights(cfg, 1234, perturb=True)
de, dn = so.synth_dataset(9, 2)
model = HipModel(cfg, w, device=0, infer=True)
eng = model.engine
eng.train_begin()

def run(order):
    inputs, t = so.pad_batch(de[order], dn[order], True)
    pk = _hip.pack_inputs(inputs)
    rb = eng.upload(pk)
    eng.train_forward(rb, np.asarray(t, np.float32), dropout=0.0, seed=1)
    out = {n: eng.debug_read(rb, k, 1) for k, n in ((3, "K"), (4, "ang"), (5, "V"), (6, "T"), (7, "q"), (2, "ctx"), (1, "g"))}
    tiles = _hip.plan_tiles(pk, 64, 24)[1]
    rb.free()
    return pk, out, tiles

ident = np.arange(9)
pk0, a, t0 = run(ident)
perm = np.random.default_rng(1).permutation(9)
pk1, b, t1 = run(perm)
mol0 = pk0.mol_offset
amap = np.concatenate([np.arange(mol0[m], mol0[m + 1]) for m in perm])
e0 = pk0.edge_offset
emap = np.concatenate([np.arange(e0[a_], e0[a_ + 1]) for a_ in amap])
for k in a:
    mp = amap if a[k].shape[0] == pk0.n_atom else emap
    d = a[k][mp] != b[k]
    print(k, "differing elements:", int(d.sum()), "rows:", int(d.any(axis=1).sum()), "max abs %.3e" % float(np.abs(a[k][mp] - b[k]).max()))
d = (a["ctx"][amap] != b["ctx"]).any(axis=1)
rows1 = np.nonzero(d)[0]
print("differing atoms (permuted numbering):", rows1.tolist())
# tile-local position of those atoms in both runs
def where(tiles, atom, pk):
    for t in tiles:
        if t[0] <= atom < t[1]:
            return (int(atom - t[0]), int(pk.edge_offset[atom] - t[2]), int(pk.edge_offset[atom + 1] - t[2]))
for r in rows1[:12]:
    print("atom", int(r), "perm run (local atom, e0, e1):", where(t1, r, pk1), " original run:", where(t0, amap[r], pk0))
