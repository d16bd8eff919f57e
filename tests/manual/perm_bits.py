#!/usr/bin/env python3
"""Diagnostic: which tensor first differs in bits when the molecules of a batch are permuted (and is a repeat run identical)?"""
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
eng.set_debug(True)
L = cfg["model"]["n_attention"]

def run(order):
    inputs, _ = so.pad_batch(de[order], dn[order], True)
    pk = _hip.pack_inputs(inputs)
    rb = eng.upload(pk)
    eng.forward_resident(rb, 0)
    eng.sync()
    out = {}
    for l in range(L + 1):
        out["c%d" % l] = eng.debug_read(rb, 0, l)
        out["g%d" % l] = eng.debug_read(rb, 1, l)
        if l >= 1:
            out["x%d" % l] = eng.debug_read(rb, 2, l)
    y, ga = eng.download(rb)
    out["y"] = y
    rb.free()
    return pk, out

ident = np.arange(9)
pk0, a = run(ident)
_, a2 = run(ident)
print("repeat run identical:", all(np.array_equal(a[k], a2[k]) for k in a))
perm = np.random.default_rng(1).permutation(9)
pk1, b = run(perm)
# map atoms / edges of the permuted batch back
mol0, mol1 = pk0.mol_offset, pk1.mol_offset
amap = np.concatenate([np.arange(mol0[m], mol0[m + 1]) for m in perm])          # permuted atom row -> original atom row
e0 = pk0.edge_offset
emap = np.concatenate([np.arange(e0[a_], e0[a_ + 1]) for a_ in amap])
for l in range(L + 1):
    for k, mp in (("c%d" % l, amap), ("g%d" % l, emap), ("x%d" % l, amap)):
        if k in a:
            d = a[k][mp] != b[k]
            print(k, "differing elements:", int(d.sum()), "rows:", int(d.any(axis=1).sum()), "max abs diff %.3e" % float(np.abs(a[k][mp] - b[k]).max()))
print("y equal:", np.array_equal(a["y"][perm], b["y"]))
