#!/usr/bin/env python3
"""The one-batch drop-in call split three ways: model.predict / Engine.forward_padded / the bare C call on pre-converted arrays (how much
is Python?), the resident forward + sync (how much is the device?), upload_padded + sync (what has to happen before the device can start)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[os.path.join(ROOT,"scann--material_amd"), os.path.join(ROOT,"oracle"), ROOT]
import scann_oracle as so
from scann import _hip
from scann.models.scann_model import HipModel
cfg = so.default_config("qm9"); w = so.init_weights(cfg, 1234)
de, dn = so.synth_dataset(128, 3)
inputs, _ = so.pad_batch(de, dn, True)
m = HipModel(cfg, w, device=0, infer=False)
eng = m.engine
def med(f, n=300):
    for _ in range(30): f()
    t=[]
    for _ in range(n):
        t0=time.perf_counter(); f(); t.append(time.perf_counter()-t0)
    return 1e6*float(np.median(t))
print("model.predict(inputs)            %.1f us" % med(lambda: m.predict(inputs)))
print("engine.forward_padded(inputs)    %.1f us" % med(lambda: eng.forward_padded(inputs, want_ga=False)))
# pre-converted arrays: only the C call
atomic = np.ascontiguousarray(inputs["atomic"], dtype=np.int32); B,M = atomic.shape
amask = np.ascontiguousarray(np.asarray(inputs["atom_mask"]).reshape(B,M), dtype=np.uint8)
nbr = np.ascontiguousarray(inputs["neighbors"], dtype=np.int32); N = nbr.shape[2]
nmask = np.ascontiguousarray(inputs["neighbor_mask"], dtype=np.uint8)
wgt = np.ascontiguousarray(inputs["neighbor_weight"], dtype=np.float32); dst = np.ascontiguousarray(inputs["neighbor_distance"], dtype=np.float32)
y = np.empty(B, np.float32)
P=_hip._ptr
print("scann_forward_padded (C only)    %.1f us" % med(lambda: eng.lib.scann_forward_padded(eng._h, B, M, N, P(atomic), P(amask), P(nbr), P(nmask), P(wgt), P(dst), P(y), None)))
rb = eng.upload_padded(inputs)
def fwd():
    eng.forward_resident(rb, 0); eng.sync()
print("resident forward + sync          %.1f us" % med(fwd))
def up():
    r = eng.upload_padded(inputs); eng.sync(); r.free()
print("upload_padded + sync + free      %.1f us" % med(up))
print("mask dtypes:", inputs["atom_mask"].dtype, inputs["neighbor_mask"].dtype, "shape", nbr.shape)
