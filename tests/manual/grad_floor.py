#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient error of the HIP backward against fp64 autograd, next to the error of the SAME graph run by
torch in fp32 (the rounding floor of any single-precision step).  Prints the worst tensors for L = 2 and L = 7."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT]
import scann_oracle as so
import torch_ref
from scann import _hip
from scann.models.scann_model import HipModel

def err(got, ref):
    out = {}
    for k, r in ref.items():
        scale = max(float(np.sqrt(np.mean(r * r))), 1e-12)
        out[k] = float(np.max(np.abs(got[k].astype(np.float64) - r)) / max(float(np.abs(r).max()), scale))
    return out

for L, n, seed in ((2, 6, 1), (7, 5, 4), (7, 24, 9)):
    cfg = so.default_config("qm9"); cfg["model"]["n_attention"] = L
    w = so.init_weights(cfg, 3, perturb=True)
    de, dn = so.synth_dataset(n, seed)
    inputs, targets = so.pad_batch(de, dn, True)
    pk = _hip.pack_inputs(inputs)
    eng = HipModel(cfg, w, device=0).engine
    eng.train_begin()
    rb = eng.upload(pk)
    sse = eng.train_forward(rb, targets)
    eng.zero_grads(); eng.train_backward(rb, sse, pk.n_struct)
    got = eng.get_grads()
    _, _, ref, _ = torch_ref.loss_and_grads(cfg, w, pk, targets)
    _, _, g32, _ = torch_ref.loss_and_grads(cfg, w, pk, targets, dtype="float32")
    for k in ref:
        if k.endswith(torch_ref.REGULARIZED):
            ref[k] = ref[k] - 2e-4 * w[k].astype(np.float64)
            g32[k] = g32[k] - 2e-4 * w[k].astype(np.float64)
    e_gpu, e_32 = err(got, ref), err(g32, ref)
    worst = sorted(e_gpu, key=lambda k: -e_gpu[k])[:6]
    print("L=%d n=%d: max gpu err %.2e, max torch-fp32 err %.2e, max ratio gpu/fp32 %.1f" % (
        L, n, max(e_gpu.values()), max(e_32.values()), max(e_gpu[k] / max(e_32[k], 1e-9) for k in ref)))
    for k in worst:
        print("   %-42s gpu %.2e   torch fp32 %.2e" % (k, e_gpu[k], e_32[k]))
    rb.free()
