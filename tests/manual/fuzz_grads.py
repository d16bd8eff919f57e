#!/usr/bin/env python3
"""Randomised gradient sweep: the fused backward chains (default) against the modular exact-fp32 kernels (SCANN_TRAIN_FUSED=0, the
path validated against fp64 autograd in tests/test_gpu_training.py) on adversarial batches -- isolated atoms, 60..200-neighbour
atoms, 2-atom molecules beside 200-atom ones, 1-structure batches, with and without dropout.  Two processes' worth of state in one:
the switch is read at scann_train_begin, so two engines are created under different environments."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), os.path.join(ROOT, "oracle"), ROOT, os.path.dirname(os.path.abspath(__file__))]
import scann_oracle as so
from scann import _hip
from scann.models.scann_model import HipModel, normalize_config
from fuzz_parity_lib import random_batch

import ast
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
cfg = normalize_config(so.default_config("qm9"))
cfg["model"].update(n_attention=3)
cfg["model"].update({kv.split("=")[0]: ast.literal_eval(kv.split("=")[1]) for kv in sys.argv[2:]})  # e.g. g_update=False
w = so.init_weights(cfg, 77, perturb=True)
engines = {}
for mode in ("1", "0"):
    os.environ["SCANN_TRAIN_FUSED"] = mode
    m = HipModel(cfg, w, device=0)
    m.engine.train_begin()
    engines[mode] = m
rng = np.random.default_rng(5)
t_end, n, worst, bad = time.time() + budget, 0, {}, 0
while time.time() < t_end:
    inputs, targets = random_batch(rng, cfg["model"]["g_update"], big=(n % 7 == 0), max_struct=1 if n % 5 == 0 else 8)
    pk = _hip.pack_inputs(inputs)
    if np.any(np.diff(pk.mol_offset) == 1):
        continue  # 1-atom structures are NaN with use_ga_norm, by the reference's formula
    drop, seed = (0.1 if n % 2 else 0.0), 100 + n
    grads = {}
    for mode, m in engines.items():
        eng = m.engine
        rb = eng.upload(pk)
        sse = eng.train_forward(rb, targets, dropout=drop, seed=seed)
        eng.zero_grads()
        eng.train_backward(rb, sse, pk.n_struct)
        grads[mode] = eng.get_grads()
        rb.free()
    for k, ref in grads["0"].items():
        if ref.size == 1:
            continue  # predict_property/bias = sum of dy: a cancelling sum of float atomics, identical code in both modes
        scale = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))) + 1e-30
        err = float(np.max(np.abs(grads["1"][k].astype(np.float64) - ref))) / scale
        worst[k] = max(worst.get(k, 0.0), err)
        if not (err < 1e-4) or not np.isfinite(grads["1"][k]).all():
            bad += 1
            print("MISMATCH batch %d (structures %d, atoms %d, edges %d, dropout %.1f): %s err %.3e" % (n, pk.n_struct, pk.n_atom, pk.n_edge, drop, k, err))
    n += 1
top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
print("%d batches; largest fused-vs-modular differences (of the tensor's rms): %s" % (n, ", ".join("%s %.2e" % kv for kv in top)))
sys.exit(1 if bad else 0)
