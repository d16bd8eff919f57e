#!/usr/bin/env python3
"""Randomised gradient sweep: the fused backward chains (default) against the modular exact-fp32 kernels (SCANN_TRAIN_FUSED=0, the
path validated against fp64 autograd in tests/test_gpu_training.py) on adversarial batches -- isolated atoms, 60..200-neighbour
atoms, 2-atom molecules beside 200-atom ones, 1-structure batches, with and without dropout.  Two processes' worth of state in one:
the switch is read at scann_train_begin, so two engines are created under different environments.  A third engine runs the same
batches on the plain-fp32 training kernels written for other widths (SCANN_GENERIC=1, csrc/scann_generic_train.hip), an implementation
that shares no kernel with the other two: quick bound 4e-4 of a tensor's rms + 4 x the difference of the two FORWARDS relative to the
rmse (the seed of the backward); a tensor beyond it goes to arbitration -- fp64 autograd of the torch graph with the library's Dropout
masks, and the same graph in fp32 as the floor.  These batches are adversarial on purpose: a two-atom molecule whose GlobalAttention
score k_0 . q_1 nearly cancels (1.5e-3 of the sum of its terms' magnitudes in batch 7797) is divided by that score's norm, and ONE fp32
rounding of gq / gk moves its whole gradient -- and every tensor upstream of the readout with it -- by 2e-4 of the tensor's rms
(tools/debug_plain_grads.py prints the stages; profiles/r06_notes.md section 3).  A tensor more than 16 x the fp32 graph's distance
from fp64 is REPORTED as ill-conditioned.  Round 5: 25 tensors in 4 of 18,376 batches, same factor in every tensor of a batch -- a third
of it the plain pooling backward forming the scores a second time in fp32, the rest the luck of the roundings (the fp32 graph is the lucky
one on batches SELECTED by this ratio).  Round 6 (fp64 scores shared by forward and backward, dense outputs as four partial sums): 0 in a
3,017-batch sweep, and `census=K` answers the symmetric question -- every K-th batch arbitrated whatever the quick bound says, all three
implementations tallied against the fp32 graph (profiles/r06_fuzz_census.txt: plain 0, MFMA 3 of 24,192 tensors beyond 4 x).  Beyond 2e-2 of the
rms, or not finite, it is a MISMATCH -- what a wrong formula or index would read."""
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
over = {kv.split("=")[0]: ast.literal_eval(kv.split("=")[1]) for kv in sys.argv[2:]}
# census=K: EVERY K-th batch goes to the fp64 / fp32 arbitration whatever the quick bound says, and the distances of the plain path AND of the
# MFMA (modular) path from fp64 are tallied against the fp32 graph's -- the symmetric question the quick bound cannot answer: is one
# implementation further from fp64 than the others more often than they are from it?  (profiles/r06_fuzz_census.txt)
CENSUS = int(over.pop("census", 0))
cfg["model"].update(over)  # e.g. g_update=False
w = so.init_weights(cfg, 77, perturb=True)
engines = {}
for mode in ("1", "0"):
    os.environ["SCANN_TRAIN_FUSED"] = mode
    m = HipModel(cfg, w, device=0)
    m.engine.train_begin()
    engines[mode] = m
os.environ["SCANN_GENERIC"] = "1"
m = HipModel(cfg, w, device=0)
os.environ.pop("SCANN_GENERIC")
m.engine.train_begin()
engines["plain"] = m
rng = np.random.default_rng(5)
t_end, n, worst, bad, worst_p, n_arb, n_ill, n_4x = time.time() + budget, 0, {}, 0, {}, 0, 0, 0
census = {"plain": [], "modular": [], "fused": []}  # per arbitrated tensor: distance from fp64 / max(the fp32 graph's distance, 2e-5)


def arbitrate():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_ref
    dr = (seed, drop) if drop else None
    return (torch_ref.loss_and_grads(cfg, w, pk, targets, drop=dr)[2], torch_ref.loss_and_grads(cfg, w, pk, targets, drop=dr, dtype="float32")[2],
            torch_ref.REGULARIZED)
while time.time() < t_end:
    inputs, targets = random_batch(rng, cfg["model"]["g_update"], big=(n % 7 == 0), max_struct=1 if n % 5 == 0 else 8)
    pk = _hip.pack_inputs(inputs)
    if np.any(np.diff(pk.mol_offset) == 1):
        continue  # 1-atom structures are NaN with use_ga_norm, by the reference's formula
    drop, seed = (0.1 if n % 2 else 0.0), 100 + n
    grads, ys, arb = {}, {}, None
    for mode, m in engines.items():
        eng = m.engine
        rb = eng.upload(pk)
        sse = eng.train_forward(rb, targets, dropout=drop, seed=seed)
        ys[mode] = np.asarray(eng.download(rb)[0], np.float64).ravel()
        eng.zero_grads()
        eng.train_backward(rb, sse, pk.n_struct)
        grads[mode] = eng.get_grads()
        rb.free()
    # d loss / d y = (y - t) / (n rmse): two implementations whose y differ by dy hand the backward a seed that differs by dy / rmse
    # of its typical size -- when the predictions sit close to the targets, a rounding-level difference of the FORWARDS is a visible
    # one in every gradient.  The plain path's bound scales with it (fused and modular share one forward: no such term).
    rmse = float(np.sqrt(np.mean((ys["0"] - np.asarray(targets, np.float64).ravel()) ** 2)))
    seed_diff = float(np.max(np.abs(ys["plain"] - ys["0"]))) / max(rmse, 1e-30)
    for k, ref in grads["0"].items():
        if ref.size == 1:
            continue  # predict_property/bias = sum of dy: a cancelling sum of float atomics, identical code in both modes
        scale = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))) + 1e-30
        err = float(np.max(np.abs(grads["1"][k].astype(np.float64) - ref))) / scale
        worst[k] = max(worst.get(k, 0.0), err)
        if not (err < 1e-4) or not np.isfinite(grads["1"][k]).all():
            bad += 1
            print("MISMATCH batch %d (structures %d, atoms %d, edges %d, dropout %.1f): %s err %.3e" % (n, pk.n_struct, pk.n_atom, pk.n_edge, drop, k, err))
        err_p = float(np.max(np.abs(grads["plain"][k].astype(np.float64) - ref))) / scale
        worst_p[k] = max(worst_p.get(k, 0.0), err_p)
        if not (err_p < 4e-4 + 4 * seed_diff) or not np.isfinite(grads["plain"][k]).all():
            # beyond the quick bound: who is right?  fp64 autograd of the torch graph on this batch (with the library's Dropout masks),
            # and the SAME graph in fp32 -- the rule of tests/test_gpu_training.py::check_grads: an fp32 implementation is held to
            # 4 x the distance of the fp32 graph from the fp64 one (an ill-conditioned batch moves every fp32 implementation)
            if arb is None:
                arb = arbitrate()
                n_arb += 1
            reg = 2e-4 * w[k].astype(np.float64) if k.endswith(arb[2]) else 0.0
            r64 = arb[0][k] - reg
            e = lambda gg: float(np.max(np.abs(np.asarray(gg, np.float64).reshape(r64.shape) - r64))) / scale
            e_plain, e_mod, e_t32 = e(grads["plain"][k]), e(grads["0"][k]), e(arb[1][k] - reg)
            gross = not (e_plain <= 2e-2) or not np.isfinite(grads["plain"][k]).all()  # (a wrong formula or index reads 1e-1 .. 1)
            if gross:
                bad += 1
            if not gross and not (e_plain <= max(2e-5, 4 * e_t32)):
                n_4x += 1  # (the review's ruler: beyond 4 x the fp32 graph's distance)
                print("beyond 4 x (plain fp32) batch %d (structures %d, atoms %d, edges %d, dropout %.1f): %s  plain %.3e  modular %.3e  the torch graph in fp32 %.3e"
                      % (n, pk.n_struct, pk.n_atom, pk.n_edge, drop, k, e_plain, e_mod, e_t32))
            if gross or not (e_plain <= max(2e-5, 16 * e_t32)):
                n_ill += 0 if gross else 1
                print(("MISMATCH" if gross else "ill-conditioned") + " (plain fp32) batch %d (structures %d, atoms %d, edges %d, dropout %.1f): %s differs from the modular path by %.3e; "
                      "against fp64 autograd: plain %.3e  modular %.3e  the torch graph in fp32 %.3e"
                      % (n, pk.n_struct, pk.n_atom, pk.n_edge, drop, k, err_p, e_plain, e_mod, e_t32))
    if CENSUS and n % CENSUS == 0:
        if arb is None:
            arb = arbitrate()
        for k, ref in grads["0"].items():
            if ref.size == 1:
                continue
            reg = 2e-4 * w[k].astype(np.float64) if k.endswith(arb[2]) else 0.0
            r64 = arb[0][k] - reg
            scale = float(np.sqrt(np.mean(r64 ** 2))) + 1e-30
            e = lambda gg: float(np.max(np.abs(np.asarray(gg, np.float64).reshape(r64.shape) - r64))) / scale
            floor = max(e(arb[1][k] - reg), 2e-5)
            for name, mode in (("plain", "plain"), ("modular", "0"), ("fused", "1")):
                census[name].append(e(grads[mode][k]) / floor)
    n += 1
if CENSUS:
    print("census: every %d-th batch arbitrated (%d tensors per implementation); distance from fp64 autograd / max(the fp32 graph's distance, 2e-5 of the rms):" % (CENSUS, len(census["plain"])))
    for name in ("plain", "modular", "fused"):
        r = np.asarray(census[name])
        if r.size:
            print("  %-8s median %.2f  90 %% %.2f  99 %% %.2f  max %.1f   beyond 4 x: %d   beyond 16 x: %d" % (name, np.median(r), np.quantile(r, 0.9), np.quantile(r, 0.99), r.max(), int((r > 4).sum()), int((r > 16).sum())))
top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
print("%d batches; largest fused-vs-modular differences (of the tensor's rms): %s" % (n, ", ".join("%s %.2e" % kv for kv in top)))
top = sorted(worst_p.items(), key=lambda kv: -kv[1])[:5]
print("largest plain-fp32-vs-modular differences: %s; %d batches went to the fp64 arbitration, %d tensors sat more than 16 x and %d more than 4 x the fp32 graph's distance from fp64 (none beyond 2e-2 of the rms unless reported as MISMATCH)"
      % (", ".join("%s %.2e" % kv for kv in top), n_arb, n_ill, n_4x))
sys.exit(1 if bad else 0)
