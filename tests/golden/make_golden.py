#!/usr/bin/env python3
"""Writes tests/golden/*.npz from the NumPy oracle (oracle/scann_oracle.py).  The reference itself cannot run
here (TensorFlow absent), so these vectors pin the oracle against drift and give the GPU tests
committed inputs/outputs; each file holds the padded inputs, the weight seed + a digest of the weights, the fp64 and
fp32 oracle outputs and per-layer checksums.   Run:  python tests/golden/make_golden.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import scann_oracle as so  # noqa: E402

CASES = {
    # name: (base config, model overrides, hyper overrides, dataset kind, n structures, data seed)
    "qm9_plus": ("qm9", {}, {}, "qm9", 6, 101),
    "qm9_base": ("qm9", {"g_update": False}, {}, "qm9", 5, 102),
    "qm9_no_norms": ("qm9", {"use_attn_norm": False, "use_ga_norm": False, "n_attention": 3}, {}, "qm9", 4, 103),
    "qm9_e_b": ("qm9", {"n_attention": 2}, {"target": "e_b"}, "qm9", 4, 104),
    "mp2018": ("mp2018", {"n_attention": 3}, {}, "mp2018", 3, 105),
    # neighbour lists far from the QM9 shape: atoms with 65 / 100 / 129 neighbours (more than one 64-edge tile), a chain
    # (one or two neighbours per atom: the atoms-per-tile limit binds) and isolated atoms, next to an ordinary molecule
    "dense_and_sparse": ("qm9", {"n_attention": 2}, {}, "dense_and_sparse", 3, 106),
}
WEIGHT_SEED = 4321


def weights_digest(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


def dense_and_sparse(seed):
    rng = np.random.default_rng(seed)

    def rec(j):
        return [6, int(j), float(rng.uniform(0.4, 3.5)), 1.0, float(rng.uniform(0.9, 4.0))]

    A = 130
    deg = {0: 129, 3: 65, 64: 100}
    dense = [[rec(j) for j in rng.choice(np.delete(np.arange(A), a), deg.get(a, int(rng.integers(0, 6))), replace=False)]
             for a in range(A)]
    C = 40
    chain = [[rec(j) for j in (a - 1, a + 1) if 0 <= j < C and a % 7 != 3] for a in range(C)]  # every 7th atom isolated
    de1, dn1 = so.synth_dataset(1, seed)
    de, dn = np.empty(3, dtype=object), np.empty(3, dtype=object)
    de[0], dn[0] = [[int(z) for z in rng.choice([1, 6, 7, 8], A)], 0.0], dense
    de[1], dn[1] = de1[0], dn1[0]
    de[2], dn[2] = [[int(z) for z in rng.choice([1, 6, 7, 8], C)], 0.0], chain
    return de, dn


def build(name):
    base, mo, hy, kind, n, seed = CASES[name]
    cfg = so.default_config(base)
    cfg["model"].update(mo)
    cfg["hyper"].update(hy)
    w = so.init_weights(cfg, WEIGHT_SEED, perturb=True)
    de, dn = dense_and_sparse(seed) if kind == "dense_and_sparse" else so.synth_dataset(n, seed, kind)
    inputs, _ = so.pad_batch(de, dn, g_update=cfg["model"]["g_update"])
    return cfg, w, inputs


def main():
    for name in CASES:
        cfg, w, inputs = build(name)
        inter = {}
        y64, ga64 = so.forward(cfg, w, inputs, np.float64, intermediates=inter)
        y32, ga32 = so.forward(cfg, w, inputs, np.float32)
        sums = {"sum_" + k: np.float64(np.sum(v[np.isfinite(v)])) for k, v in inter.items() if k.startswith(("centers", "context"))}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), weight_seed=WEIGHT_SEED, weights_sha256=weights_digest(w),
                            y64=y64, ga64=ga64, y32=y32, ga32=ga32, **sums, **{"in_" + k: v for k, v in inputs.items()})
        print(name, y64[:3, 0])


if __name__ == "__main__":
    main()
