#!/usr/bin/env python3
"""Resident-input forward rate of the shipped configurations other than the headline's: MP2018-shaped crystals (L = 9, 95 species,
12 neighbours per atom), qm9_std (L = 8), the base branch (g_update = False), on launch groups of comparable edge counts.
  python3 tools/config_rates.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scann_oracle as so
from scann import _hip
from scann.models.scann_model import HipModel

for label, name, kind, n, over in (("qm9 (headline config)", "qm9", "qm9", 1536, {}), ("qm9_std (L = 8)", "qm9_std", "qm9", 1536, {}),
                                   ("mp2018 (L = 9, crystals)", "mp2018", "mp2018", 384, {}), ("qm9, base branch (g_update = False)", "qm9", "qm9", 1536, {"g_update": False})):
    cfg = so.default_config(name)
    cfg["model"].update(over)
    de, dn = so.synth_dataset(n, 3, kind)
    pk = _hip.pack_inputs(so.pad_batch(de, dn, cfg["model"].get("g_update", True))[0])
    model = HipModel(cfg, device=0, seed=1, infer=True)
    rb = model.engine.upload(pk)
    for _ in range(5):
        model.engine.forward_resident(rb, 0)
    model.engine.sync()
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        model.engine.forward_resident(rb, 0)
        k += 1
    model.engine.sync()
    dt = time.perf_counter() - t0
    print("%-38s %5d structures %7d atoms %8d edges per launch sequence: %9.0f structures/s  %6.1f M atoms/s  %6.1f M edge-layers/s" % (
        label, pk.n_struct, pk.n_atom, pk.n_edge, k * pk.n_struct / dt, k * pk.n_atom / dt / 1e6, k * pk.n_edge * cfg["model"]["n_attention"] / dt / 1e6), flush=True)
    rb.free()
