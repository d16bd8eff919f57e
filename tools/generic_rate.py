#!/usr/bin/env python3
"""Rate of the plain-fp32 forward (csrc/scann_generic.hip) on resident QM9-shaped batches, for the record: widths the MFMA kernels
do not implement, and -- with SCANN_GENERIC=1 -- the 128 / 8 config itself beside the MFMA path; then the training step
(scann_train_step on one resident 128-molecule batch, Dropout 0.1) of the same handles (csrc/scann_generic_train.hip).
  python3 tools/generic_rate.py [batches per launch = 8]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd")); sys.path.insert(0, ROOT)
from scann import _hip
from scann.models.scann_model import HipModel, normalize_config
import bench

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(0)
pk = _hip.concat_packed([bench.synth_packed_batch(rng, 128) for _ in range(G)])
for label, over, env in (("128 x 8 (MFMA kernels)", {}, None), ("128 x 8 (plain fp32, SCANN_GENERIC=1)", {}, "1"),
                         ("64 x 4, global 64, out 64", dict(local_dim=64, num_head=4, global_dim=64, dense_out=64), None),
                         ("256 x 8, global 256, out 256", dict(local_dim=256, num_head=8, global_dim=256, dense_out=256), None)):
    m = dict(bench.QM9_MODEL)
    m.update(over)
    cfg = normalize_config({"model": m, "hyper": {"target": "homo"}})
    if env:
        os.environ["SCANN_GENERIC"] = env
    model = HipModel(cfg, device=0, seed=1234, infer=True)
    os.environ.pop("SCANN_GENERIC", None)
    eng = model.engine
    rb = eng.upload(pk)
    for _ in range(3):
        eng.forward_resident(rb, 0)
    eng.sync()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        eng.forward_resident(rb, 0)
        n += 1
    eng.sync()
    dt = time.perf_counter() - t0
    print("%-40s %10.0f molecules/s  (%.2f ms per %d-molecule launch sequence)" % (label, n * pk.n_struct / dt, 1e3 * dt / n, pk.n_struct))
    rb.free()
    # training step on one batch of 128 (the reference's batch size), resident
    if env:
        os.environ["SCANN_GENERIC"] = env
    tmodel = HipModel(cfg, device=0, seed=1234)
    os.environ.pop("SCANN_GENERIC", None)
    teng = tmodel.engine
    teng.train_begin()
    pk1 = bench.synth_packed_batch(np.random.default_rng(1), 128)
    rb = teng.upload(pk1)
    tgt = np.random.default_rng(2).normal(size=pk1.n_struct).astype(np.float32)
    for i in range(3):
        teng.train_step(rb, tgt, 1e-3, dropout=0.1, seed=i)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        teng.train_step(rb, tgt, 1e-3, dropout=0.1, seed=n)
        n += 1
    dt = time.perf_counter() - t0
    print("%-40s %10.2f ms per training step of %d molecules (%d atoms, %d edges)" % ("", 1e3 * dt / n, pk1.n_struct, pk1.n_atom, pk1.n_edge))
    rb.free()
