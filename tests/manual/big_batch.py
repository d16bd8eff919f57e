#!/usr/bin/env python3
"""One resident batch of 65,536 molecules (~1.2 M atoms, ~8.8 M edges: > 2^32 bytes per edge tensor) against the same
molecules run 128 at a time: checks the 64-bit indexing of every kernel at scale."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
from scann import _hip
from scann.models.scann_model import HipModel, normalize_config
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234, infer=True)
eng = model.engine
rng = np.random.default_rng(0)
parts = [bench.synth_packed_batch(rng, 128) for _ in range(N // 128)]
big = _hip.concat_packed(parts)
print("atoms %d edges %d (edge tensor %.2f GB)" % (big.n_atom, big.n_edge, big.n_edge * 512 / 1e9))
t0 = time.perf_counter(); rb = eng.upload(big); eng.forward_resident(rb, 0); y, ga = eng.download(rb); dt = time.perf_counter() - t0
print("one batch: %.3f s incl. upload/download" % dt)
worst = 0.0
for k in (0, 1, len(parts) // 2, len(parts) - 1):
    yk, gak = eng.forward(parts[k], want_ga=True)
    lo = 128 * k
    a0 = sum(p.n_atom for p in parts[:k])
    worst = max(worst, float(np.max(np.abs(y[lo:lo + 128] - yk) / np.maximum(np.abs(yk), 1e-3))), float(np.max(np.abs(ga[a0:a0 + parts[k].n_atom] - gak))))
print("max deviation from the 128-molecule runs: %.2e" % worst)
assert worst <= 1e-5 and np.isfinite(y).all()
rb.free()
