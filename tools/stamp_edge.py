#!/usr/bin/env python3
"""Diagnostic: phase shares of the edge kernel from in-kernel s_memtime stamps.
Run with SCANN_HIP_LIB=scann--material_amd/lib/libscann_hip_stamps.so (make -C scann--material_amd/csrc stamps)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd")); sys.path.insert(0, ROOT)
from scann.models.scann_model import HipModel, normalize_config
import bench

cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo"}})
model = HipModel(cfg, device=0, seed=1234)
eng = model.engine
rng = np.random.default_rng(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rb = eng.upload(bench.synth_packed_batch(rng, nb))
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):  # argv[2]: forwards before the stamped one (sustained clocks)
    eng.forward_resident(rb, 0)
eng.sync()
st = eng.debug_stamps(rb).astype(np.int64)
order = [0, 1, 2, 3, 4, 5, 8, 9, 7]  # stamp slots in program order
names = ["prologue (loads -> planes) + barrier", "GEMM1", "epilogue 1a: V, T, LN_g statistics + barrier", "epilogue 1b: normalise, geom' store, ang planes, q rows + barrier",
         "GEMM2", "logits from accumulators, K dump + 2 barriers", "softmax + context + barrier", "LayerNorm + ctx store"]
print("tiles", st.shape[0], "total cycles/tile mean", (st[:, 7] - st[:, 0]).mean())
for n, (i, j) in zip(names, zip(order[:-1], order[1:])):
    col = st[:, j] - st[:, i]
    print("%-70s mean %8.0f  median %8.0f  max %8.0f" % (n, col.mean(), np.median(col), col.max()))
if st[:, 6].any():  # finer prologue stamps (thread 0 of the workgroup): descriptor, indices, rows staged
    for nm, (i, j) in (("  prologue: entry -> tile descriptor", (0, 6)), ("  prologue: descriptor -> neighbour indices", (6, 10)),
                       ("  prologue: indices -> rows arrived, split and written to the planes", (10, 11)), ("  prologue: staged -> barrier passed", (11, 1))):
        col = st[:, j] - st[:, i]
        print("%-70s mean %8.0f  median %8.0f  max %8.0f" % (nm, col.mean(), np.median(col), col.max()))
print("kernel span (first start -> last end) cycles:", st[:, 7].max() - st[:, 0].min(), "(s_memtime ticks at 100MHz? see guide: tick = shader cycle)")
if st[:, 12].any() and st[:, 13].any():
    ticks = (st[:, 13] - st[:, 12]).astype(np.float64)
    cyc = (st[:, 7] - st[:, 0]).astype(np.float64)
    ok = ticks > 0
    print("shader clock while the kernel runs: %.0f MHz (sum of s_memtime cycles / sum of s_memrealtime ticks x 100 MHz over %d tiles)"
          % (cyc[ok].sum() / ticks[ok].sum() * 100.0, int(ok.sum())))
