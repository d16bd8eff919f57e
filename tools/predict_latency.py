#!/usr/bin/env python3
"""Wall time of the reference's literal call -- `model.predict(one padded batch of 128)` (scann_model.py:315-319 as predict_model.py
calls it) -- split into its host and device parts: padded dict -> packed CSR (`pack_inputs`), upload + forward + download
(`scann_forward`), and the resident forward alone.
  python3 tools/predict_latency.py [molecules per batch = 128]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scann--material_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scann_oracle as so
from scann import _hip
from scann.models.scann_model import HipModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = so.default_config("qm9")
w = so.init_weights(cfg, 1, perturb=True)
de, dn = so.synth_dataset(n, 0)
inputs, _ = so.pad_batch(de, dn, True)
model = HipModel(cfg, w, device=0, infer=True)
for _ in range(20):
    model.predict(inputs)

def timeit(f, reps=300):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    return float(np.median(t)), float(np.percentile(t, 90))

pk = _hip.pack_inputs(inputs)
rb = model.engine.upload(pk)
def resident():
    model.engine.forward_resident(rb, 0); model.engine.sync()
rows = [("model.predict(inputs): the whole call", timeit(lambda: model.predict(inputs))),
        ("  pack_inputs (padded dict -> packed CSR, host)", timeit(lambda: _hip.pack_inputs(inputs))),
        ("  engine.forward(packed): upload + forward + download", timeit(lambda: model.engine.forward(pk))),
        ("  forward of the resident batch + sync", timeit(resident))]
for name, (med, p90) in rows:
    print("%-58s median %8.1f us   p90 %8.1f us   -> %.0f molecules/s" % (name, med, p90, n / med * 1e6))

# the pieces of engine.forward as separate calls (resident-batch entry points)
def pieces():
    t0 = time.perf_counter(); r = model.engine.upload(pk)
    t1 = time.perf_counter(); model.engine.forward_resident(r, 0)
    t2 = time.perf_counter(); model.engine.download(r)
    t3 = time.perf_counter(); r.free()
    t4 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2, t4 - t3
acc = np.array([pieces() for _ in range(300)]) * 1e6
for name, col in zip(("upload (plan + staging + H2D enqueue)", "forward_resident (15 launches enqueued)", "download (D2H + wait for the stream)", "free"), np.median(acc, 0)):
    print("  %-56s median %8.1f us" % (name, col))
