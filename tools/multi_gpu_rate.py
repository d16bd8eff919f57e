#!/usr/bin/env python3
"""STRONG-scaling, host-inclusive inference rate over S134k-sized data (BASELINE configs[1] shapes): ONE host dataset of
130,831 synthetic QM9-shaped molecules (flat CSR in host memory) -> MultiGpuPredictor.predict_dataset over N devices
(one handle + host thread per device; native slicing, upload, forward, download; no collective) -> predictions in order.

  python tools/multi_gpu_rate.py [n_devices ...]        e.g.  python tools/multi_gpu_rate.py 1 2 4 8

On a one-GPU box `1` is the real number; `2` puts two handles on the same device (rehearsal of the threading, not scaling).
Prints one JSON line per device count."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "scann--material_amd"), ROOT]
os.environ.setdefault("SCANN_STREAMS", "2")
import bench  # noqa: E402
from scann import _hip  # noqa: E402
from scann.models.scann_model import normalize_config  # noqa: E402
from scann.parallel import MultiGpuPredictor  # noqa: E402
from scann.utils import PackedDataset  # noqa: E402

N, B = 130831, 128
PROCESSES = "--processes" in sys.argv  # one worker process per device (MultiProcessPredictor) instead of one thread
counts = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1]
cfg = normalize_config({"model": dict(bench.QM9_MODEL), "hyper": {"target": "homo", "batch_size": B}})
rng = np.random.default_rng(0)
t0 = time.perf_counter()
pool = [bench.synth_packed_batch(rng, B) for _ in range(64)]  # 8,192 distinct molecules, tiled to S134k size
mol, eoff, atomic, local, dist, wgt = [0], [0], [], [], [], []
n_done = 0
while n_done < N:
    b = pool[(n_done // B) % len(pool)]
    take = min(B, N - n_done)
    a1 = int(b.mol_offset[take])
    e1 = int(b.edge_offset[a1])
    base = np.repeat(b.mol_offset[:take], np.diff(b.mol_offset[:take + 1]))
    local.append(b.edge_col[:e1] - np.repeat(base, np.diff(b.edge_offset[:a1 + 1])))
    mol.extend((b.mol_offset[1:take + 1].astype(np.int64) + mol[-1]).tolist())
    eoff.extend((b.edge_offset[1:a1 + 1].astype(np.int64) + eoff[-1]).tolist())
    atomic.append(b.atomic[:a1]); dist.append(b.edge_dist[:e1]); wgt.append(b.edge_weight[:e1])
    n_done += take
ds = PackedDataset.from_arrays(mol, np.concatenate(atomic), eoff, np.concatenate(local), np.concatenate(dist), np.concatenate(wgt),
                               np.zeros(N, np.float32), batch_size=B)
t_build = time.perf_counter() - t0
ndev = _hip.load_library().scann_device_count()

# Where does a device thread's time go?  Every entry point of the library is wrapped with a timer (ctypes releases the GIL inside
# the call): thread wall time - time inside the library = Python time of that thread, which is what the threads of ONE process
# have to share (the GIL).  N device threads can be fed as long as the sum of their Python shares stays below the wall time.
import threading  # noqa: E402

_tl = threading.local()
_lib = _hip.load_library()
for _name, _, _ in _hip.SYMBOLS:
    _fn = getattr(_lib, _name)

    def _timed(*a, _fn=_fn):
        t = time.perf_counter()
        try:
            return _fn(*a)
        finally:
            _tl.inside = getattr(_tl, "inside", 0.0) + time.perf_counter() - t
    setattr(_lib, _name, _timed)
_orig_work = MultiGpuPredictor.predict_dataset
_py_share = []
_orig_hm = None
from scann.models.scann_model import HipModel  # noqa: E402

_hm_predict = HipModel.predict_dataset


def _hm_timed(self, *a, **k):
    _tl.inside = 0.0
    w0 = time.perf_counter()
    out = _hm_predict(self, *a, **k)
    wall = time.perf_counter() - w0
    _py_share.append({"wall_s": wall, "in_library_s": _tl.inside, "python_s": wall - _tl.inside, "structures": int(len(out[0]))})
    return out


HipModel.predict_dataset = _hm_timed
for n in counts:
    devices = [d % max(ndev, 1) for d in range(n)]
    if PROCESSES:
        from scann.models.scann_model import keras_default_init
        from scann.parallel import MultiProcessPredictor

        w0 = HipModel(cfg, device=0, seed=1234)
        weights = w0.get_weights()
        w0.engine.close()
        with MultiProcessPredictor(cfg, weights, devices=devices) as multi:
            multi.predict_dataset(ds, group=8)  # warm: workers map the shared dataset, allocator caches, clocks
            t0 = time.perf_counter()
            y, _, _ = multi.predict_dataset(ds, group=8)
            dt = time.perf_counter() - t0
        assert y.shape == (N,) and np.isfinite(y).all()
        print(json.dumps({"metric": "QM9 molecules/s forward, host-inclusive, strong scaling", "value": N / dt, "unit": "molecules/s",
                          "n_workers": n, "devices": devices, "distinct_devices": len(set(devices)), "molecules": N, "seconds": dt,
                          "path": "host PackedDataset in shared memory -> one worker PROCESS per device: slice, upload, forward, download; "
                                  "outputs written to a shared array in dataset order"}), flush=True)
        continue
    multi = MultiGpuPredictor(cfg, None, devices=devices, seed=1234)
    multi.predict_dataset(ds, group=8)  # warm: allocator caches, clocks
    t0 = time.perf_counter()
    y, _, _ = multi.predict_dataset(ds, group=8)
    dt = time.perf_counter() - t0
    assert y.shape == (N,) and np.isfinite(y).all()
    st = _py_share[-n:]
    cpu_us = 1e6 * sum(s["python_s"] for s in st) / N  # Python (GIL-held) microseconds per molecule, summed over the device threads
    print(json.dumps({"metric": "QM9 molecules/s forward, host-inclusive, strong scaling", "value": N / dt, "unit": "molecules/s",
                      "n_handles": n, "devices": devices, "distinct_devices": len(set(devices)), "molecules": N, "seconds": dt,
                      "python_us_per_molecule": cpu_us,
                      "host_ceiling_molecules_per_s": 1e6 / cpu_us,  # the device threads of one process share the GIL for this part
                      "per_thread": st,
                      "path": "host PackedDataset -> per-device thread: slice, upload, forward, download; outputs concatenated in order",
                      "dataset_build_s": t_build}), flush=True)
    for m in multi.models:
        m.engine.close()
