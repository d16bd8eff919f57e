"""All GPUs of one node, one PROCESS per GPU: the strong-scaling inference path without a shared interpreter.

``MultiGpuPredictor`` (multi_gpu.py) drives N handles from N threads of one process; the threads share the GIL for the Python
part of every launch group (0.035 us per molecule alone, 0.09 us summed over four contending threads:
profiles/r03_multi_gpu_rate.jsonl) -- a ceiling of 11-29 M molecules/s, borderline against the 8 x 1.5 M that eight MI355X take.
Here every device gets a worker process of its own (``python -m scann.parallel._mp_worker`` started with subprocess: a fresh
interpreter that has never touched HIP, whatever the parent has done, and that does not re-import the parent's main module),
which connects back over an authenticated local socket, builds its model once and then serves ``predict_dataset`` calls:

* the flat CSR arrays of a ``PackedDataset`` are placed in POSIX shared memory once per dataset (``share``) and mapped by the
  workers -- no pickling of the data per call;
* a call sends every worker its run of batches (contiguous, balanced by edge count like MultiGpuPredictor's) and collects the
  predictions from a shared output array, in dataset order;
* no collective: structures are independent.
"""
from __future__ import annotations

import os
import secrets
import subprocess
import sys
import tempfile
from multiprocessing import shared_memory
from multiprocessing.connection import Listener

import numpy as np

_FIELDS = ("mol_offset", "edge_offset", "atomic", "edge_local", "edge_dist", "edge_weight", "target", "ring", "indexes")




class MultiProcessPredictor:
    """``MultiProcessPredictor(config, weights, devices).predict_dataset(packed_dataset)`` -> ``(y [N], ga | None, targets [N])``
    like ``HipModel.predict_dataset``; ``devices`` defaults to every visible GPU (repeating an id puts two workers on one GPU).
    Use as a context manager or call ``close()``: the workers and the shared segments live until then."""

    def __init__(self, config, weights, devices=None, infer=False):
        if devices is None:
            from .. import _hip

            devices = list(range(_hip.load_library().scann_device_count()))  # counting devices does not initialise the GPU
        if not devices:
            raise RuntimeError("MultiProcessPredictor: no HIP device visible (there is no CPU fallback)")
        if weights is None:
            raise ValueError("MultiProcessPredictor needs the weights (every worker loads the same parameters)")
        self.devices = list(devices)
        pkg_dir = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        w = {k: np.asarray(v, dtype=np.float32) for k, v in weights.items()}
        self._procs, self._conns, self._shared = [], [], {}
        self._dir = tempfile.mkdtemp(prefix="scann_mp_")
        key = secrets.token_bytes(32)
        listener = Listener(os.path.join(self._dir, "sock"), family="AF_UNIX", authkey=key)
        env = dict(os.environ, SCANN_MP_KEY=key.hex(), PYTHONPATH=pkg_dir + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.pop("WORLD_SIZE", None)  # a worker is a single-device process of its own, not a rank
        try:
            for d in self.devices:
                self._procs.append(subprocess.Popen([sys.executable, "-m", "scann.parallel._mp_worker", listener.address], env=env))
            for d in self.devices:
                c = listener.accept()
                c.send((d, config, w, infer))
                self._conns.append(c)
            for c in self._conns:
                self._expect(c, "ready")
        except BaseException:
            self.close()
            raise
        finally:
            listener.close()

    @staticmethod
    def _expect(conn, what):
        kind, val = conn.recv()
        if kind == "error":
            raise RuntimeError("MultiProcessPredictor worker failed:\n" + val)
        if kind != what:
            raise RuntimeError("MultiProcessPredictor: unexpected reply %r" % kind)
        return val

    @staticmethod
    def _put(arr):
        arr = np.ascontiguousarray(arr)
        shm = shared_memory.SharedMemory(create=True, size=max(arr.nbytes, 1))
        np.ndarray(arr.shape, dtype=arr.dtype, buffer=shm.buf)[...] = arr
        return shm, (shm.name, arr.shape, arr.dtype.str)

    def share(self, dataset):
        """Place the dataset's flat arrays in shared memory (once); later calls with the same object reuse them."""
        key = id(dataset)
        if key not in self._shared:
            if getattr(dataset, "cgcnn_table", None) is not None:
                raise ValueError("MultiProcessPredictor: feature='cgcnn' datasets are not supported (use MultiGpuPredictor)")
            desc, keep = {}, []
            for f in _FIELDS:
                a = getattr(dataset, f)
                if a is None:
                    desc[f] = None
                else:
                    shm, d = self._put(a)
                    keep.append(shm)
                    desc[f] = d
            self._shared[key] = (desc, keep)
        return key

    def forget(self, dataset):
        key = id(dataset)
        if key in self._shared:
            for c in self._conns:
                c.send(("forget", key))
            for c in self._conns:
                self._expect(c, "ok")
            for shm in self._shared.pop(key)[1]:
                shm.close()
                shm.unlink()

    def predict_dataset(self, dataset, group=None, want_ga=False):
        from .multi_gpu import MultiGpuPredictor

        if not (hasattr(dataset, "batches") and hasattr(dataset, "edge_offset")):
            raise TypeError("MultiProcessPredictor.predict_dataset takes a PackedDataset")
        n = len(dataset)
        n_struct = len(dataset.indexes)
        if want_ga and not np.array_equal(dataset.indexes, np.arange(n_struct)):
            raise ValueError("want_ga needs the dataset in its natural order (shuffle=False)")
        key = self.share(dataset)
        desc = self._shared[key][0]
        mol, eoff = dataset.mol_offset, dataset.edge_offset
        per_struct = (eoff[mol[1:]] - eoff[mol[:-1]]) + 8 * np.diff(mol)
        sel_cost = per_struct[dataset.indexes].astype(np.float64)
        costs = np.add.reduceat(sel_cost, np.arange(0, len(sel_cost), dataset.batch_size)) if n else []
        runs = MultiGpuPredictor._runs(costs, len(self.devices)) if n else []
        y_shm, y_desc = self._put(np.zeros(n_struct, dtype=np.float32))
        out_desc, keep = {"y": y_desc, "ga": None}, [y_shm]
        if want_ga:
            ga_shm, ga_desc = self._put(np.zeros(int(mol[-1]), dtype=np.float32))
            out_desc["ga"] = ga_desc
            keep.append(ga_shm)
        try:
            for c, (lo, hi) in zip(self._conns, runs):
                c.send(("predict", key, desc, dataset.batch_size, lo, hi, group, want_ga, out_desc))
            for c, _ in zip(self._conns, runs):
                self._expect(c, "done")
            y = np.array(np.ndarray(y_desc[1], dtype=np.float32, buffer=y_shm.buf))
            ga = np.array(np.ndarray(out_desc["ga"][1], dtype=np.float32, buffer=keep[1].buf)) if want_ga else None
        finally:
            for shm in keep:
                shm.close()
                shm.unlink()
        return y, ga, np.asarray(dataset.target[dataset.indexes], dtype=np.float32)

    def close(self):
        for key in list(self._shared):
            for shm in self._shared.pop(key)[1]:
                shm.close()
                try:
                    shm.unlink()
                except FileNotFoundError:
                    pass
        for c in self._conns:
            try:
                c.send(("stop",))
            except Exception:
                pass
        for p in self._procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        self._procs, self._conns = [], []
        d = getattr(self, "_dir", None)
        if d and os.path.isdir(d):
            for f in os.listdir(d):
                try:
                    os.unlink(os.path.join(d, f))
                except OSError:
                    pass
            try:
                os.rmdir(d)
            except OSError:
                pass
            self._dir = None

    __enter__ = lambda self: self  # noqa: E731

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
