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

import itertools
import os
import secrets
import subprocess
import sys
import tempfile
import threading
import time
import weakref
from multiprocessing import shared_memory
from multiprocessing.connection import Listener

import numpy as np

# the dataset's immutable flat arrays: shared once.  `indexes` (reshuffled by on_epoch_end) and `batch_size` (an attribute the caller
# may change) travel with every predict message instead -- the workers never slice with a stale copy.
_FIELDS = ("mol_offset", "edge_offset", "atomic", "edge_local", "edge_dist", "edge_weight", "target", "ring")
_tokens = itertools.count(1)  # dataset tokens: never reused (id() can be, once a dataset is collected)
START_TIMEOUT = float(os.environ.get("SCANN_MP_START_TIMEOUT", "300"))  # seconds for all workers to connect and load their model




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
        self._collected = []  # tokens of datasets that were garbage-collected: dropped at the next call
        self._dir = tempfile.mkdtemp(prefix="scann_mp_")
        key = secrets.token_bytes(32)
        listener = Listener(os.path.join(self._dir, "sock"), family="AF_UNIX", authkey=key)
        env = dict(os.environ, SCANN_MP_KEY=key.hex(), PYTHONPATH=pkg_dir + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.pop("WORLD_SIZE", None)  # a worker is a single-device process of its own, not a rank
        # (before the try: the handler below sets `stop` -- a Popen that raises, ENOMEM or a bad interpreter, must reach the clean-up, not
        # an UnboundLocalError that hides the reason and leaves the workers already started, the listener and the temp dir behind)
        accepted, failure, stop = [], [], threading.Event()
        try:
            for d in self.devices:
                self._procs.append(subprocess.Popen([sys.executable, "-m", "scann.parallel._mp_worker", listener.address], env=env))
            # accept() has no timeout of its own: do it on a thread and watch the children meanwhile -- a worker that dies before
            # it connects (import error, missing libscann_hip.so, bad PYTHONPATH) must not hang the parent
            try:  # accept() polls (0.2 s) so that the thread ends when start-up is abandoned: closing a listener does not wake a blocked accept()
                listener._listener._socket.settimeout(0.2)
            except Exception:
                pass

            def _accept_all():
                import socket

                try:
                    while len(accepted) < len(self.devices) and not stop.is_set():
                        try:
                            accepted.append(listener.accept())
                        except socket.timeout:
                            continue
                except BaseException as e:  # listener closed under us, authentication failure
                    if not stop.is_set():
                        failure.append(e)

            t = threading.Thread(target=_accept_all, daemon=True)
            t.start()
            deadline = time.monotonic() + START_TIMEOUT
            while t.is_alive():
                t.join(0.05)
                dead = [(i, p.poll()) for i, p in enumerate(self._procs) if p.poll() is not None]
                if dead and len(accepted) < len(self.devices):
                    i, code = dead[0]
                    raise RuntimeError("MultiProcessPredictor: the worker started for device %s exited with code %s during start-up, before "
                                       "every worker had connected (%d of %d had; its stderr has the reason)"
                                       % (self.devices[i], code, len(accepted), len(self.devices)))
                if time.monotonic() > deadline:
                    raise RuntimeError("MultiProcessPredictor: %d of %d workers connected within %.0f s (SCANN_MP_START_TIMEOUT)"
                                       % (len(accepted), len(self.devices), START_TIMEOUT))
            if failure:
                raise RuntimeError("MultiProcessPredictor: accepting the workers failed: %r" % (failure[0],))
            for d, c in zip(self.devices, accepted):
                c.send((d, config, w, infer))
                self._conns.append(c)
            for c in self._conns:
                # the model is being built: keep watching the processes -- ALL of them: connections were accepted in arrival order, so
                # connection i is not necessarily process i's, and a dead worker must not be waited for behind a live one's handle
                while not c.poll(0.05):
                    dead = [p.poll() for p in self._procs if p.poll() is not None]
                    if dead and not c.poll(0):
                        raise RuntimeError("MultiProcessPredictor: a worker exited with code %s while loading its model" % dead[0])
                    if time.monotonic() > deadline:
                        raise RuntimeError("MultiProcessPredictor: a worker did not report ready within %.0f s" % START_TIMEOUT)
                self._expect(c, "ready")
        except BaseException:
            stop.set()  # the accept thread sees it within its 0.2 s poll
            try:
                listener.close()
            except Exception:
                pass
            for p in self._procs:
                if p.poll() is None:
                    p.kill()
            self.close()
            raise
        finally:
            try:
                listener.close()
            except Exception:
                pass

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
        """Place the dataset's flat arrays in shared memory (once); later calls with the same object reuse them.  The token is
        kept ON the dataset object and never reused; when the dataset is collected its segments are dropped (here and in the
        workers) at the next call."""
        key = getattr(dataset, "_scann_mp_token", None)
        if key is None or key not in self._shared:
            if getattr(dataset, "cgcnn_table", None) is not None:
                raise ValueError("MultiProcessPredictor: feature='cgcnn' datasets are not supported (use MultiGpuPredictor)")
            key = next(_tokens)
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
            dataset._scann_mp_token = key
            weakref.finalize(dataset, self._collected.append, key)
        return key

    def _drop(self, key):
        if key in self._shared:
            for c in self._conns:
                c.send(("forget", key))
            for c in self._conns:
                self._expect(c, "ok")
            for shm in self._shared.pop(key)[1]:
                shm.close()
                try:
                    shm.unlink()
                except FileNotFoundError:
                    pass

    def forget(self, dataset):
        key = getattr(dataset, "_scann_mp_token", None)
        if key is not None:
            self._drop(key)

    def predict_dataset(self, dataset, group=None, want_ga=False):
        from .multi_gpu import MultiGpuPredictor

        if not (hasattr(dataset, "batches") and hasattr(dataset, "edge_offset")):
            raise TypeError("MultiProcessPredictor.predict_dataset takes a PackedDataset")
        n = len(dataset)
        n_struct = len(dataset.indexes)
        if want_ga and not np.array_equal(dataset.indexes, np.arange(n_struct)):
            raise ValueError("want_ga needs the dataset in its natural order (shuffle=False)")
        while self._collected:
            self._drop(self._collected.pop())
        key = self.share(dataset)
        desc = self._shared[key][0]
        indexes = np.ascontiguousarray(dataset.indexes, dtype=np.int64)  # as of THIS call (on_epoch_end reshuffles)
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
                c.send(("predict", key, desc, int(dataset.batch_size), indexes, lo, hi, group, want_ga, out_desc))
            for c, _ in zip(self._conns, runs):
                self._expect(c, "done")
            y = np.array(np.ndarray(y_desc[1], dtype=np.float32, buffer=y_shm.buf))
            ga = np.array(np.ndarray(out_desc["ga"][1], dtype=np.float32, buffer=keep[1].buf)) if want_ga else None
        finally:
            for shm in keep:
                shm.close()
                shm.unlink()
        return y, ga, np.asarray(dataset.target[indexes], dtype=np.float32)

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
