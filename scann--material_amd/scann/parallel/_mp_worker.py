"""Worker of ``MultiProcessPredictor`` (multi_proc.py): ``python -m scann.parallel._mp_worker <socket path>`` with the connection
key in ``SCANN_MP_KEY``.  One model on one device, then a loop of predict requests over datasets mapped from shared memory."""
from __future__ import annotations

import os
import sys
import traceback
from multiprocessing import resource_tracker, shared_memory
from multiprocessing.connection import Client

import numpy as np


def attach(desc, untrack=False):
    """desc: {field: (shm name, shape, dtype str) | None} -> ({field: ndarray view}, [SharedMemory]).  ``untrack``: the segment
    belongs to another process -- take it off THIS process' resource tracker, which would otherwise unlink it at exit and
    warn about a leak (the attach side of multiprocessing.shared_memory registers every segment it opens)."""
    arrays, keep = {}, []
    for k, d in desc.items():
        if d is None:
            arrays[k] = None
            continue
        shm = shared_memory.SharedMemory(name=d[0])
        if untrack:
            try:
                resource_tracker.unregister(shm._name, "shared_memory")
            except Exception:
                pass
        keep.append(shm)
        arrays[k] = np.ndarray(d[1], dtype=np.dtype(d[2]), buffer=shm.buf)
    return arrays, keep


def _worker(conn):
    """Worker process: one model on one device, then a loop of predict requests."""
    try:
        device, config, weights, infer = conn.recv()
        from scann.parallel.affinity import pin_to_device

        pin_to_device(device)  # the cores of this device's NUMA node, before the first GPU call and the producer thread
        from scann.models.scann_model import HipModel
        from scann.parallel.multi_gpu import _Run
        from scann.utils.packed_dataset import PackedDataset

        model = HipModel(config, weights, device=device, infer=infer)
        conn.send(("ready", None))
        cache = {}  # dataset token -> (PackedDataset view, [SharedMemory])
        while True:
            msg = conn.recv()
            if msg[0] == "stop":
                break
            if msg[0] == "forget":
                ds, keep = cache.pop(msg[1], (None, []))
                del ds
                for s in keep:
                    s.close()
                conn.send(("ok", None))
                continue
            _, token, desc, batch_size, indexes, lo, hi, group, want_ga, out_desc = msg
            if token not in cache:
                arrays, keep = attach(desc, True)
                ds = PackedDataset.from_arrays(arrays["mol_offset"], arrays["atomic"], arrays["edge_offset"], arrays["edge_local"],
                                               arrays["edge_dist"], arrays["edge_weight"], arrays["target"], batch_size=batch_size,
                                               ring=arrays["ring"])
                cache[token] = (ds, keep)
            ds = cache[token][0]
            # the order and the batch size of THIS call (the parent's dataset may have been reshuffled or re-batched since the last)
            ds.indexes, ds.batch_size = np.asarray(indexes, dtype=np.int64), int(batch_size)
            outs, keep_out = attach(out_desc, True)
            y, ga, _ = model.predict_dataset(_Run(ds, lo, hi), group=group, want_ga=want_ga)
            s0 = lo * batch_size
            outs["y"][s0:s0 + len(y)] = y
            if want_ga:
                a0 = int(ds.mol_offset[ds.indexes[s0]])  # (want_ga: the parent checked that the order is the natural one)
                outs["ga"][a0:a0 + len(ga)] = ga
            del outs
            for s in keep_out:
                s.close()
            conn.send(("done", len(y)))
        model.engine.close()
    except BaseException:  # surfaced in the parent
        try:
            conn.send(("error", traceback.format_exc()))
        except Exception:
            pass



if __name__ == "__main__":
    _key = os.environ.pop("SCANN_MP_KEY", None)
    if _key is None:
        sys.exit("scann.parallel._mp_worker: SCANN_MP_KEY is not set (this module is started by MultiProcessPredictor)")
    _worker(Client(sys.argv[1], family="AF_UNIX", authkey=bytes.fromhex(_key)))
