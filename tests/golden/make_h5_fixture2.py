#!/usr/bin/env python3
"""Writes the second pin of scann/utils/hdf5_lite.py with h5py (libhdf5): tests/golden/hdf5_lite_chunked.h5 (chunked datasets
without filters -- read -- and one gzip-compressed dataset -- refused with a message naming the filter) and
tests/golden/hdf5_lite_latest.h5 (libver='latest': superblock 3, version-2 object headers -- refused with a message naming the
format and the way out).  Needs an interpreter with h5py (this image: /opt/conda/bin/python3.9); the expected values are recomputed
from the same seeds by tests/test_keras_import.py."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def contents():
    rng = np.random.default_rng(7)
    return {"a": rng.normal(size=(37, 128)).astype(np.float32), "b": np.arange(1000, dtype=np.int64).reshape(10, 100) * 3 - 7,
            "c": rng.normal(size=(5, 3, 9)), "d": rng.normal(size=(300,)).astype(">f4")}


if __name__ == "__main__":
    c = contents()
    with h5py.File(os.path.join(HERE, "hdf5_lite_chunked.h5"), "w") as f:
        f.create_dataset("a", data=c["a"], chunks=(8, 48))          # ragged edge chunks in both dimensions
        f.create_dataset("b", data=c["b"], chunks=(1, 7))           # 150 chunks: more than one B-tree leaf
        f.create_dataset("c", data=c["c"], chunks=(5, 3, 9))        # a single chunk
        f.create_dataset("d", data=c["d"], chunks=(64,))            # big-endian elements
        f.create_dataset("never_written", (6, 4), dtype="float32", chunks=(2, 2))
        f.create_dataset("gz", data=c["a"], chunks=(8, 48), compression="gzip", shuffle=True)
    with h5py.File(os.path.join(HERE, "hdf5_lite_latest.h5"), "w", libver="latest") as f:
        f.create_dataset("a", data=c["a"])
