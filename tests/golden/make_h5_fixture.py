#!/usr/bin/env python3
"""Writes tests/golden/hdf5_lite_fixture.h5 with h5py (libhdf5) -- the pin of scann/utils/hdf5_lite.py, the pure-Python reader
behind the Keras checkpoint importer.  Needs an interpreter with h5py (this image: /opt/conda/bin/python3.9); the expected
values are recomputed from the same seeds by tests/test_keras_import.py."""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def contents():
    rng = np.random.default_rng(42)
    names = ["layer_%d" % i for i in range(37)]  # more links than one symbol-table node holds: exercises the B-tree walk
    return {
        "names": names,
        "kernels": {n: rng.normal(size=(6, 5)).astype(np.float32) for n in names},
        "vec64": rng.normal(size=7),
        "ints": np.arange(12, dtype=np.int32).reshape(3, 4) - 5,
        "be": np.arange(5, dtype=">f4") * 0.5,
        "scalar": np.float64(3.5),
        "config": json.dumps({"class_name": "Functional", "config": {"layers": [{"class_name": "Dense", "config": {"name": "x" * 300}}]}}),
    }


if __name__ == "__main__":
    c = contents()
    f = h5py.File(os.path.join(HERE, "hdf5_lite_fixture.h5"), "w")
    f.attrs["keras_version"] = "2.10.0"          # variable-length string attribute (global heap)
    f.attrs["model_config"] = c["config"]        # long variable-length string
    f.attrs["backend"] = np.bytes_("tensorflow")  # fixed-length string scalar
    g = f.create_group("model_weights")
    g.attrs["layer_names"] = np.array([n.encode() for n in c["names"]])  # fixed-length string array
    for n in c["names"]:
        lg = g.create_group(n)
        lg.attrs["weight_names"] = np.array([("%s/sub/kernel:0" % n).encode()])
        lg.create_dataset("%s/sub/kernel:0" % n, (6, 5), dtype="float32")[...] = c["kernels"][n]
    f.create_dataset("vec64", data=c["vec64"])
    f.create_dataset("ints", data=c["ints"])
    f.create_dataset("be", data=c["be"])
    f.create_dataset("scalar", (), dtype="float64")[()] = c["scalar"]
    f.create_dataset("empty_attr_holder", (2,), dtype="float32").attrs["weight_names"] = np.array([], dtype="S1")
    f.close()
