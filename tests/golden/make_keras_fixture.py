#!/usr/bin/env python3
"""Writes tests/golden/keras_layout_qm9_L2.h5: a small seeded SCANN+ model (configs/model_qm9.yaml widths, n_attention = 2; weights =
oracle.init_weights(cfg, 3, perturb=True)) in the layout Keras 2.10's ModelCheckpoint writes (scann_model.py:166-177), through h5py /
libhdf5 -- NOT through TensorFlow (absent here): the layer / weight names are Keras' auto-naming as worked out in
tools/make_keras_h5_fixture.py.  The tests regenerate the weights from the seed and compare (tests/test_keras_import.py,
tests/test_gpu_parity.py::test_keras_h5_checkpoint_loads_and_predicts), so the file has no .npz twin.

  python3 tests/golden/make_keras_fixture.py          (needs /opt/conda/bin/python3.9 with h5py for the writing step)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scann--material_amd")]
SEED, NAME = 3, "keras_layout_qm9_L2.h5"


def fixture_config():
    import scann_oracle as so
    from scann.models.scann_model import normalize_config

    cfg = normalize_config(so.default_config("qm9"))
    cfg["model"].update(n_attention=2)
    return cfg


def fixture_weights():
    import scann_oracle as so

    return so.init_weights(fixture_config(), SEED, perturb=True)


if __name__ == "__main__":
    cfg, w = fixture_config(), fixture_weights()
    with tempfile.TemporaryDirectory() as d:
        npz = os.path.join(d, "m.npz")
        np.savez(npz, __config__=np.array(json.dumps(cfg)), **w)
        subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "tools", "make_keras_h5_fixture.py"), npz, os.path.join(HERE, NAME)], check=True)
    print(NAME, os.path.getsize(os.path.join(HERE, NAME)), "bytes,", sum(int(v.size) for v in w.values()), "parameters")
