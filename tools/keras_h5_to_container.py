#!/usr/bin/env python3
"""Convert a Keras full-model HDF5 checkpoint of the reference (<save_path>_<target>/models/model_<target>.h5,
scann_model.py:166-177) into this package's weight container (.npz: named fp32 tensors + config JSON).

  python tools/keras_h5_to_container.py model_homo.h5 out.npz [config.yaml]

Pure Python (scann/utils/hdf5_lite.py): needs neither h5py nor TensorFlow.  `config.yaml` (the file train.py dumps next to
the checkpoint) supplies the hyper-parameters the HDF5 file does not determine.  SCANN(config, pretrained="model_homo.h5",
mode="infer") does the same conversion on the fly."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scann--material_amd"))

from scann.models.keras_import import load_keras_h5  # noqa: E402
from scann.models.scann_model import normalize_config  # noqa: E402


def main(argv):
    if len(argv) < 3:
        raise SystemExit(__doc__)
    config = None
    if len(argv) > 3:
        import yaml

        config = yaml.safe_load(open(argv[3]))
    cfg, weights = load_keras_h5(argv[1], config)
    cfg = normalize_config(cfg)
    np.savez(argv[2], __config__=np.array(json.dumps(cfg)), **weights)
    n = sum(int(v.size) for v in weights.values())
    print("wrote %s: %d tensors, %d parameters, n_attention=%d, g_update=%s" % (
        argv[2], len(weights), n, cfg["model"]["n_attention"], cfg["model"]["g_update"]))


if __name__ == "__main__":
    main(sys.argv)
