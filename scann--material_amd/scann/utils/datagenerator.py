"""Batch iterator with the reference's contract (scann/utils/datagenerator.py:11-135), without Keras."""
from __future__ import annotations

from math import ceil

import numpy as np

import json
import os

from .general import pad_sequence


CGCNN_TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cgcnn_atom_init.json")


def load_cgcnn_table(table=None):
    """The 92-d CGCNN element descriptors the reference keeps as a Python dict (scann/utils/dataset/atomic_data.py:27-531, used
    at datagenerator.py:109-110).  Default: the table shipped as a data file (``utils/data/cgcnn_atom_init.json``, written by
    tools/make_cgcnn_table.py); a mapping {Z: 92 floats}, the path of a JSON file in the public ``atom_init.json`` format
    (keys "1", "2", ...) or ``SCANN_CGCNN_TABLE`` override it.  -> float32 [Zmax+1, 92] with row 0 = zeros (padding), and the
    mask of the elements present."""
    if table is None:
        table = os.environ.get("SCANN_CGCNN_TABLE") or CGCNN_TABLE_PATH
    if isinstance(table, (str, os.PathLike)):
        with open(table) as f:
            table = json.load(f)
        if "ones" in table and "dim" in table:  # the shipped compact form: indices of the ones of every binary vector
            dense = {}
            for k, idx in table["ones"].items():
                v = np.zeros(int(table["dim"]), dtype=np.float32)
                v[idx] = 1.0
                dense[k] = v
            table = dense
    items = {int(k): np.asarray(v, dtype=np.float32) for k, v in table.items()}
    if not items or any(v.shape != (92,) for v in items.values()):
        raise ValueError("CGCNN table: every element needs 92 values")
    out = np.zeros((max(items) + 1, 92), dtype=np.float32)
    has = np.zeros(max(items) + 1, dtype=bool)
    for z, v in items.items():
        out[z] = v
        has[z] = True
    return out, has


class DataIterator:
    """``__getitem__(i) -> (inputs dict, target[B])`` exactly as the reference's keras ``Sequence``:
    per-batch maxima M, N; neighbour sentinel 1000 -> mask then 0; weight column 2 (raw solid angle) when
    ``g_update`` else 3 (normalised); ``atom_mask = atomic != 0``."""

    def __init__(self, data_energy, data_neighbor, batch_size=32, converter=False, use_ring=False,
                 shuffle=False, feature="atomic", g_update=False, atomic_features=None):
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.data_neighbor = data_neighbor
        self.data_energy = data_energy
        self.use_ring = use_ring
        self.weight_index = 2 if g_update else 3
        self.feature = feature
        self.cgcnn_table, self.cgcnn_has = load_cgcnn_table(atomic_features) if feature == "cgcnn" else (None, None)
        self.converter = 1000 if converter else 1.0
        self.on_epoch_end()

    def on_epoch_end(self):
        self.indexes = np.arange(len(self.data_energy))
        if self.shuffle:
            np.random.shuffle(self.indexes)

    def __len__(self):
        return ceil(len(self.data_energy) / self.batch_size)

    def __getitem__(self, idx):
        sel = self.indexes[idx * self.batch_size: (idx + 1) * self.batch_size]
        batch_nei = [self.data_neighbor[i] for i in sel]
        batch_atom = [self.data_energy[i] for i in sel]
        B = len(batch_nei)
        M = max(len(c) for c in batch_nei)
        N = max(len(n) for c in batch_nei for n in c)
        energy = np.array([float(p[1]) * self.converter for p in batch_atom], "float32")
        nbr = np.full((B, M, N), 1000, dtype="int32")
        wgt = np.zeros((B, M, N), dtype="float32")
        dst = np.zeros((B, M, N), dtype="float32")
        wi = self.weight_index
        for b, centers in enumerate(batch_nei):
            for a, lst in enumerate(centers):
                if len(lst):
                    arr = np.asarray([(n[1], n[wi], n[-1]) for n in lst], dtype="float64")
                    k = arr.shape[0]
                    nbr[b, a, :k] = arr[:, 0].astype("int32")
                    wgt[b, a, :k] = arr[:, 1]
                    dst[b, a, :k] = arr[:, 2]
        mask_local = nbr != 1000
        nbr[~mask_local] = 0
        pad_atom = pad_sequence([c[0] for c in batch_atom], padding="post", maxlen=M, value=0, dtype="int32")
        mask_atom = pad_atom != 0
        if self.feature == "cgcnn":  # [B, M, 92] element descriptors instead of atomic numbers (datagenerator.py:109-110)
            zs = pad_atom[mask_atom]
            if zs.size and (zs.max() >= self.cgcnn_table.shape[0] or not self.cgcnn_has[zs].all()):
                missing = sorted({int(z) for z in zs if z >= self.cgcnn_table.shape[0] or not self.cgcnn_has[z]})
                raise KeyError("atomic numbers %s are not in the CGCNN table" % missing)
            pad_atom = self.cgcnn_table[pad_atom]
        inputs = {
            "atomic": pad_atom,
            "atom_mask": np.expand_dims(mask_atom, -1),
            "neighbors": nbr,
            "neighbor_mask": mask_local,
            "neighbor_weight": wgt,
            "neighbor_distance": dst,
        }
        if self.use_ring:
            inputs["ring_aromatic"] = pad_sequence([c[2] for c in batch_atom], padding="post", maxlen=M, value=0,
                                                   dtype="int32")
        return inputs, energy
