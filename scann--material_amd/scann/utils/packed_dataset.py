"""Host batch packer: the reference rebuilds every batch from nested Python lists
(`DataIterator.__getitem__`, datagenerator.py:69-135; `pad_sequence` / `pad_nested_sequences`, general.py:14-50), which is
orders of magnitude slower than the GPU forward.  Here the object arrays written by the reference's preprocessing
(`[[species, idx, solid_angle, ratio, distance], ...]` per atom, voronoi_neighbor.py:38-47; `[Atomic, target(, ring)]`
rows, general.py:127-137) are converted ONCE into flat CSR arrays; a batch is then a slice plus an offset rebase.
The semantics pinned by the reference are kept: weight column 2 (raw solid angle) when g_update else 3 (normalised)
(datagenerator.py:48-50), neighbour slot order, targets times `converter`."""
from __future__ import annotations

from math import ceil

import numpy as np

from .._hip import PackedBatch


class PackedDataset:
    def __init__(self, data_energy, data_neighbor, batch_size=32, converter=False, use_ring=False, shuffle=False,
                 feature="atomic", g_update=False):
        if feature != "atomic":
            raise NotImplementedError("PackedDataset covers feature='atomic'")
        n = len(data_energy)
        self.batch_size, self.shuffle, self.use_ring = batch_size, shuffle, use_ring
        wi = 2 if g_update else 3
        sizes = np.fromiter((len(d[0]) for d in data_energy), dtype=np.int64, count=n)
        self.mol_offset = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(sizes, out=self.mol_offset[1:])
        self.atomic = np.concatenate([np.asarray(d[0], dtype=np.int32) for d in data_energy]) if n else np.zeros(0, np.int32)
        self.target = np.array([float(d[1]) * (1000 if converter else 1.0) for d in data_energy], dtype=np.float32)
        deg = np.fromiter((len(lst) for c in data_neighbor for lst in c), dtype=np.int64, count=int(sizes.sum()))
        self.edge_offset = np.zeros(deg.shape[0] + 1, dtype=np.int64)
        np.cumsum(deg, out=self.edge_offset[1:])
        flat = [e for c in data_neighbor for lst in c for e in lst]
        arr = np.asarray([(e[1], e[wi], e[-1]) for e in flat], dtype=np.float64).reshape(-1, 3)
        self.edge_local = arr[:, 0].astype(np.int32)  # neighbour index INSIDE its structure
        self.edge_weight = arr[:, 1].astype(np.float32)
        self.edge_dist = arr[:, 2].astype(np.float32)
        self.ring = np.concatenate([np.asarray(d[2], dtype=np.float32).reshape(-1, 2) for d in data_energy]) if use_ring else None
        self.on_epoch_end()

    def on_epoch_end(self):
        self.indexes = np.arange(len(self.target))
        if self.shuffle:
            np.random.shuffle(self.indexes)

    def __len__(self):
        return ceil(len(self.target) / self.batch_size)

    def batch(self, idx):
        """-> (PackedBatch, targets) of batch `idx` (the structures DataIterator.__getitem__(idx) would hold)."""
        sel = self.indexes[idx * self.batch_size:(idx + 1) * self.batch_size]
        a0, a1 = self.mol_offset[sel], self.mol_offset[sel + 1]
        n_at = a1 - a0
        new_mol = np.zeros(len(sel) + 1, dtype=np.int64)
        np.cumsum(n_at, out=new_mol[1:])
        atom_idx = np.repeat(a0 - new_mol[:-1], n_at) + np.arange(new_mol[-1])  # source atom row of every packed atom
        e0, e1 = self.edge_offset[atom_idx], self.edge_offset[atom_idx + 1]
        deg = e1 - e0
        new_eoff = np.zeros(len(atom_idx) + 1, dtype=np.int64)
        np.cumsum(deg, out=new_eoff[1:])
        edge_idx = np.repeat(e0 - new_eoff[:-1], deg) + np.arange(new_eoff[-1])
        base = np.repeat(np.repeat(new_mol[:-1], n_at), deg)  # first packed atom row of the edge's structure
        pk = PackedBatch(self.atomic[atom_idx], new_mol, new_eoff, self.edge_local[edge_idx] + base,
                         self.edge_dist[edge_idx], self.edge_weight[edge_idx],
                         ring=self.ring[atom_idx] if self.ring is not None else None)
        return pk, self.target[sel]

    def __getitem__(self, idx):
        return self.batch(idx)
