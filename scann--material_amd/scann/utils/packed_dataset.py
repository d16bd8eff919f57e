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

from .. import _listwalk  # native walker, built by csrc/Makefile (no Python fallback)
from .._hip import slice_dataset


class PackedDataset:
    def __init__(self, data_energy, data_neighbor, batch_size=32, converter=False, use_ring=False, shuffle=False,
                 feature="atomic", g_update=False, atomic_features=None):
        # same keyword set as DataIterator (datagenerator.py:12-23); feature="cgcnn": the batches carry the 92-d element
        # descriptors of their atoms (datagenerator.py:109-110) looked up in the table DataIterator uses
        if feature not in ("atomic", "cgcnn"):
            raise ValueError("feature must be 'atomic' or 'cgcnn'")
        self.cgcnn_table = None
        if feature == "cgcnn":
            from .datagenerator import load_cgcnn_table

            self.cgcnn_table, has = load_cgcnn_table(atomic_features)
        n = len(data_energy)
        self.batch_size, self.shuffle, self.use_ring = batch_size, shuffle, use_ring
        wi = 2 if g_update else 3
        sizes = np.fromiter((len(d[0]) for d in data_energy), dtype=np.int64, count=n)
        self.mol_offset = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(sizes, out=self.mol_offset[1:])
        self.atomic = np.concatenate([np.asarray(d[0], dtype=np.int32) for d in data_energy]) if n else np.zeros(0, np.int32)
        self.target = np.array([float(d[1]) * (1000 if converter else 1.0) for d in data_energy], dtype=np.float32)
        per_struct, deg, local, weight, dist = (np.frombuffer(b, dtype=t) for b, t in zip(
            _listwalk.convert(data_neighbor, wi), (np.int64, np.int64, np.int32, np.float32, np.float32)))
        if not np.array_equal(per_struct, sizes):
            raise ValueError("data_energy and data_neighbor disagree on the number of atoms per structure")
        self.edge_offset = np.zeros(deg.shape[0] + 1, dtype=np.int64)
        np.cumsum(deg, out=self.edge_offset[1:])
        self.edge_local, self.edge_weight, self.edge_dist = local, weight, dist  # edge_local: index INSIDE its structure
        self.ring = np.concatenate([np.asarray(d[2], dtype=np.float32).reshape(-1, 2) for d in data_energy]) if use_ring else None
        if self.cgcnn_table is not None and self.atomic.size and (self.atomic.max() >= self.cgcnn_table.shape[0] or not has[self.atomic].all()):
            missing = sorted({int(z) for z in self.atomic if z >= self.cgcnn_table.shape[0] or not has[z]})
            raise KeyError("atomic numbers %s are not in the CGCNN table" % missing)
        self.on_epoch_end()

    @classmethod
    def from_arrays(cls, mol_offset, atomic, edge_offset, edge_local, edge_dist, edge_weight, target, batch_size=32,
                    shuffle=False, ring=None):
        """A dataset that is already flat: mol_offset[n+1] (atoms), edge_offset[atoms+1], edge_local = neighbour index INSIDE
        its structure, like the preprocessing stores it (voronoi_neighbor.py:38-47)."""
        self = cls.__new__(cls)
        self.batch_size, self.shuffle, self.use_ring = batch_size, shuffle, ring is not None
        self.mol_offset = np.ascontiguousarray(mol_offset, dtype=np.int64)
        self.edge_offset = np.ascontiguousarray(edge_offset, dtype=np.int64)
        self.atomic = np.ascontiguousarray(atomic, dtype=np.int32)
        self.edge_local = np.ascontiguousarray(edge_local, dtype=np.int32)
        self.edge_dist = np.ascontiguousarray(edge_dist, dtype=np.float32)
        self.edge_weight = np.ascontiguousarray(edge_weight, dtype=np.float32)
        self.target = np.ascontiguousarray(target, dtype=np.float32)
        self.ring = np.ascontiguousarray(ring, dtype=np.float32) if ring is not None else None
        self.cgcnn_table = None
        self.on_epoch_end()
        return self

    def on_epoch_end(self):
        self.indexes = np.arange(len(self.target))
        if self.shuffle:
            np.random.shuffle(self.indexes)

    def __len__(self):
        return ceil(len(self.target) / self.batch_size)

    def _slice(self, sel):
        pk = slice_dataset(self.mol_offset, self.edge_offset, self.atomic, self.ring, self.edge_local, self.edge_dist,
                           self.edge_weight, sel)
        if self.cgcnn_table is not None:
            pk.cgcnn = np.ascontiguousarray(self.cgcnn_table[pk.atomic])
        return pk, self.target[sel]

    def batch(self, idx):
        """-> (PackedBatch, targets) of batch `idx` (the structures DataIterator.__getitem__(idx) would hold)."""
        return self._slice(self.indexes[idx * self.batch_size:(idx + 1) * self.batch_size])

    def batches(self, i0, i1):
        """Batches i0 .. i1-1 as ONE PackedBatch (their structures in order) with ONE native slice call, plus the targets --
        what ``concat_packed([self[i][0] for i in range(i0, i1)])`` builds, without the per-batch Python work.  Structures
        are independent and the packed layout has no per-batch padding, so a group of batches is just a longer batch."""
        return self._slice(self.indexes[i0 * self.batch_size:i1 * self.batch_size])

    def batch_part(self, idx, rank, world, to_end=False):
        """Rank ``rank``'s contiguous share of batch ``idx`` (data-parallel training: a rank never packs the structures of
        the other ranks).  Same split as ``scann.parallel.rank_slice`` applied to the whole batch.  ``to_end``: the batch
        also takes every structure after it (a final batch with fewer structures than ranks is folded into its predecessor,
        ``scann.models.trainer.dp_batches``)."""
        sel = self.indexes[idx * self.batch_size:] if to_end else self.indexes[idx * self.batch_size:(idx + 1) * self.batch_size]
        base, rem = divmod(len(sel), world)
        lo = rank * base + min(rank, rem)
        return self._slice(sel[lo:lo + base + (1 if rank < rem else 0)])

    def __getitem__(self, idx):
        return self.batch(idx)
