from __future__ import annotations

import numpy as np

from .._hip import PackedBatch


def rank_slice(n_items, rank, world):
    """Contiguous [lo, hi) share of ``n_items`` for ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slice_packed(packed, lo, hi):
    """Structures [lo, hi) of a packed batch as a packed batch of their own (offsets and neighbour rows rebased; the
    optional ring / cgcnn atom features travel with their atoms)."""
    mol = packed.mol_offset.astype(np.int64)
    eoff = packed.edge_offset.astype(np.int64)
    a0, a1 = int(mol[lo]), int(mol[hi])
    e0, e1 = int(eoff[a0]), int(eoff[a1])
    cut = lambda x: x[a0:a1] if x is not None else None  # noqa: E731
    return PackedBatch(cut(packed.atomic), mol[lo:hi + 1] - a0, eoff[a0:a1 + 1] - e0, packed.edge_col[e0:e1] - a0,
                       packed.edge_dist[e0:e1], packed.edge_weight[e0:e1], ring=cut(packed.ring), cgcnn=cut(packed.cgcnn))


def split_packed(packed, n_shards):
    """Cut a packed batch into ``n_shards`` contiguous runs of whole structures with balanced edge counts
    (edges carry ~3/4 of the FLOPs).  Every shard gets at least one structure when there are enough of them.
    Returns the list of PackedBatch shards; concatenating their outputs in order restores the batch order."""
    B = packed.n_struct
    n_shards = max(1, min(n_shards, B))
    mol = packed.mol_offset.astype(np.int64)
    eoff = packed.edge_offset.astype(np.int64)
    edges_before = eoff[mol]  # edges preceding each structure boundary, length B+1
    cost = edges_before + 8 * mol  # edge work plus a per-atom term
    cuts = [0]
    for s in range(1, n_shards):
        target = cost[-1] * s / n_shards
        c = int(np.searchsorted(cost, target))
        c = min(max(c, cuts[-1] + 1), B - (n_shards - s))
        cuts.append(c)
    cuts.append(B)
    return [slice_packed(packed, lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]


def concat_outputs(parts):
    """[(y_shard, ga_shard), ...] in shard order -> (y, ga) of the whole batch."""
    ys = np.concatenate([p[0] for p in parts])
    gas = [p[1] for p in parts]
    return ys, (np.concatenate(gas) if all(g is not None for g in gas) else None)
