"""Data-parallel inference helpers: structures are independent units (attention.py:136 gathers inside one
batch row; GlobalAttention reduces inside one row), so a batch shards across GPUs with no collective on the data
path -- each rank runs its shard on its own handle and the host concatenates the per-structure outputs."""
from .multi_gpu import MultiGpuPredictor
from .shard import concat_outputs, rank_slice, split_packed

__all__ = ["split_packed", "rank_slice", "concat_outputs", "MultiGpuPredictor"]
