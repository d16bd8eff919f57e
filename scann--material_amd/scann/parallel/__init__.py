"""Data-parallel helpers: structures are independent units (attention.py:136 gathers inside one
batch row; GlobalAttention reduces inside one row), so a batch shards across GPUs with no collective on the data
path -- each rank runs its shard on its own handle and the host concatenates the per-structure outputs.  Training adds
the RCCL exchanges inside libscann_hip.so; the ranks find each other through ``Rendezvous`` (loopback TCP, no torch)."""
from .affinity import cpus_for_device, pin_to_device
from .launch import spawn_ranks
from .multi_gpu import MultiGpuPredictor
from .multi_proc import MultiProcessPredictor
from .rendezvous import Rendezvous
from .shard import concat_outputs, rank_slice, slice_packed, split_packed

__all__ = ["split_packed", "slice_packed", "rank_slice", "concat_outputs", "MultiGpuPredictor", "MultiProcessPredictor", "Rendezvous", "spawn_ranks", "cpus_for_device", "pin_to_device"]
