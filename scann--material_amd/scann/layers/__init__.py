"""Names the reference exports from ``scann.layers`` (scann/layers/__init__.py).  On this stack the layers are
not Python objects but stages of the HIP forward graph (scann--material_amd/csrc/scann_kernels.hip):

  LocalAttention   -> atom_kernel (query, centre/neighbour thirds of filter_geo) + edge_kernel
  ResidualNorm     -> head of the following atom_kernel
  GlobalAttention  -> atom_kernel mode 2 (query/key) + readout_kernel
  GaussianExpansion-> basis_kernel
  gather_shape     -> global neighbour rows built by scann._hip.pack_inputs

``STAGES`` records that mapping for tooling; there is nothing to deserialise, so ``_CUSTOM_OBJECTS`` is empty.
"""
STAGES = {
    "LocalAttention": ("atom_kernel", "edge_kernel"),
    "ResidualNorm": ("atom_kernel",),
    "GlobalAttention": ("atom_kernel", "readout_kernel"),
    "GaussianExpansion": ("basis_kernel",),
    "gather_shape": ("pack_inputs",),
}
_CUSTOM_OBJECTS = {}
__all__ = ["STAGES", "_CUSTOM_OBJECTS"]
