"""MI355X-native SCANN / SCANN+ (drop-in for the ``scann`` package's forward path).

``from scann.models import SCANN`` keeps working; the Keras graph is replaced by libscann_hip.so.
"""
__all__ = ["models", "parallel", "utils"]  # (the reference's scann.layers are the HIP kernels here: csrc/)
