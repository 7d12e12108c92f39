"""efficientlo-net_amd -- MI355X-native implementation of EfficientLO-Net's
projection-aware 3D feature hot path (grouping, set-conv, attentive cost
volume, pose warp-refinement, set-upconv) behind the reference's own operator
signatures.  HIP kernels live in csrc/ behind the C ABI of include/elo.h.

The directory name contains a hyphen; import it with
    importlib.import_module("efficientlo-net_amd")
"""
from .fused_conv import fused_conv_random_k, fused_conv_select_k  # noqa: F401
