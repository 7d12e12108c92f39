"""torch.library registration of the two grouping ops: the PyTorch counterpart of the reference's
REGISTER_OP("FusedConvRandomK") / REGISTER_OP("FusedConvSelectK") (tf_ops/2d_conv_*_k/fused_conv.cpp:15-33) and of
REGISTER_KERNEL_BUILDER(... DEVICE_GPU ...) (:176).

    torch.ops.elo.fused_conv_random_k(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W,
                                      K, flag_copy, distance, stride_h, stride_w)
        -> (selected_bhw_idx, valid_idx, valid_in_dis_idx, selected_mask)

Same ten attributes, four inputs and four outputs as the TF op.  Like the reference (GPU-only registration), only
the GPU dispatch key has an implementation: it calls the HIP kernel through the C ABI (fused_conv.py).  A CPU tensor
fails in the dispatcher.  The ops are registered non-differentiable (integer indices; the mask is wrapped in
stop_gradient by every caller, utils/pointnet_util.py:54-55).  Importing this module performs the registration.
"""
import torch

from . import fused_conv

_SCHEMA = ("(Tensor xyz1, Tensor xyz2, Tensor idx_n2, Tensor random_hw, int H, int W, int npoints, int kernel_size_H, "
           "int kernel_size_W, int K, int flag_copy, float distance, int stride_h, int stride_w) "
           "-> (Tensor, Tensor, Tensor, Tensor)")

_lib = torch.library.Library("elo", "DEF")
_lib.define("fused_conv_random_k" + _SCHEMA)
_lib.define("fused_conv_select_k" + _SCHEMA)


def _random_k(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W, K, flag_copy, distance,
              stride_h, stride_w):
    return fused_conv.fused_conv_random_k(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W, K,
                                          flag_copy, distance, stride_h, stride_w)


def _select_k(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W, K, flag_copy, distance,
              stride_h, stride_w):
    return fused_conv.fused_conv_select_k(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W, K,
                                          flag_copy, distance, stride_h, stride_w)


def _meta(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W, K, flag_copy, distance,
          stride_h, stride_w):
    """Shape function (fused_conv.cpp:34-63)."""
    B, KT = xyz2.shape[0], kernel_size_H * kernel_size_W
    return (xyz1.new_empty((B, npoints, K, 3), dtype=torch.int32), xyz1.new_empty((B, npoints, KT, 1)),
            xyz1.new_empty((B, npoints, KT, 1)), xyz1.new_empty((B, npoints, K, 1)))


_lib.impl("fused_conv_random_k", _random_k, "CUDA")      # the "CUDA" dispatch key is the GPU key on ROCm builds
_lib.impl("fused_conv_select_k", _select_k, "CUDA")
_lib.impl("fused_conv_random_k", _meta, "Meta")
_lib.impl("fused_conv_select_k", _meta, "Meta")
