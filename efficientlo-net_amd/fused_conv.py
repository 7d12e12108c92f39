"""fused_conv_random_k / fused_conv_select_k with the reference's signatures.

Drop-in for tf_ops/2d_conv_random_k/fused_conv_random_k.py:14-29 and
tf_ops/2d_conv_select_k/fused_conv_select_k.py:14-29: same positional /
keyword arguments, same four outputs (dtype, shape, order).  Tensors are torch
tensors on the GPU; the work is done by hand-written HIP kernels behind the C
ABI of include/elo.h (csrc/elo_grouping.hip).  The attribute and shape checks
of the TF op (fused_conv.cpp:78-123) are kept, plus the ones it forgot.

`random_hw` stays an input (the caller owns the randomness, e.g.
torch.randperm(KT, dtype=torch.int32)), so the op is a pure function.  It is
not differentiable: indices are integers and every caller wraps the mask in
stop_gradient (utils/pointnet_util.py:54-55,110-111,203-204,277-278).
"""
import ctypes
import math

import torch

from . import _lib


def _check(name, xyz1, xyz2, idx_n2, random_hw, npoints, kernel_size_H, kernel_size_W, K, flag_copy,
           distance, stride_h, stride_w):
    # attribute checks: fused_conv.cpp:78-100
    if npoints <= 0: raise ValueError("FusedConv expects positive npoints")
    if kernel_size_H <= 0: raise ValueError("FusedConv expects positive kernel_size_H")
    if kernel_size_W <= 0: raise ValueError("FusedConv expects positive kernel_size_W")
    if K <= 0: raise ValueError("FusedConv expects positive K")
    if flag_copy not in (0, 1): raise ValueError("FusedConv expects 0 OR 1 flag_copy")
    if not distance > 0: raise ValueError("FusedConv expects positive distance")
    if stride_h <= 0: raise ValueError("FusedConv expects positive stride_h")
    if stride_w <= 0: raise ValueError("FusedConv expects positive stride_w")
    # shape checks: fused_conv.cpp:107-123 (+ W2, channel, dtype, device, contiguity)
    _lib.require_gpu(xyz1, xyz2, idx_n2, random_hw)
    if xyz1.dim() != 4 or xyz1.shape[3] != 3:
        raise ValueError("%s expects (batch_size, H, W, 3) xyz1 shape." % name)
    B, H, W, _ = xyz1.shape
    H2, W2 = math.ceil(H / float(stride_h)), math.ceil(W / float(stride_w))
    if xyz2.dim() != 4 or tuple(xyz2.shape) != (B, H2, W2, 3):
        raise ValueError("%s expects (batch_size, H/stride_h, W/stride_w, 3) xyz2 shape." % name)
    if idx_n2.dim() != 3 or idx_n2.shape[0] != B or idx_n2.shape[1] != npoints or idx_n2.shape[2] != 2:
        raise ValueError("FusedConv expects (batch_size, npoints, 2) idx_n2 shape.")
    if random_hw.dim() != 1 or random_hw.shape[0] != kernel_size_H * kernel_size_W:
        raise ValueError("FusedConv expects (kernel_size_h * kernel_size_w) random_hw shape.")
    if xyz1.dtype != torch.float32 or xyz2.dtype != torch.float32:
        raise TypeError("xyz1/xyz2 must be float32")
    if idx_n2.dtype != torch.int32 or random_hw.dtype != torch.int32:
        raise TypeError("idx_n2/random_hw must be int32")
    if not (xyz1.device == xyz2.device == idx_n2.device == random_hw.device):
        raise ValueError("all inputs must live on the same device")
    return B, H, W, H2, W2


_DENSE_INDEX = {}        # data_ptr -> tensor: index tensors known to be "every pixel, row-major" (get_hw_idx's)


def register_dense_index(idx_n2):
    """Tell the wrappers that `idx_n2` (B, H*W, 2) lists every pixel in row-major order (pointnet_util.get_hw_idx builds
    and registers such tensors): random-k calls with it take the LDS-tiled kernel (elo_fused_conv_random_k_dense)."""
    if len(_DENSE_INDEX) >= 256:
        _DENSE_INDEX.clear()
    _DENSE_INDEX[idx_n2.data_ptr()] = idx_n2
    return idx_n2


def _select_dense_fits(kH, kW, K, flag_copy, stride_h, stride_w):
    """Bounds of elo_fused_conv_select_k_dense (csrc/elo_grouping.hip): K <= 7, flag_copy 0, <= 512 slots, 64 KB of LDS at
    its largest form (16 waves per tile, with the two prefix masks)."""
    RH, RW = kH, 63 // stride_w + kW
    words = 4 * RH * RW + 2 * 64 * 32 + 16 * 64 + 64 + 2 * 16 * 64 + 64 + ((kH * kW + 3) & ~3) + 16 * 128
    return K <= 7 and flag_copy == 0 and kH * kW <= 512 and 4 * words <= 64 * 1024


def _dense_fits(kH, kW, K, stride_h, stride_w):
    """dense_lds_bytes of csrc/elo_grouping.hip for its smaller (2 x 64) tile: window union + hit lists within 64 KB of LDS."""
    RH, RW = 1 // stride_h + kH, 63 // stride_w + kW
    return 4 * ((kH * kW + 7) & ~7) + 16 * RH * RW + 4 * (129 * K + 256) <= 64 * 1024


def _launch(entry, name, xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W, K,
            flag_copy, distance, stride_h, stride_w, want_valid, dense=None):
    # H, W attributes are unused by the reference's Compute as well: the real
    # sizes come from the tensor (fused_conv.cpp:108-110).
    B, H, W, H2, W2 = _check(name, xyz1, xyz2, idx_n2, random_hw, npoints, kernel_size_H, kernel_size_W,
                             K, flag_copy, distance, stride_h, stride_w)
    xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
    idx_n2, random_hw = idx_n2.contiguous(), random_hw.contiguous()
    KT = kernel_size_H * kernel_size_W
    dev = xyz1.device
    if entry == "elo_fused_conv_random_k" and dense is not False and npoints == H * W:
        known = _DENSE_INDEX.get(idx_n2.data_ptr())
        if dense or (known is idx_n2 and _dense_fits(kernel_size_H, kernel_size_W, K, stride_h, stride_w)):
            entry = "elo_fused_conv_random_k_dense"
    elif entry == "elo_fused_conv_select_k" and dense is not False and npoints == H * W:
        known = _DENSE_INDEX.get(idx_n2.data_ptr())
        fits = _select_dense_fits(kernel_size_H, kernel_size_W, K, flag_copy, stride_h, stride_w)
        if dense and not fits:
            raise ValueError("the dense select-k form takes K <= 7, flag_copy = 0 and a window of at most 512 slots")
        if dense or (known is idx_n2 and fits):
            entry = "elo_fused_conv_select_k_dense"
    elif dense:
        raise ValueError("the dense form needs npoints == H*W (every pixel a centre)")
    sel = torch.empty((B, npoints, K, 3), dtype=torch.int32, device=dev)
    mask = torch.empty((B, npoints, K, 1), dtype=torch.float32, device=dev)
    if want_valid:
        valid = torch.empty((B, npoints, KT, 1), dtype=torch.float32, device=dev)
        indis = torch.empty((B, npoints, KT, 1), dtype=torch.float32, device=dev)
    else:
        valid = indis = None
    args = _lib.GroupArgs(B, H, W, H2, W2, npoints, kernel_size_H, kernel_size_W, K, flag_copy,
                          float(distance), stride_h, stride_w, xyz1.data_ptr(), xyz2.data_ptr(),
                          idx_n2.data_ptr(), random_hw.data_ptr(), sel.data_ptr(),
                          valid.data_ptr() if want_valid else None,
                          indis.data_ptr() if want_valid else None, mask.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(getattr(_lib.lib(), entry)(ctypes.byref(args), _lib.stream_ptr(xyz1)))
    return sel, valid, indis, mask


def fused_conv_random_k(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W, K,
                        flag_copy, distance, stride_h, stride_w, want_valid=True, dense=None):
    """First K in-range neighbours in the caller's visiting order.

    Returns (selected_bhw_idx i32 (B,N,K,3), valid_idx f32 (B,N,KT,1),
    valid_in_dis_idx f32 (B,N,KT,1), selected_mask f32 (B,N,K,1)).
    want_valid=False (an extension used by the model path) skips the two
    outputs no caller reads and returns None for them.
    dense (extension): True = the caller vouches that idx_n2 lists EVERY pixel in row-major order (get_hw_idx) and the
    LDS-tiled kernel is used; None = used automatically for index tensors registered with register_dense_index;
    False = always the general kernel.  Same outputs bit for bit either way.
    """
    return _launch("elo_fused_conv_random_k", "FusedConvRandomK", xyz1, xyz2, idx_n2, random_hw, H, W,
                   npoints, kernel_size_H, kernel_size_W, K, flag_copy, distance, stride_h, stride_w,
                   want_valid, dense)


def fused_conv_select_k(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W, K,
                        flag_copy, distance, stride_h, stride_w, want_valid=True, dense=None):
    """K nearest in-range neighbours of the window (reference tie order).
    dense (extension, as for random-k): the LDS-tiled kernel for "every pixel a centre" with K <= 7 (the refinement cost
    volumes' call); None = automatically for registered index tensors.  Same outputs bit for bit."""
    return _launch("elo_fused_conv_select_k", "FusedConvSelectK", xyz1, xyz2, idx_n2, random_hw, H, W,
                   npoints, kernel_size_H, kernel_size_W, K, flag_copy, distance, stride_h, stride_w,
                   want_valid, dense)
