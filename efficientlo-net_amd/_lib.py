"""ctypes binding of include/elo.h.  There is NO fallback: if libelo_hip.so is
missing or a call fails, the operators raise."""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libelo_hip.so")

_vp = ctypes.c_void_p


class GroupArgs(ctypes.Structure):
    """struct elo_group_args (include/elo.h)."""
    _fields_ = [("batch", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                ("H2", ctypes.c_int), ("W2", ctypes.c_int), ("npoints", ctypes.c_int),
                ("kernel_h", ctypes.c_int), ("kernel_w", ctypes.c_int), ("K", ctypes.c_int),
                ("flag_copy", ctypes.c_int), ("distance", ctypes.c_float),
                ("stride_h", ctypes.c_int), ("stride_w", ctypes.c_int),
                ("xyz1", _vp), ("xyz2", _vp), ("idx_n2", _vp), ("random_hw", _vp),
                ("selected_bhw_idx", _vp), ("valid_idx", _vp), ("valid_in_dis_idx", _vp),
                ("selected_mask", _vp)]


# every symbol include/elo.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("elo_abi_version", ctypes.c_int, []),
    ("elo_last_error", ctypes.c_char_p, []),
    ("elo_fused_conv_random_k", ctypes.c_int, [ctypes.POINTER(GroupArgs), _vp]),
    ("elo_fused_conv_select_k", ctypes.c_int, [ctypes.POINTER(GroupArgs), _vp]),
]

_lib = None


class EloError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EloError("libelo_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` -- there is no CPU fallback for the hot path" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)          # AttributeError if the symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise EloError("libelo_hip: %s (status %d)" % (lib().elo_last_error().decode(), rc))


def stream_ptr(t):
    """hipStream_t of torch's current stream on t's device, as void*."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise EloError("the EfficientLO-Net hot path runs on an AMD GPU only (got a %s tensor); "
                           "there is no CPU fallback" % t.device)
