"""ctypes front-end of oracle/libelo_oracle.so (own C restatement) and, when it
has been built, oracle/_ref/libelo_ref.so (the reference's kernel bodies built
for the host by oracle/build_ref.sh).  TEST INFRASTRUCTURE ONLY.

Signature mirrors the reference wrappers
(tf_ops/2d_conv_random_k/fused_conv_random_k.py:14-29,
 tf_ops/2d_conv_select_k/fused_conv_select_k.py:14-29) on numpy arrays.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "libelo_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libelo_ref.so")
_REF_FMA_SO = os.path.join(_HERE, "_ref", "libelo_ref_fma.so")     # the same bodies built with -ffp-contract=fast -mfma

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_COMMON = [ctypes.c_int] * 8 + [ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                _f32p, _f32p, _i32p, _i32p, _i32p, _f32p, _f32p, _f32p,
                                ctypes.c_int, ctypes.c_int]


def build(force=False):
    """Compile libelo_oracle.so (and _ref/ when /root/reference exists)."""
    src = os.path.join(_HERE, "elo_oracle.c")
    if force or not os.path.exists(_ORACLE_SO) or os.path.getmtime(_ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libelo_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir(os.environ.get("ELO_REFERENCE_DIR", "/root/reference")):
        if force or not os.path.exists(_REF_SO) or not os.path.exists(_REF_FMA_SO):
            subprocess.check_call(["bash", os.path.join(_HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)


_libs = {}


def _lib(which):
    if which not in _libs:
        if which == "oracle":
            if not os.path.exists(_ORACLE_SO):
                build()
            lib = ctypes.CDLL(_ORACLE_SO)
            for op in ("random", "select"):
                fn = getattr(lib, "elo_oracle_fused_conv_%s_k" % op)
                fn.argtypes = _COMMON + [ctypes.c_int]
                fn.restype = ctypes.c_int
        else:
            path = _REF_FMA_SO if which == "ref_fma" else _REF_SO
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            lib = ctypes.CDLL(path)
            for op in ("random", "select"):
                fn = getattr(lib, "elo_ref_fused_conv_%s_k" % op)
                fn.argtypes = _COMMON
                fn.restype = ctypes.c_int
        _libs[which] = lib
    return _libs[which]


def have_ref():
    return os.path.exists(_REF_SO)


def have_ref_fma():
    return os.path.exists(_REF_FMA_SO)


def _run(op, impl, xyz1, xyz2, idx_n2, random_hw, npoints, kH, kW, K, flag_copy, distance,
         stride_h, stride_w, threads):
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
    idx_n2 = np.ascontiguousarray(idx_n2, dtype=np.int32)
    random_hw = np.ascontiguousarray(random_hw, dtype=np.int32)
    B, H, W, _ = xyz1.shape
    H2 = math.ceil(H / float(stride_h))        # fused_conv.cpp:112-113
    W2 = math.ceil(W / float(stride_w))
    assert xyz2.shape[0] == B and xyz2.shape[1] == H2 and xyz2.shape[2] == W2 and xyz2.shape[3] == 3
    assert idx_n2.shape == (B, npoints, 2) and random_hw.shape == (kH * kW,)
    KT = kH * kW
    sel = np.empty((B, npoints, K, 3), np.int32)
    valid = np.empty((B, npoints, KT, 1), np.float32)
    indis = np.empty((B, npoints, KT, 1), np.float32)
    mask = np.empty((B, npoints, K, 1), np.float32)
    p = lambda a, t: a.ctypes.data_as(t)
    args = [B, H, W, npoints, kH, kW, K, flag_copy, float(distance), stride_h, stride_w,
            p(xyz1, _f32p), p(xyz2, _f32p), p(idx_n2, _i32p), p(random_hw, _i32p),
            p(sel, _i32p), p(valid, _f32p), p(indis, _f32p), p(mask, _f32p), H2, W2]
    if impl == "oracle":
        rc = getattr(_lib("oracle"), "elo_oracle_fused_conv_%s_k" % op)(*args, int(threads))
    else:
        rc = getattr(_lib(impl), "elo_ref_fused_conv_%s_k" % op)(*args)
    if rc != 0:
        raise ValueError("oracle rejected the arguments (rc=%d)" % rc)
    return sel, valid, indis, mask


def _threads(threads):
    """threads=None: ELO_ORACLE_THREADS (the threaded CPU-baseline leg of bench.py sets it), else 1."""
    return int(os.environ.get("ELO_ORACLE_THREADS", "1")) if threads is None else threads


def fused_conv_random_k(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W,
                        K, flag_copy, distance, stride_h, stride_w, impl="oracle", threads=None):
    threads = _threads(threads)
    return _run("random", impl, xyz1, xyz2, idx_n2, random_hw, npoints, kernel_size_H,
                kernel_size_W, K, flag_copy, distance, stride_h, stride_w, threads)


def fused_conv_select_k(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W,
                        K, flag_copy, distance, stride_h, stride_w, impl="oracle", threads=None):
    threads = _threads(threads)
    return _run("select", impl, xyz1, xyz2, idx_n2, random_hw, npoints, kernel_size_H,
                kernel_size_W, K, flag_copy, distance, stride_h, stride_w, threads)
