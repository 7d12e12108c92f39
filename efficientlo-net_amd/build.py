"""In-tree build of libelo_hip.so (hipcc, gfx950 only).  No torch headers are
involved: the library is a plain C ABI (include/elo.h) loaded with ctypes.

    python efficientlo-net_amd/build.py [--force]
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libelo_hip.so")
LIB_F32 = os.path.join(PKG, "libelo_hip_f32.so")      # the same sources with -DELO_DENSE_F32: true-fp32 MFMA, the comparison build
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off",          # arithmetic contract: no FMA contraction
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(PKG, "..", "include", "elo.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """libelo_hip.so (the product) and libelo_hip_f32.so (ELO_DENSE_F32=1 selects it: every 1x1 convolution on
    v_mfma_f32_16x16x4_f32, kept to bench and test the fp16-split products against)."""
    for lib, extra in ((LIB, []), (LIB_F32, ["-DELO_DENSE_F32"])):
        if not force and not stale(lib):
            continue
        cmd = [HIPCC] + FLAGS + extra + sources() + ["-o", lib]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
