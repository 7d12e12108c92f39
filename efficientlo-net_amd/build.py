"""In-tree build of libelo_hip.so (hipcc, gfx950 only).  No torch headers are
involved: the library is a plain C ABI (include/elo.h) loaded with ctypes.

    python efficientlo-net_amd/build.py [--force]

Every source is compiled to its own object (in parallel, only when it or a header changed) and the objects are linked:
touching one kernel file rebuilds one object per library.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")                        # objects (git-ignored, not shipped)
LIB = os.path.join(PKG, "libelo_hip.so")
LIB_F32 = os.path.join(PKG, "libelo_hip_f32.so")      # the same sources with -DELO_DENSE_F32: true-fp32 MFMA, the comparison build
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-ffp-contract=off",          # arithmetic contract: no FMA contraction
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(PKG, "..", "include", "elo.h")]


def _newer(deps, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def stale(lib=LIB):
    return _newer(sources() + headers(), lib)


def _compile(job):
    src, obj, extra, verbose = job
    cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=False, extra_flags=(), only=None):
    """libelo_hip.so (the product) and libelo_hip_f32.so (ELO_DENSE_F32=1 selects it: every 1x1 convolution on
    v_mfma_f32_16x16x4_f32, kept to bench and test the fp16-split products against).  `only`: "hip" / "f32"."""
    os.makedirs(OBJ, exist_ok=True)
    variants = [("hip", LIB, []), ("f32", LIB_F32, ["-DELO_DENSE_F32"])]
    jobs, links = [], []
    for tag, lib, extra in variants:
        if only and tag != only:
            continue
        extra = extra + list(extra_flags)
        objs = []
        for src in sources():
            obj = os.path.join(OBJ, "%s.%s.o" % (os.path.basename(src), tag))
            objs.append(obj)
            if force or _newer([src] + headers(), obj):
                jobs.append((src, obj, extra, verbose))
        links.append((lib, objs))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(_compile, jobs))
    linked = 0
    global LAST_BUILD
    for lib, objs in links:
        linked += 1 if (force or _newer(objs, lib)) else 0
        if force or _newer(objs, lib):
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    # what this call did (the driver's "does it build" check reuses objects that are newer than their sources: say so)
    LAST_BUILD = {"build_mode": "compiled" if jobs else "reused", "objects_compiled": len(jobs), "objects_total": sum(len(o) for _l, o in links),
                  "libraries_linked": linked, "forced": bool(force)}
    if verbose:
        print("[build] %s" % LAST_BUILD, flush=True)
    return LIB


LAST_BUILD = None

if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
