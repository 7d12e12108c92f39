#!/bin/bash
# Where one workgroup of cv1_rr_kernel spends its time: -DELO_CV1_CLOCK builds stamp s_memtime at the phase boundaries of
# waves 0 and 7 of workgroup ELO_RR_CLOCK_BLOCK (gather | barrier | CV_0 | CV_1 | CV_2 | CV_xyz | sum_CV_0 | sum_CV_1 | pooling).
#   usage (GPU box, repo root):  bash tools/rr_clock.sh [batch] [block ...]
set -eu
B=${1:-8}; shift || true
BLOCKS=${*:-0 700}
for blk in $BLOCKS; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DELO_CV1_CLOCK -DELO_RR_CLOCK_BLOCK=$blk ${ELO_RR_FLAGS:-} \
    efficientlo-net_amd/csrc/*.hip efficientlo-net_amd/csrc/*.cpp -o /tmp/libelo_clock.so
ELO_LIB_PATH=/tmp/libelo_clock.so python - "$B" "$blk" <<'PY'
import ctypes, importlib, os, sys, runpy
sys.path.insert(0, os.getcwd())
blk = sys.argv[2]
sys.argv = ["roofline_micro", "--kernel", "cv1", "--batch", sys.argv[1], "--reps", "5", "--pregrouped"]
runpy.run_path("tools/roofline_micro.py", run_name="__main__")
out = (ctypes.c_ulonglong * 24)()
fn = ctypes.CDLL(os.environ["ELO_LIB_PATH"]).elo_debug_cv1_clock
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(out) == 0
names = ["gather", "barrier", "CV_0", "CV_1", "CV_2", "CV_xyz", "sum_CV_0", "sum_CV_1", "pooling"]
if "ELO_RR_CLOCK_POOL" in os.environ.get("ELO_RR_FLAGS", ""):
    names = ["gather", "barrier + all layers + mask", "half 0: write", "barrier", "softmax + store", "half 1: barrier + write", "barrier", "softmax", "store"]
t = list(out)
for w, o in ((0, 0), (7, 12)):
    print("workgroup %s wave %d, s_memtime ticks (100 MHz x ?): " % (blk, w) + " | ".join("%s %d" % (n, t[o + i + 1] - t[o + i]) for i, n in enumerate(names)) +
          " | total %d" % (t[o + 9] - t[o]))
PY
done
