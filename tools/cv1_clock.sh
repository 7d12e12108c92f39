#!/bin/bash
# Where one cv1_kernel tile spends its time: a -DELO_CV1_CLOCK build stamps s_memtime at the phase boundaries of wave 0 of
# workgroup 0 (grouping | gather + encode | CV_0 | CV_1 | CV_2 | CV_xyz | sum_CV_0 | sum_CV_1 | pooling).
#   usage (GPU box, repo root):  bash tools/cv1_clock.sh [batch]
set -eu
B=${1:-1}; REPO=$(pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DELO_CV1_CLOCK \
    efficientlo-net_amd/csrc/*.hip efficientlo-net_amd/csrc/*.cpp -o /tmp/libelo_clock.so
ELO_LIB_PATH=/tmp/libelo_clock.so python - "$B" <<'PY'
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.getcwd())
sys.argv = ["roofline_micro", "--kernel", "cv1", "--batch", sys.argv[1], "--reps", "5"]
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import runpy
runpy.run_path("tools/roofline_micro.py", run_name="__main__")
L = importlib.import_module("efficientlo-net_amd._lib")
out = (ctypes.c_ulonglong * 24)()
fn = ctypes.CDLL(os.environ["ELO_LIB_PATH"]).elo_debug_cv1_clock
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(out) == 0
names = ["grouping", "gather+encode", "CV_0", "CV_1", "CV_2", "CV_xyz", "sum_CV_0", "sum_CV_1", "pooling"]
t = list(out)
print("s_memtime ticks (shader-clock cycles, ~2.4 GHz) per phase, wave 0 of workgroup 0:")
for i, n in enumerate(names):
    print("  %-14s %6d ticks" % (n, t[i + 1] - t[i]))
print("  total          %6d ticks" % (t[9] - t[0]))
print("inside the grouping (the LAST centre wave 0 handled): offsets staged +%d | its probes %d | selection %d | "
      "first centre done at +%d, second at +%d" % (t[10] - t[0], t[13] - t[12], t[15] - t[13], t[16] - t[0], t[17] - t[0]))
print("  last centre: loop top +%d, probes start +%d, probes end +%d, selection end +%d, centre done +%d" % (
    t[11] - t[0], t[12] - t[0], t[13] - t[0], t[15] - t[0], t[17] - t[0]))
PY
