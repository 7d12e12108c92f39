#!/bin/bash
# Where one workgroup of setconv_rr_kernel (set-upconv stage 1 at l0, batch 8, fp16 features) spends its time: a
# -DELO_CV1_CLOCK build stamps s_memtime at the phase boundaries of waves 0 and 7 of workgroup 1700 during one eager forward.
#   usage (GPU box, repo root):  bash tools/setconv_rr_clock.sh
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DELO_CV1_CLOCK -DELO_RR_CLOCK_BLOCK=1700 efficientlo-net_amd/csrc/*.hip efficientlo-net_amd/csrc/*.cpp -o /tmp/libelo_clock.so
ELO_LIB_PATH=/tmp/libelo_clock.so python - <<'PY'
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
rec = bench.recorded_cost_volume(dev, 8, 64, 1800, True)     # runs one eager batch-8 forward (fp16 features)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 24)()
fn = ctypes.CDLL(os.environ["ELO_LIB_PATH"]).elo_debug_cv1_clock
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(out) == 0
t = list(out)
names = ["centres+order", "grouping", "sync", "gather", "barrier", "layer 1", "layer 2", "pool"]
for w, o in ((0, 0), (7, 12)):
    print("wave %d: " % w + " | ".join("%s %d" % (n, t[o + i + 1] - t[o + i]) for i, n in enumerate(names)) + " | total %d" % (t[o + 8] - t[o]))
PY
