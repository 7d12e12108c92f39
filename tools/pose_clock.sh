#!/bin/bash
# Where pose_head_kernel -- 1 to 15 workgroups, 8-12 us: the smallest launch of a forward and at 8 lanes the largest single consumer of
# the timeline -- spends its time: a -DELO_POSE_CLOCK build stamps the shader clock at the phase boundaries of thread 0 of workgroup
# (0, 0), per pose head of a forward (l3, l2, l1, l0).  One 64 x 1800 pair, batch 1, eager forwards (the last one is read).
#   usage (GPU box, repo root):  bash tools/pose_clock.sh
set -eu
REPO=$(pwd); mkdir -p /tmp/elo_pose_clock
for f in efficientlo-net_amd/csrc/*.hip efficientlo-net_amd/csrc/*.cpp; do
    o=/tmp/elo_pose_clock/$(basename $f).o
    if [ "$(basename $f)" = elo_features.hip ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DELO_POSE_CLOCK -c $f -o $o
    else cp efficientlo-net_amd/build/$(basename $f).hip.o $o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/elo_pose_clock/*.o -o /tmp/libelo_pose_clock.so
ELO_LIB_PATH=/tmp/libelo_pose_clock.so python - <<'PY'
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
model, synth = pkg("model"), pkg("synth")
dev = "cuda:0"
net = model.PWCLONet(dev, seed=0)
f1, f2 = synth.frame_pair(1, 64, 1800, seed=5)
both = torch.from_numpy(np.concatenate([f1, f2], 0)).to(dev)
for _ in range(4):
    net.forward(both[:1], both[1:]); torch.cuda.synchronize()
fn = ctypes.CDLL(os.environ["ELO_LIB_PATH"]).elo_debug_pose_clock
out = (ctypes.c_ulonglong * 33)(); fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(out) == 0
t = list(out); n = t[32]
names = ["slices merged (partials landed)", "softmax_valid vector in LDS", "hidden layer 64 -> 256", "heads 256 -> 4 + 3", "compose + store", "next orders (l0)", "warp + bin 256 points"]
print("pose_head_kernel, thread 0 of workgroup (0,0): s_memtime ticks per phase (shader-clock cycles, ~2.4 GHz: 1000 ticks ~ 0.42 us; as tools/cv1_clock.sh)" )
first = n % 4                                    # the slot of the oldest of the last four launches = l3 of the last forward
for lvl, slot in zip(("l3", "l2", "l1", "l0"), [(first + i) % 4 for i in range(4)]):
    s = t[slot * 8: slot * 8 + 8]
    row = ["%s: " % lvl]
    for i, nm in enumerate(names):
        if s[i + 1] >= s[i] and s[i + 1] - s[i] < 10**7: row.append("%s %d" % (nm, s[i + 1] - s[i]))
    last = max(x for x in s if x >= s[0] and x - s[0] < 10**7)
    print("  " + " | ".join(row) + " | total %d" % (last - s[0]))
PY
