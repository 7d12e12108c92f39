"""Per-level pose error of the fp16-feature-storage forward (batch 8) against the oracle in the same storage mode,
beside the distance between the fp16-storage and fp32 oracles: the data behind FP16_STORAGE_TOL in tests/test_model_gpu.py.

    python tools/fp16_drift.py [seeds...]
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ops_np as O                      # noqa: E402
from util_params import export, randomise, shuffle_fn   # noqa: E402

pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
model, perm, synth = pkg("model"), pkg("perm"), pkg("synth")
DEV = "cuda:0"
B, H, W = 8, 64, 1800
for seed in [int(a) for a in sys.argv[1:]] or [52]:
    f1, f2 = synth.frame_pair(B, H, W, seed=seed)
    both = torch.from_numpy(np.concatenate([f1, f2], 0)).to(DEV)
    res = {}
    for name, dt in (("f16", torch.float16), ("f32", torch.float32)):
        net = model.PWCLONet(DEV, seed=5, perm_source=perm.PermSource(fn=shuffle_fn), feature_dtype=dt)
        net.forward(both[:B], both[B:])
        randomise(net.store, seed=7)
        res[name] = [g.cpu().numpy() for g in net.forward(both[:B], both[B:])]
        params = export(net.store)
    with O.feature_storage(np.float16):
        o16 = O.get_model_from_projection(params, shuffle_fn, f1, f2)
        r = res["f16"]
        o16f = O.get_model_from_projection(params, shuffle_fn, f1, f2, coarse_pose={3: (r[6], r[7]), 2: (r[4], r[5]), 1: (r[2], r[3])})
    o32 = O.get_model_from_projection(params, shuffle_fn, f1, f2)
    r = res["f32"]
    o32f = O.get_model_from_projection(params, shuffle_fn, f1, f2, coarse_pose={3: (r[6], r[7]), 2: (r[4], r[5]), 1: (r[2], r[3])})
    names = ["l0_q", "l0_t", "l1_q", "l1_t", "l2_q", "l2_t", "l3_q", "l3_t"]
    print("seed", seed)
    for i, n in enumerate(names):
        e16 = np.abs(res["f16"][i] - o16[i]).max(-1)
        e32 = np.abs(res["f32"][i] - o32[i]).max(-1)
        d = np.abs(o16[i] - o32[i]).max(-1)
        ef16 = np.abs(res["f16"][i] - o16f[i]).max(-1)
        ef32 = np.abs(res["f32"][i] - o32f[i]).max(-1)
        print("%-5s FORCED f16 max %.2e med %.2e | f32 max %.2e med %.2e | f16 per element %s" % (n, ef16.max(), np.median(ef16), ef32.max(), np.median(ef32), " ".join("%.1e" % v for v in ef16)))
        print("%-5s gpu16-vs-oracle16 max %.2e med %.2e | gpu32-vs-oracle32 max %.2e | oracle16-vs-oracle32 max %.2e med %.2e | per element %s"
              % (n, e16.max(), np.median(e16), e32.max(), d.max(), np.median(d), " ".join("%.1e" % v for v in e16)))
