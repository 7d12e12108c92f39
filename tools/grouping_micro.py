"""Time the two stand-alone grouping ops at the BASELINE shapes (HIP events, want_valid on/off)."""
import importlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
elo = importlib.import_module("efficientlo-net_amd"); synth = importlib.import_module("efficientlo-net_amd.synth")
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for (name, fn, H, W, centres, win, K, dist) in [
        ("config1 random_k 64x1800 all pixels 9x15 K=16 d=0.5", elo.fused_conv_random_k, 64, 1800, "all", (9, 15), 16, 0.5),
        ("random_k layer0: 3600 strided centres 9x15 K=32", elo.fused_conv_random_k, 64, 1800, "strided", (9, 15), 32, 0.5),
        ("select_k 64x1800 all pixels 5x15 K=32 d=1000", elo.fused_conv_select_k, 64, 1800, "all", (5, 15), 32, 1000.0),
        ("select_k cost volume l0: 16x225, 11x41, K=6", elo.fused_conv_select_k, 16, 225, "all", (11, 41), 6, 1000.0),
        ("select_k 64x1800 all pixels 5x35 K=32", elo.fused_conv_select_k, 64, 1800, "all", (5, 35), 32, 1000.0),
        ("select_k 64x1800 all pixels 7x25 K=6", elo.fused_conv_select_k, 64, 1800, "all", (7, 25), 6, 1000.0),
        ("select_k 64x1800 all pixels 11x41 K=6", elo.fused_conv_select_k, 64, 1800, "all", (11, 41), 6, 1000.0)]:
    f1, f2 = synth.frame_pair(1, H, W, seed=3)
    idx = synth.hw_index(1, H, W) if centres == "all" else synth.strided_index(1, 16, 225, 4, 8)
    perm = np.random.default_rng(0).permutation(win[0] * win[1]).astype(np.int32)
    a = [t(f1), t(f2), t(idx), t(perm)]
    for want in (True, False):
        run = lambda: fn(a[0], a[1], a[2], a[3], H, W, idx.shape[1], win[0], win[1], K, 0, dist, 1, 1, want_valid=want)
        us = bench._time_launches(run, dev, 100) * 1e6
        print(json.dumps({"case": name, "valid_outputs": want, "us": round(us, 1)}))
        if fn is elo.fused_conv_random_k and centres == "all":          # the LDS-tiled form (every pixel a centre)
            run = lambda: fn(a[0], a[1], a[2], a[3], H, W, idx.shape[1], win[0], win[1], K, 0, dist, 1, 1, want_valid=want, dense=True)
            us = bench._time_launches(run, dev, 100) * 1e6
            print(json.dumps({"case": name + " [dense: LDS-staged windows]", "valid_outputs": want, "us": round(us, 1)}))
