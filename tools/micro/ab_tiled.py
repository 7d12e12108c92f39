"""NEEDS tools/micro/patches/r06_rejected_forms.patch applied to the tree (setconv_tiled_kernel was taken out of csrc/ in round 6).
A/B of the narrow set-conv forms (round 5): setconv_small_kernel (half a wave per centre walks its window from L2), setconv_narrow_kernel
(the same walk, MLP on the matrix cores) and setconv_tiled_kernel (window union of 32 centres in LDS, eight lanes per centre) on the two
narrow layers of the pyramid at their batch-8 sizes (the Siamese pyramid runs both frames as one batch of 16 images):
    python tools/ab_tiled.py [--batch 16]"""
import argparse, importlib, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda sub=None: importlib.import_module("efficientlo-net_amd" + ("." + sub if sub else ""))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda:0")
fused, tf_util, pu, mu, synth, tuning = pkg("fused"), pkg("tf_util"), pkg("pointnet_util"), pkg("model_util"), pkg("synth"), pkg("tuning")
B = a.batch
for name, (H, W, C, cs, win, dist, mlp) in {"layer0 6->8->8->16": (64, 1800, 3, (4, 8), (9, 15), 0.5, [8, 8, 16]),
                                             "layer1 19->16->16->32": (16, 225, 16, (2, 2), (7, 11), 3.0, [16, 16, 32])}.items():
    f1, _ = synth.frame_pair(B, H, W, seed=5)
    xyz = torch.from_numpy(f1).to(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    for storage in ("f16", "f32"):
        feat = torch.zeros((B, H, W, C), device=dev) if C == 3 else torch.randn((B, H, W, C), generator=g).to(dev)
        feat = feat.half() if storage == "f16" else feat
        oh, ow = -(-H // cs[0]), -(-W // cs[1])
        sel = mu.get_selected_idx(xyz, cs[0], cs[1], oh, ow)
        chw = pu._centre_hw(sel)
        order = torch.randperm(win[0] * win[1], generator=g).to(torch.int32).to(dev)
        store = tf_util.VariableStore(dev, seed=0)
        widths = [3 + C] + mlp
        with tf_util.default_store(store), torch.no_grad():
            layers = [fused.packed_layer("ab%d" % i, widths[i], widths[i + 1], row_order=fused.setconv_row_order(C) if i == 0 else None) for i in range(3)]
        grp = fused.Grouping(order, list(win), dist)
        run = lambda: fused.setconv(xyz, feat, None, None, layers, xyz1_grid=xyz, centre_hw=chw, K=32, group=grp)
        row = {"layer": name, "images": B, "centres": B * oh * ow, "features": storage}
        for form, fields in (("small (VALU, L2 walk)", dict(tiled_setconv=0, narrow_mfma=0)), ("narrow (MFMA, L2 walk)", dict(tiled_setconv=0, narrow_mfma=2)),
                             ("tiled (VALU, LDS walk)", dict(tiled_setconv=2, narrow_mfma=0))):
            with tuning.override(**fields):
                row[form] = round(bench._time_launches(run, dev, a.reps) * 1e6, 2)
        print(json.dumps(row), flush=True)
