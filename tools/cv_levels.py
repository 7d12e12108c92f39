"""The eight fused cost-volume launches of a forward at batch B (bench.py's cost_volume leg), fp32 and fp16 features.
    python tools/cv_levels.py [B]"""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
for half in (False, True):
    r = bench.cost_volume_leg(dev, B, 64, 1800, half)
    print("batch %d %s: %.1f us, %.1f GB/s (%.3f of HBM), mfma %.3f of fp16 peak | %s" % (
        B, r["features"], r["us"], r["achieved"], r["frac"], r["mfma"]["frac"],
        " ".join("%s %.1f+%.1f" % (k, v["cv1_us"], v["cv2_us"]) for k, v in r["levels"].items())))
