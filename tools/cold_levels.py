"""The 16 per-operator cost-volume launches of a forward (A1, P1, A2, P2 at l0, l1, l2, l2_origin), HBM-cold and warm, per term:
bench.per_operator_all_levels_leg on the 128 x 2048 level shapes (BASELINE configs[4]) or the 64 x 1800 ones.
    python tools/cold_levels.py [--half] [--grid64] [--batch 8]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--half", action="store_true"); ap.add_argument("--grid64", action="store_true")
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
table = bench.PER_OPERATOR_LEVELS if a.grid64 else bench.PER_OPERATOR_LEVELS_HIRES
tot_b = tot_c = tot_w = 0.0
for lv in table:
    r = bench.per_operator_leg(dev, a.batch, a.half, level=lv, table=table)
    print("%-10s %7.1f MB  cold %7.2f us (%.3f)  warm %7.2f us (%.3f)   " % (lv, r["bytes"] / 1e6, r["us"], r["frac"], r["us_warm"], r["frac_warm"]) +
          "  ".join("%s %.2f/%.2f us (%.3f)" % (t, v["us"], v["us_warm"], v["frac"]) for t, v in r["terms"].items()), flush=True)
    tot_b += r["bytes"]; tot_c += r["us"]; tot_w += r["us_warm"]
print("all levels %7.1f MB  cold %7.2f us (%.4f of 8 TB/s)  warm %7.2f us (%.4f);  60 %% = %.1f us" % (
    tot_b / 1e6, tot_c, tot_b / tot_c / 8e6, tot_w, tot_b / tot_w / 8e6, tot_b / 4.8e6))
