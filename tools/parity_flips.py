"""Free-running parity statistic (tests/test_parity_flips_gpu.py) as a table: per configuration the share of (pair, level)
outputs within 1e-4 of the free-running oracle, the misses and the discrete flips that explain them.
    python tools/parity_flips.py > gpurun_out/parity_flips.txt"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_flips_gpu as T  # noqa: E402

for B, seeds, features, H, W in ((1, range(200, 232), "f32", 64, 1800), (8, (300, 301, 302, 303), "f32", 64, 1800), (8, (310, 311), "f16", 64, 1800),
                                 (8, (320, 322), "f32", 128, 2048), (8, (321,), "f16", 128, 2048)):
    rows = T.statistic(B, list(seeds), features, H, W)
    print(json.dumps({"batch": B, "features": features, "grid": "%dx%d" % (H, W), "seeds": len(list(seeds)), **T.summarise(rows)}))
    for lvl in T.LEVELS:
        sub = [r for r in rows if r["level"] == lvl]
        print("   l%d: %d outputs, %d within 1e-4 free-running, worst forced %.2e, worst free %.2e" % (
            lvl, len(sub), sum(r["free"] <= T.TOL for r in sub), max(r["forced"] for r in sub), max(r["free"] for r in sub)))
