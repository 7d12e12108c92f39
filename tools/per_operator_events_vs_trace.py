"""The four per-operator cost-volume kernels at l0, batch 8, fp16 storage: HIP-event time per launch (bench.py's
_time_launches at 20 and at 200 launches per graph) beside the kernel-trace durations of the same process:
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o t -- python <repo>/tools/per_operator_events_vs_trace.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import torch
dev = torch.device("cuda:0")
for reps in (20, 200):
    r = bench.per_operator_leg(dev, 8, True, reps=reps)
    print("reps %3d: %.2f us  frac %.4f  " % (reps, r["us"], r["frac"]) + "  ".join("%s %.2f" % (k, v["us"]) for k, v in r["terms"].items()))
