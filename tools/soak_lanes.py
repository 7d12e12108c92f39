"""Soak: many different synthetic pairs through 8 concurrent graph lanes, every result compared with the eager
single-stream forward of the same pair (bit for bit).   python tools/soak_lanes.py [--pairs 400]"""
import argparse, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = lambda m: importlib.import_module("efficientlo-net_amd" + ("." + m if m else ""))
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=400)
ap.add_argument("--lanes", type=int, default=8)
args = ap.parse_args()
model, synth = pkg("model"), pkg("synth")
dev = torch.device("cuda")
net = model.PWCLONet(dev, seed=0)
pairs = []
for i in range(args.pairs):
    f1, f2 = synth.frame_pair(1, 64, 1800, seed=5000 + i)
    pairs.append(torch.cat([torch.from_numpy(f1), torch.from_numpy(f2)], 0).to(dev))
want = []
for p in pairs:
    out = net.forward(p[:1], p[1:])
    want.append(torch.cat([out[0].reshape(-1), out[1].reshape(-1)]).clone())        # l0 q, t
net.capture(1, 64, 1800, lanes=args.lanes)
bad = 0
for start in range(0, args.pairs, args.lanes):
    chunk = list(range(start, min(args.pairs, start + args.lanes)))
    for lane, i in enumerate(chunk):
        net.submit(lane, pairs[i])
    torch.cuda.synchronize()
    for lane, i in enumerate(chunk):
        o = net._lanes[lane]["out"]
        got = torch.cat([o[0].reshape(-1), o[1].reshape(-1)])
        if not torch.equal(got, want[i]):
            bad += 1
            print("pair", i, "lane", lane, "max diff", float((got - want[i]).abs().max()))
print("soak: %d pairs through %d lanes, %d mismatches against the eager forward" % (args.pairs, args.lanes, bad))
sys.exit(1 if bad else 0)
