import importlib, sys, time
import torch
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
model, synth = pkg("model"), pkg("synth")
dev = torch.device("cuda:0")
net = model.PWCLONet(dev, seed=0)
f1, f2 = synth.frame_pair(1, 64, 1800, seed=1)
pair = torch.cat([torch.from_numpy(f1), torch.from_numpy(f2)], 0).to(dev)
lanes = 8
net.capture(1, 64, 1800, lanes=lanes, pose_ring=64)
for i in range(32): net.submit(i % lanes, pair)
torch.cuda.synchronize()
for mode in ("copy+replay", "replay only"):
    for rep in range(3):
        torch.cuda.synchronize(); ts = []
        t0 = time.perf_counter()
        for i in range(20):
            a = time.perf_counter()
            net.submit(i % lanes, pair) if mode == "copy+replay" else net.submit(i % lanes)
            ts.append((time.perf_counter() - a) * 1e6)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(mode, "host per submit us:", " ".join("%.0f" % t for t in ts[:20]), "| host total %.0f us, with sync %.0f us" % ((t1 - t0) * 1e6, (t2 - t0) * 1e6))
