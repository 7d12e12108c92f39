"""Host-side cost of one lane submit (copy + hipGraphLaunch + copy) vs the GPU time it buys.
python tools/submit_cost.py [lanes] [threads]"""
import importlib, sys, threading, time
import torch
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
model, synth = pkg("model"), pkg("synth")
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 12
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
net = model.PWCLONet(dev, seed=0)
f1, f2 = synth.frame_pair(1, 64, 1800, seed=1)
pair = torch.cat([torch.from_numpy(f1), torch.from_numpy(f2)], 0).to(dev)
net.capture(1, 64, 1800, lanes=lanes)
log = torch.empty((4096, 1, 7), device=dev)
def work(ids):
    for i in ids:
        lane = i % lanes
        net.submit(lane, pair)
        with torch.cuda.stream(net.lane_stream(lane)):
            log[i].copy_(net.lane_pose(lane), non_blocking=True)
for steps in (240, 1200):
    work(range(48)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if threads == 1:
        work(range(steps))
    else:   # thread t owns lanes t, t+threads, ...
        ts = [threading.Thread(target=work, args=([i for i in range(steps) if (i % lanes) % threads == t],)) for t in range(threads)]
        [t.start() for t in ts]; [t.join() for t in ts]
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("lanes %d threads %d steps %d: host submit %.1f us/step, total %.1f us/step" % (lanes, threads, steps, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
