"""Throughput of the lane scheme when the lane streams are chosen by hardware queue.
python tools/lane_queues.py"""
import importlib, sys, time
import torch
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
model, synth = pkg("model"), pkg("synth")
dev = torch.device("cuda:0")
pool = [torch.cuda.Stream(device=dev) for _ in range(32)]
CYC = 4_000_000
def spin(ids):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in ids:
        with torch.cuda.stream(pool[i]): torch.cuda._sleep(CYC)
    torch.cuda.synchronize(); return time.perf_counter() - t0
spin(list(range(len(pool)))); spin([0]); base = min(spin([0]) for _ in range(3))
classes = []
for i in range(len(pool)):
    for c in classes:
        if spin([c[0], i]) > 1.5 * base:
            c.append(i); break
    else:
        classes.append([i])
print("queue classes:", classes)
f1, f2 = synth.frame_pair(1, 64, 1800, seed=1)
pair = torch.cat([torch.from_numpy(f1), torch.from_numpy(f2)], 0).to(dev)
def bench(stream_ids, steps=600):
    net = model.PWCLONet(dev, seed=0)
    net.capture(1, 64, 1800, lanes=len(stream_ids))
    for lane, sid in zip(net._lanes, stream_ids): lane["stream"] = pool[sid]
    n = len(stream_ids)
    for i in range(2 * n): net.submit(i % n, pair)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): net.submit(i % n, pair)
    torch.cuda.synchronize(); return steps / (time.perf_counter() - t0)
nq = len(classes)
for per in (1, 2, 3, 4):
    ids = [c[k] for k in range(per) for c in classes if k < len(c)]     # round-robin over queues
    print("%d lanes (%d per queue, %d queues): %.0f pairs/s" % (len(ids), per, nq, bench(ids)))
for q in range(1, nq + 1):
    ids = [c[0] for c in classes[:q]]; print("%d lanes on %d queues: %.0f" % (q, q, bench(ids)))
ids = classes[0][:3]; print("3 lanes on 1 queue: %.0f" % bench(ids))
ids = classes[0][:2] + classes[1][:1] + classes[2][:1] + classes[3][:1]; print("5 lanes on 4 queues: %.0f" % bench(ids))
