"""One eager training step at batch B under the torch profiler: GPU time per kernel (top 30) and the launch count.
    python tools/train_kernel_stats.py [B] [--by-count] [--sequence]   (--sequence: every launch of >= 12 us in launch order)"""
import importlib, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
model, training, synth = pkg("model"), pkg("training"), pkg("synth")
dev = "cuda:0"; B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
net = model.PWCLONet(dev, seed=0); tr = training.Trainer(net)
f1, f2 = synth.frame_pair(B, 64, 1800, seed=1)
a, b = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
q = torch.tensor([[0.99995, 0, 0, 0.01]] * B, device=dev); t = torch.tensor([[[0.8], [0.0], [0.0]]] * B, device=dev)
for _ in range(3): tr.step(a, b, q, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    tr.step(a, b, q, t); torch.cuda.synchronize()
if "--sequence" in sys.argv:
    evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
    t0 = evs[0].time_range.start
    for i, e in enumerate(evs):
        if e.device_time >= 12:
            print("%5d %9.1f us  %6.1f us  %s" % (i, e.time_range.start - t0, e.device_time, e.name[:110]))
agg = collections.defaultdict(lambda: [0, 0.0])
buckets = collections.OrderedDict((k, [0, 0.0]) for k in (6, 12, 25, 50, 100, 1e9))
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        agg[e.name[:90]][0] += 1; agg[e.name[:90]][1] += e.device_time
        for k in buckets:
            if e.device_time < k:
                buckets[k][0] += 1; buckets[k][1] += e.device_time
                break
tot = sum(v[1] for v in agg.values()); n = sum(v[0] for v in agg.values())
print("batch %d: %d launches, %.2f ms of GPU time" % (B, n, tot / 1e3))
print("by duration: " + "  ".join("<%s us: %d launches %.2f ms" % ("%g" % k if k < 1e9 else "inf", v[0], v[1] / 1e3) for k, v in buckets.items()))
by_count = "--by-count" in sys.argv
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0 if by_count else 1])[:200 if by_count else 30]:
    print("%6.2f ms %5.1f%% x%-5d %s" % (v[1] / 1e3, 100 * v[1] / tot, v[0], k))
