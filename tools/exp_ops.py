"""Which aten ops run in one inference forward (eager), with their kernel counts."""
import sys, torch
sys.path.insert(0, '.')
from importlib import import_module
from torch.profiler import profile, ProfilerActivity
model = import_module('efficientlo-net_amd.model'); synth = import_module('efficientlo-net_amd.synth')
dev = torch.device('cuda:0')
net = model.PWCLONet(dev, seed=0)
f1, f2 = synth.frame_pair(1, 64, 1800, seed=1)
a, b = torch.from_numpy(f1.copy()).to(dev), torch.from_numpy(f2.copy()).to(dev)
for _ in range(3): net.forward(a, b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    net.forward(a, b); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:40]:
    print("%4d  cpu %8.1fus  dev %8.1fus  %s" % (e.count, e.cpu_time_total, e.device_time_total, e.key[:90]))
