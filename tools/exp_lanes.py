"""Experiment: is multi-lane replay limited by the CPU (hipGraphLaunch) or by the GPU?"""
import sys, time, threading, torch
sys.path.insert(0, '.')
from importlib import import_module
model = import_module('efficientlo-net_amd.model'); synth = import_module('efficientlo-net_amd.synth')
dev = torch.device('cuda:0')
net = model.PWCLONet(dev, seed=0)
f1, f2 = synth.frame_pair(1, 64, 1800, seed=1)
a, b = torch.from_numpy(f1.copy()).to(dev), torch.from_numpy(f2.copy()).to(dev)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net.capture(1, 64, 1800, lanes=L)
def run_serial(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): net.submit(i % L, a, b)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return n / (t2 - t0), (t1 - t0) / n * 1e3
def run_threads(n, T):
    def work(tid):
        for i in range(tid, n, T): net.submit(i % L, a, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); return n / (time.perf_counter() - t0)
run_serial(50)
r, cpu_ms = run_serial(400)
print("lanes", L, "serial pairs/s %.0f  cpu ms per submit %.3f" % (r, cpu_ms))
for T in (2, 4, 8):
    if T <= L: print("  threads", T, "pairs/s %.0f" % run_threads(400, T))
