"""Host cost of one lane submit, piece by piece: torch's copy_ + CUDAGraph.replay() against hipMemcpyAsync + hipGraphLaunch
called straight from ctypes on the graph's raw exec handle.   python tools/submit_native_probe.py"""
import ctypes, importlib, sys, time
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
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGraphLaunch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
L = net._lanes
try:
    execs = [l["graph"].raw_cuda_graph_exec() for l in L]
    print("exec handles:", [hex(e) for e in execs[:2]])
except Exception as e:
    print("raw_cuda_graph_exec failed:", repr(e)); execs = None
streams = [l["stream"].cuda_stream for l in L]
nbytes = pair.numel() * 4
def timed(name, fn, n=20, reps=3):
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n): fn(i)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("%-34s host %.1f us per step, 20 steps with sync %.0f us" % (name, (t1 - t0) / n * 1e6, (t2 - t0) * 1e6))
timed("net.submit(lane, pair)", lambda i: net.submit(i % lanes, pair))
timed("net.submit(lane) (no copy)", lambda i: net.submit(i % lanes))
def torch_copy(i):
    with torch.cuda.stream(L[i % lanes]["stream"]): L[i % lanes]["pair"].copy_(pair, non_blocking=True)
timed("torch copy_ only", torch_copy)
def torch_replay(i):
    with torch.cuda.stream(L[i % lanes]["stream"]): L[i % lanes]["graph"].replay()
timed("torch replay only", torch_replay)
if execs:
    def raw(i):
        k = i % lanes
        hip.hipMemcpyAsync(L[k]["pair"].data_ptr(), pair.data_ptr(), nbytes, 3, streams[k])
        hip.hipGraphLaunch(execs[k], streams[k])
    timed("ctypes memcpyAsync + graphLaunch", raw)
    timed("ctypes graphLaunch only", lambda i: hip.hipGraphLaunch(execs[i % lanes], streams[i % lanes]))
