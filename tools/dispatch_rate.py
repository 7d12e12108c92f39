"""How many dependent tiny kernels per second does the GPU retire, as a function of the number of
streams replaying graphs of them?  python tools/dispatch_rate.py"""
import time, torch
dev = torch.device("cuda:0")
NODES = 50
def build(lanes):
    out = []
    for _ in range(lanes):
        x = torch.zeros(256, device=dev)
        s = torch.cuda.Stream(device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            for _ in range(3): x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(NODES): x.add_(1.0)
        out.append((s, g, x))
    return out
for lanes in (1, 2, 3, 4, 6, 8, 12, 16):
    L = build(lanes)
    reps = 200
    for i in range(2 * lanes):
        s, g, _ = L[i % lanes]
        with torch.cuda.stream(s): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        s, g, _ = L[i % lanes]
        with torch.cuda.stream(s): g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("lanes %2d: %.2f us per kernel (host %.2f)" % (lanes, (t2 - t0) / (reps * NODES) * 1e6, (t1 - t0) / (reps * NODES) * 1e6))
