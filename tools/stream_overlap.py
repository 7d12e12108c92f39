"""Which torch streams run concurrently?  Spin kernels on pairs / groups of streams."""
import time, torch
dev = torch.device("cuda:0")
N = 12
streams = [torch.cuda.Stream(device=dev) for _ in range(N)]
CYC = 2_000_000
def run(ids):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in ids:
        with torch.cuda.stream(streams[i]):
            torch.cuda._sleep(CYC)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
run([0]); base = run([0])
print("one spin: %.2f ms" % base)
print("ptrs", [hex(s.cuda_stream) for s in streams])
for i in range(N):
    print(i, " ".join("%.1f" % (run([i, j]) / base) for j in range(N)))
for k in (2, 3, 4, 6, 8, 12):
    print("first %d streams together: %.2fx of one" % (k, run(list(range(k))) / base))
