"""What a plain fill / copy reaches on this GPU (the practical ceiling the gather kernels are compared with)."""
import torch
dev = torch.device("cuda:0")
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
for mb in (32, 256, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    print("%5d MB  fill %.0f GB/s   copy (r+w) %.0f GB/s   read-sum %.0f GB/s" % (
        mb, n * 4 / t(lambda: x.fill_(1.0)) / 1e9, 2 * n * 4 / t(lambda: y.copy_(x)) / 1e9, n * 4 / t(lambda: x.sum()) / 1e9))
