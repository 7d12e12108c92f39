"""elo_dense_weight_grad on the products of a training step (batch 8, 64 x 1800): time per call inside one captured graph of 20,
the library's x^T dz beside it, and what the operands' bytes would take at 8 TB/s / the MFMAs at the fp32 matrix peak.
    python tools/wgrad_micro.py"""
import importlib, sys
import torch
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
ops, L = pkg("_ops"), pkg("_lib")
dev = "cuda:0"
SHAPES = [(921600, 6, 8), (921600, 8, 8), (921600, 8, 16), (231424, 19, 16), (231424, 16, 16), (231424, 16, 32), (29184, 32, 64),
          (58368, 138, 128), (58368, 128, 64), (172800, 42, 128), (172800, 128, 64), (172800, 64, 64), (172800, 10, 64),
          (172800, 128, 128), (115200, 144, 128), (115200, 128, 64), (230400, 67, 128), (230400, 128, 64), (28800, 144, 128), (7296, 192, 128)]
def graph_time(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    for _ in range(2): g.replay()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
print("%8s %4s %4s | %8s %8s | %7s %7s | %6s | %s" % ("rows", "Cin", "Cout", "lib", "own", "hbm", "mfma", "slices", "max err dW / db"))
tl = to = 0.0
for M, K, N in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(M, K, device=dev); dz = torch.randn(M, N, device=dev)
    dW = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev)
    slices = L.lib().elo_weight_grad_slices(M, K, N)
    scratch = torch.empty(slices * (K * N + N), device=dev)
    def own():
        L.call("elo_dense_weight_grad", L.WeightGradArgs(M, K, N, x.data_ptr(), dz.data_ptr(), dW.data_ptr(), db.data_ptr(), scratch.data_ptr()), x)
    def lib():
        return x.t() @ dz, dz.sum(0)
    t_l, t_o = graph_time(lib), graph_time(own)
    own(); ref = x.double().t() @ dz.double()
    e_w = ((dW.double() - ref).abs().max() / ref.abs().max()).item(); e_b = ((db.double() - dz.double().sum(0)).abs().max() / (M ** 0.5)).item()
    hbm = M * (K + N) * 4 / 8e12 * 1e6; mf = 2.0 * M * (-(-K // 16) * 16) * (-(-N // 16) * 16) / 157.3e12 * 1e6
    tl += t_l; to += t_o
    print("%8d %4d %4d | %8.1f %8.1f | %7.1f %7.1f | %6d | %.1e %.1e" % (M, K, N, t_l, t_o, hbm, mf, slices, e_w, e_b))
print("sums (us): lib %.0f, own %.0f" % (tl, to))
