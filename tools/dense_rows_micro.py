"""elo_dense_rows against the library GEMM on the tall-and-skinny products of a training step (batch 8, 64 x 1800): time per
call in one captured graph of 20 calls each (no host launch floor), agreement with torch, and the fused batch moments.
    python tools/dense_rows_micro.py"""
import importlib, sys
import torch
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
ops, L = pkg("_ops"), pkg("_lib")
dev = "cuda:0"
SHAPES = [(921600, 6, 8), (921600, 8, 8), (921600, 8, 16), (231424, 19, 16), (231424, 16, 16), (231424, 16, 32), (29184, 35, 32),
          (29184, 32, 64), (14848, 67, 64), (14848, 64, 128), (58368, 138, 128), (58368, 128, 64), (58368, 10, 64), (7296, 192, 128),
          (172800, 42, 128), (172800, 128, 64), (172800, 64, 64), (172800, 10, 64), (172800, 128, 128), (115200, 144, 128),
          (115200, 128, 64), (230400, 67, 128), (230400, 128, 64), (28800, 80, 128), (28800, 144, 128), (1824, 192, 128)]
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
print("%8s %4s %4s | %8s %8s %8s | %8s %8s | %7s | %s" % ("rows", "Cin", "Cout", "lib fwd", "lib+stat", "own fwd", "lib dx", "own dx", "ideal", "max err fwd / dx / mean / invstd"))
tot = [0.0] * 5
for M, K, N in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) * 0.2; b = torch.randn(N, device=dev); dz = torch.randn(M, N, device=dev)
    mean, invstd = torch.empty(N, device=dev), torch.empty(N, device=dev)
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    scratch = torch.empty(L.lib().elo_bn_scratch_floats(N, 1), device=dev)
    def lib_stats():
        z = torch.addmm(b, x, W)
        L.call("elo_bn_stats", L.BnStatsArgs(M, N, z.data_ptr(), scratch.data_ptr(), 1e-3, 0.1, mean.data_ptr(), invstd.data_ptr(), rm.data_ptr(), rv.data_ptr(), 1), z)
        return z
    pow2 = N & (N - 1) == 0
    own = lambda: ops.dense_rows(x, W, b, moments=(1e-3, 0.1, mean, invstd, rm, rv))
    t_lib = graph_time(lambda: torch.addmm(b, x, W)); t_ls = graph_time(lib_stats) if pow2 else float("nan"); t_own = graph_time(own)
    t_ldx = graph_time(lambda: dz @ W.t()); t_odx = graph_time(lambda: ops.dense_rows(dz, W, None, transposed=True))
    z_ref = torch.addmm(b.double(), x.double(), W.double()); z = own(); m_own, i_own = mean.clone(), invstd.clone()
    e_f = ((z.double() - z_ref).abs().max() / z_ref.abs().max()).item()
    dx_ref = dz.double() @ W.double().t(); dx = ops.dense_rows(dz, W, None, transposed=True)
    e_b = ((dx.double() - dx_ref).abs().max() / dx_ref.abs().max()).item()
    m_ref = z_ref.mean(0); i_ref = 1.0 / torch.sqrt(z_ref.var(0, unbiased=False) + 1e-3)
    e_m = (m_own.double() - m_ref).abs().max().item(); e_i = ((i_own.double() - i_ref).abs() / i_ref).max().item()
    ideal = M * (K + N) * 4 / 8e12 * 1e6
    for i, v in enumerate((t_lib, t_ls if pow2 else t_lib, t_own, t_ldx, t_odx)): tot[i] += v
    print("%8d %4d %4d | %8.1f %8.1f %8.1f | %8.1f %8.1f | %7.1f | %.1e %.1e %.1e %.1e" % (M, K, N, t_lib, t_ls, t_own, t_ldx, t_odx, ideal, e_f, e_b, e_m, e_i))
print("sums (us): lib fwd %.0f, lib fwd + stats %.0f, own fwd with stats %.0f | lib dx %.0f, own dx %.0f" % tuple(tot))
