"""The batch-norm row kernels of csrc/elo_train.hip on the big layers of a training step (batch 8, 64 x 1800): time per call inside one
captured graph of 20 and the fraction of 8 TB/s their algorithmic bytes make (inputs rotate over a ring >= 2 x 256 MB: HBM-cold).
    python tools/bn_micro.py"""
import importlib, sys
import torch
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
ops, L = pkg("_ops"), pkg("_lib")
dev = "cuda:0"
SHAPES = [(921600, 8), (921600, 16), (1843200, 8), (1843200, 16), (231424, 16), (231424, 32), (462848, 32), (172800, 64), (172800, 128), (230400, 64), (230400, 128),
          (115200, 128), (58368, 128), (29184, 64)]
def graph_time(fns):
    for f in fns: f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for f in fns: f()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * len(fns)) * 1e3
print("%8s %4s | %-22s %-22s %-22s %-22s" % ("rows", "C", "bn_stats (1 unit)", "bn_apply (2)", "bwd reduce+combine (2)", "bwd apply (3)"))
for M, C in SHAPES:
    unit = M * C * 4
    ring = max(2, int(2.2 * 256e6 / (2 * unit)) + 1)
    zs = [torch.randn(M, C, device=dev) for _ in range(ring)]; dys = [torch.randn(M, C, device=dev) for _ in range(ring)]
    out = torch.empty(M, C, device=dev)
    mean, invstd, gamma, beta = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv, sums = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.empty(2 * C, device=dev)
    scratch = torch.empty(L.lib().elo_bn_scratch_floats(C, 1), device=dev)
    n = max(20, ring)
    stats = [lambda z=zs[i % ring]: L.call("elo_bn_stats", L.BnStatsArgs(M, C, z.data_ptr(), scratch.data_ptr(), 1e-3, 0.1, mean.data_ptr(), invstd.data_ptr(), rm.data_ptr(), rv.data_ptr(), 1), z) for i in range(n)]
    apply_ = [lambda z=zs[i % ring]: L.call("elo_bn_apply", L.BnApplyArgs(M, C, z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, out.data_ptr(), 1), z) for i in range(n)]
    red = [lambda z=zs[i % ring], d=dys[i % ring]: L.call("elo_bn_backward", L.BnBackwardArgs(M, C, d.data_ptr(), z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, scratch.data_ptr(), sums.data_ptr(), None, 1), z) for i in range(n)]
    full = [lambda z=zs[i % ring], d=dys[i % ring]: L.call("elo_bn_backward", L.BnBackwardArgs(M, C, d.data_ptr(), z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, scratch.data_ptr(), sums.data_ptr(), out.data_ptr(), 1), z) for i in range(n)]
    ts, ta, tr, tf = graph_time(stats), graph_time(apply_), graph_time(red), graph_time(full)
    fr = lambda t, units: "%6.1f us (%.2f)" % (t, units * unit / (t * 1e-6) / 8e12)
    print("%8d %4d | %-22s %-22s %-22s %-22s" % (M, C, fr(ts, 1), fr(ta, 2), fr(tr, 2), fr(tf - tr, 3)))
    del zs, dys
