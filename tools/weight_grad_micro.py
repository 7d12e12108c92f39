"""elo_dense_weight_grad on the model's training shapes: time per call, achieved TFLOP/s and GB/s.
    python tools/weight_grad_micro.py"""
import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("efficientlo-net_amd._lib")
dev = torch.device("cuda:0")
shapes = [(172800, 138, 128), (172800, 128, 64), (172800, 64, 64), (172800, 10, 64), (172800, 128, 128), (115200, 90, 128),
          (1843200, 6, 8), (1843200, 8, 8), (1843200, 8, 16), (462848, 19, 16), (462848, 16, 32), (58368, 35, 32), (58368, 64, 64),
          (230400, 80, 128), (230400, 128, 64)]
for M, cin, cout in shapes:
    x = torch.randn((M, cin), device=dev); g = torch.randn((M, cout), device=dev)
    dW = torch.empty((cin, cout), device=dev); db = torch.empty((cout,), device=dev)
    slices = L.lib().elo_weight_grad_slices(M, cin, cout)
    sc = torch.empty((slices * (cin * cout + cout),), device=dev)
    a = L.WeightGradArgs(M, cin, cout, x.data_ptr(), g.data_ptr(), dW.data_ptr(), db.data_ptr(), sc.data_ptr())
    for _ in range(3): L.call("elo_dense_weight_grad", a, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.call("elo_dense_weight_grad", a, x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = x.double().t() @ g.double()
    err = float((dW.double() - ref).abs().max() / ref.abs().max())
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): torch.mm(x.t(), g)
    t1.record(); torch.cuda.synchronize()
    print("rows %8d %4d -> %4d: %7.1f us  %6.1f TFLOP/s  %6.0f GB/s  slices %4d  rel err %.1e | torch mm %7.1f us" % (
        M, cin, cout, us, 2.0 * M * cin * cout / us / 1e6, M * (cin + cout) * 4 / us / 1e3, slices, err, t0.elapsed_time(t1) / 20 * 1e3))
