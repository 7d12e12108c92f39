"""One shape of elo_dense_rows, 20 launches (for rocprofv3 --pmc):  python tools/dense_rows_pmc.py ROWS CIN COUT [dx|fwd|wgrad]"""
import importlib, sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("efficientlo-net_amd._ops")
M, K, N = (int(v) for v in sys.argv[1:4]); mode = sys.argv[4] if len(sys.argv) > 4 else "dx"
dev = "cuda:0"
x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev); dz = torch.randn(M, N, device=dev)
mean, invstd, rm, rv = (torch.zeros(N, device=dev) for _ in range(4))
L = importlib.import_module("efficientlo-net_amd._lib")
dW = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev)
scratch = torch.empty(L.lib().elo_weight_grad_slices(M, K, N) * (K * N + N), device=dev)
for _ in range(20):
    if mode == "wgrad":
        L.call("elo_dense_weight_grad", L.WeightGradArgs(M, K, N, x.data_ptr(), dz.data_ptr(), dW.data_ptr(), db.data_ptr(), scratch.data_ptr()), x)
    elif mode == "dx": ops.dense_rows(dz, W, None, transposed=True)
    else: ops.dense_rows(x, W, b, moments=(1e-3, 0.1, mean, invstd, rm, rv))
torch.cuda.synchronize()
