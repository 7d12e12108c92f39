"""Experiment: cost of conv-layer variants (addmm+relu_ vs _addmm_activation) at the model's shapes."""
import torch, time
from torch.profiler import profile, ProfilerActivity
dev='cuda:0'
def bench(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for (M,K,N) in [(115200,6,8),(115200,8,16),(21600,42,128),(21600,128,64),(3600,144,128),(7296,138,128),(28800,67,128),(116,192,128)]:
    x=torch.randn(M,K,device=dev); W=torch.randn(K,N,device=dev); b=torch.randn(N,device=dev)
    f1=lambda: torch.addmm(b,x,W).relu_()
    f2=lambda: torch._addmm_activation(b,x,W)
    f3=lambda: torch.relu(torch.nn.functional.linear(x, W.t().contiguous(), b))
    ok=torch.allclose(f1(),f2(),atol=1e-4,rtol=1e-4)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        f2(); torch.cuda.synchronize()
    names=[e.key[:60] for e in prof.key_averages()]
    print((M,K,N),"addmm+relu %.1fus  _addmm_activation %.1fus ok=%s kernels=%s"%(bench(f1),bench(f2),ok,names))
