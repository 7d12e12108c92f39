import sys, numpy as np, torch
sys.path.insert(0, '.')
from importlib import import_module
elo = import_module('efficientlo-net_amd'); synth = import_module('efficientlo-net_amd.synth'); ops = import_module('efficientlo-net_amd._ops')
dev='cuda:0'
f1,f2 = synth.frame_pair(1,16,225,seed=1)
a,b = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
idx = torch.from_numpy(synth.hw_index(1,16,225)).to(dev)
perm = torch.randperm(15, dtype=torch.int32).to(dev)
def f():
    sel,_,_,m = elo.fused_conv_random_k(a,b,idx,perm,16,225,3600,3,5,4,0,1.0,1,1,want_valid=False)
    return sel, m
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    e = f()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    print("capture stream", torch.cuda.current_stream().cuda_stream, torch.cuda.is_current_stream_capturing())
    o = f()
print("before replay sum", o[1].sum().item())
g.replay(); torch.cuda.synchronize()
print("eager mask sum", e[1].sum().item(), "graph", o[1].sum().item(), torch.equal(e[0],o[0]))
