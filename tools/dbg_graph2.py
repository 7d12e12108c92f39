import sys, numpy as np, torch
sys.path.insert(0, '.')
from importlib import import_module
model = import_module('efficientlo-net_amd.model'); synth = import_module('efficientlo-net_amd.synth')
dev='cuda:0'
net = model.PWCLONet(dev, seed=5)
f1,f2 = synth.frame_pair(1,64,1800,seed=41)
a,b = torch.from_numpy(f1.copy()).to(dev), torch.from_numpy(f2.copy()).to(dev)
eager = [x.clone() for x in net.forward(a,b)]
print("eager q", eager[0])
net.capture(1,64,1800)
net.load_inputs(a,b); torch.cuda.synchronize()
print("static in sum", net._static_in[0].abs().sum().item())
out = net.replay(); torch.cuda.synchronize()
print("replay q", out[0], out[6])
print("static in sum after", net._static_in[0].abs().sum().item())
# eager again on static inputs
print("eager on static", net.forward(*net._static_in)[0])
