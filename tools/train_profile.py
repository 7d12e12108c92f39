import importlib, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
model, training, synth = pkg("model"), pkg("training"), pkg("synth")
dev = "cuda:0"; B = 2
net = model.PWCLONet(dev, seed=0); tr = training.Trainer(net)
f1, f2 = synth.frame_pair(B, 64, 1800, seed=1)
a, b = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
q = torch.tensor([[0.99995, 0, 0, 0.01]] * B, device=dev); t = torch.tensor([[[0.8], [0.0], [0.0]]] * B, device=dev)
for _ in range(2): tr.step(a, b, q, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(a, b, q, t); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=12, max_name_column_width=60))
