"""Which torch (aten) operators launch kernels in one eager training step, and from where: op, input shapes, innermost repo frame, count, GPU time.
    python tools/train_aten_ops.py [B]"""
import importlib, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
model, training, synth = pkg("model"), pkg("training"), pkg("synth")
dev = "cuda:0"; B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = model.PWCLONet(dev, seed=0); tr = training.Trainer(net)
f1, f2 = synth.frame_pair(B, 64, 1800, seed=1)
a, b = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
q = torch.tensor([[0.99995, 0, 0, 0.01]] * B, device=dev); t = torch.tensor([[[0.8], [0.0], [0.0]]] * B, device=dev)
for _ in range(3): tr.step(a, b, q, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(a, b, q, t); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::") or e.device_time <= 0: continue
    if any(c.name.startswith("aten::") and c.device_time > 0 for c in (e.cpu_children or [])): continue      # count leaves only
    frame = next((s for s in (e.stack or []) if "efficientlo-net_amd" in s or "/tools/" in s), "(autograd engine)")
    key = (e.name, str(e.input_shapes)[:70], frame.split("efficientlo-net_amd/")[-1][:60])
    agg[key][0] += 1; agg[key][1] += e.device_time
tot = sum(v[1] for v in agg.values())
print("aten leaves with GPU time: %d calls, %.2f ms" % (sum(v[0] for v in agg.values()), tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%7.1f us x%-3d %-22s %-70s %s" % (v[1], v[0], k[0], k[1], k[2]))
