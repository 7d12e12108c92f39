"""One eager training step at batch B under the torch profiler: which ATen operators launch the small kernels
(count per operator and per input shape).   python tools/train_op_counts.py [B]"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(a, b, q, t); torch.cuda.synchronize()
ops = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and not any(c.name.startswith("aten::") for c in e.cpu_children) \
            and e.name not in ("aten::view", "aten::reshape", "aten::as_strided", "aten::slice", "aten::select", "aten::empty", "aten::empty_like",
                               "aten::empty_strided", "aten::expand", "aten::t", "aten::transpose", "aten::detach", "aten::alias", "aten::squeeze",
                               "aten::unsqueeze", "aten::narrow", "aten::permute", "aten::_unsafe_view", "aten::resize_", "aten::is_nonzero",
                               "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::result_type", "aten::unbind", "aten::split"):
        ops[(e.name, str(e.input_shapes)[:70])] += 1
tot = collections.Counter()
for (n, s), c in ops.items(): tot[n] += c
print(tot.most_common(20))
for (n, s), c in ops.most_common(40): print("%4d  %-28s %s" % (c, n, s))
