"""shapes of the weight-gradient calls of one batch-8 training step"""
import importlib, sys, collections
import torch
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
model, training, synth, L = pkg("model"), pkg("training"), pkg("synth"), pkg("_lib")
dev = "cuda:0"; B = 8
net = model.PWCLONet(dev, seed=0); tr = training.Trainer(net)
f1, f2 = synth.frame_pair(B, 64, 1800, seed=1)
a, b = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
q = torch.tensor([[0.99995, 0, 0, 0.01]] * B, device=dev); t = torch.tensor([[[0.8], [0.0], [0.0]]] * B, device=dev)
tr.step(a, b, q, t)
seen = []
orig = L.call
def spy(entry, args, like):
    if entry == "elo_dense_weight_grad":
        s = L.lib().elo_weight_grad_slices(args.rows, args.Cin, args.Cout)
        seen.append((args.rows, args.Cin, args.Cout, s, s * (args.Cin * args.Cout + args.Cout)))
    return orig(entry, args, like)
L.call = spy
tr.step(a, b, q, t); torch.cuda.synchronize()
for r in sorted(seen, key=lambda r: r[4]): print("rows %8d  %3d -> %3d  slices %5d  floats %9d" % r)
