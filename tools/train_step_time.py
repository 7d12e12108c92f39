"""Time Trainer.step (torch-twin forward + autograd backward + flat-bucket all-reduce (no-op at world 1) + Adam)."""
import importlib, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
model, training, synth = pkg("model"), pkg("training"), pkg("synth")
dev = "cuda:0"
for B in ([int(x) for x in sys.argv[1:]] or (1, 4, 8)):      # python tools/train_step_time.py [B ...]
    net = model.PWCLONet(dev, seed=0)
    tr = training.Trainer(net)
    f1, f2 = synth.frame_pair(B, 64, 1800, seed=1)
    a, b = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
    q = torch.tensor([[0.99995, 0, 0, 0.01]] * B, device=dev); t = torch.tensor([[[0.8], [0.0], [0.0]]] * B, device=dev)
    for _ in range(3): tr.step(a, b, q, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): tr.step(a, b, q, t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("batch %d: %.1f ms per training step, %.1f pairs/s, peak memory %.2f GB" % (B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 2**30))
    tg = training.Trainer(model.PWCLONet(dev, seed=0), capturable=True).capture(a, b, q, t)      # the step as one hipGraph
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tg.step_graph(a, b, q, t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("batch %d: %.1f ms per CAPTURED training step, %.1f pairs/s" % (B, dt * 1e3, B / dt))
