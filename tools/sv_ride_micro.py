"""Kernel time of a level's tail with and without softmax_valid's ride (run under rocprofv3 --kernel-trace --stats):
python tools/sv_ride_micro.py [iterations]   -- the four level shapes of a 64 x 1800 forward at batch 1, fp32 storage."""
import importlib, sys
import numpy as np
import torch
sys.path.insert(0, ".")
pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
fused, tf_util, ops = pkg("fused"), pkg("tf_util"), pkg("_ops")
DEV = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
r = lambda *s: t(rng.normal(0, 1, s).astype(np.float32))
store = tf_util.VariableStore(DEV, seed=0)
for N, C in ((228, 64), (904, 32), (3600, 16)):
    def job(tag):
        with tf_util.default_store(store), torch.no_grad(), tf_util.variable_scope("m_%d_%s" % (N, tag)):
            P = fused.packed_layer
            layers = [P("a0", 64 + C, 128), P("a1", 128, 64)]
            layers2 = [P("b0", C + 64 + 64, 128, row_order=fused.stage2_row_order(C, 64, 64)), P("b1", 128, 64)]
        return dict(sources=[r(1, N, 64), r(1, N, C)], layers=layers, before=r(1, N, C), after=r(1, N, 64), layers2=layers2)
    ja, jb = job("w"), job("c")
    xyz = r(1, N, 3)
    for i in range(iters):
        fused.mlp2_pair(ja, jb)
        fused.mlp2_pair(ja, jb, sv=ops.SvPartials(xyz))
    torch.cuda.synchronize()
with tf_util.default_store(store), torch.no_grad(), tf_util.variable_scope("single"):
    layers = [fused.packed_layer("p0", 128 + 64, 128), fused.packed_layer("p1", 128, 64)]
srcs = [r(1, 58, 128), r(1, 58, 64)]
xyz = r(1, 58, 3)
for i in range(iters):
    fused.mlp(srcs, layers)
    fused.mlp(srcs, layers, sv=ops.SvPartials(xyz, srcs[1]))
torch.cuda.synchronize()
