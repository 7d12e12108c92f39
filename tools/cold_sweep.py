"""How cold is cold: the per-operator cost-volume kernels at the 128 x 2048 l0 shape, batch 8, timed over rings of tensor
sets with 0 (one set: warm), 0.5, 1, 2 and 4 x the 256 MB Infinity Cache of other sets' traffic between two uses of a
set (bench._time_ring).  The reading is HBM-cold once it stops moving with the ring.
    python tools/cold_sweep.py [--half] [--batch 8]"""
import argparse, importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = bench.pkg
ap = argparse.ArgumentParser(); ap.add_argument("--half", action="store_true"); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--grid", default="32x256")
a = ap.parse_args()
dev = torch.device("cuda:0")
ops, synth, elo = pkg("_ops"), pkg("synth"), pkg()
B, (H, W), C, Kq, Kp = a.batch, map(int, a.grid.split("x")), 16, 6, 4
N = H * W
g = torch.Generator(device="cpu").manual_seed(3)
cast = (lambda x: x.half()) if a.half else (lambda x: x)
cvb = bench.cost_volume_bytes(N, C, Kq, Kp, 2 if a.half else 4)
f1, f2 = synth.frame_pair(B, H, W, seed=6)
x1, x2 = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
ft1, ft2 = (cast(torch.randn((B, H, W, C), generator=g).to(dev)) for _ in range(2))
hw = torch.from_numpy(synth.hw_index(B, H, W)).to(dev)
order = torch.randperm(11 * 41, generator=g).to(torch.int32).to(dev)
idx_q, _, _, m_q = elo.fused_conv_select_k(x1, x2, hw, order, H, W, N, 11, 41, Kq, 0, 1000.0, 1, 1, want_valid=False)
order_p = torch.randperm(15, generator=g).to(torch.int32).to(dev)
idx_p, _, _, m_p = elo.fused_conv_random_k(x1, x1, hw, order_p, H, W, N, 3, 5, Kp, 0, 1000.0, 1, 1, want_valid=False)
m_q, m_p = m_q.reshape(B, N, Kq), m_p.reshape(B, N, Kp)
cost = cast(torch.randn((B, H, W, 64), generator=g).to(dev))
lq, vq = (cast(torch.randn((B, N, Kq, 64), generator=g).to(dev)) for _ in range(2))
lp, vp = (cast(torch.randn((B, N, Kp, 64), generator=g).to(dev)) for _ in range(2))
legs = {"A1": (ops.cv_encode1, (x1.reshape(B, N, 3), ft1.reshape(B, N, C), x2, ft2, idx_q, m_q)),
        "P1": (ops.masked_softmax_pool, (lq, vq, m_q)),
        "A2": (ops.cv_encode2, (x1, ft1, cost, idx_p, m_p)),
        "P2": (ops.masked_softmax_pool, (lp, vp, m_p))}
print("grid %dx%d batch %d %s: us per launch (fraction of 8 TB/s on the term's algorithmic bytes)" % (H, W, B, "f16" if a.half else "f32"))
for term, (fn, args) in legs.items():
    nbytes = cvb[term] * B
    row = ["%s %6.1f MB" % (term, nbytes / 1e6)]
    warm = bench._time_launches(lambda: fn(*args), dev, 20)
    row.append("one set %6.2f (%.3f)" % (warm * 1e6, nbytes / warm / 8e12))
    for mult in (0.5, 1, 2, 4):
        sec, ring = bench._time_ring(fn, args, None, dev, between=int(mult * bench.LLC_BYTES))
        row.append("%gx %6.2f (%.3f) R=%d" % (mult, sec * 1e6, nbytes / sec / 8e12, ring["ring"]))
    print("  ".join(row), flush=True)
