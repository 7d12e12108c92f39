"""Launch one cost-volume kernel repeatedly (for rocprofv3 --kernel-trace / --pmc captures).

    python tools/roofline_micro.py --kernel cv1|pool|pool2|encode1|encode2|select32|select32_l2|random16 --batch B [--reps N]
cv1     = fused stage 1 at l0 (16x225, K=6, C=16)          -> cv1_kernel
cv1_recorded = the same launch of a real 64x1800 forward, on its own tensors (bench.py's roofline object; its live counter passes)
pool    = per-operator masked softmax pool, K=6 (P1 term)  -> softmax_pool_vec_kernel
pool2   = the same with K=4 (P2 term)
encode1 = per-operator gather + geometry encode (A1 term)  -> cv_encode1_col_kernel (C = 16), cv_encode1_vec_kernel otherwise
encode2 = per-operator stage-2 gather + encode (A2 term)    -> cv_encode2_vec_kernel
Prints the algorithmic bytes / flops per launch it used.
"""
import argparse, importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda sub=None: importlib.import_module("efficientlo-net_amd" + ("." + sub if sub else ""))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--kernel", default="cv1"); ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--reps", type=int, default=20); ap.add_argument("--pregrouped", action="store_true"); ap.add_argument("--time", action="store_true")
ap.add_argument("--half", action="store_true", help="fp16 feature storage (cv1/encode1/encode2/pool/pool2)")
ap.add_argument("--cold", action="store_true",
                help="pool / pool2 / encode1 / encode2 / cv1_recorded: every launch on its own tensor set of a ring with >= 2 x 256 MB of "
                     "other sets' traffic between two uses of a set (bench._time_ring's ring): the launches read HBM, not the Infinity Cache")
ap.add_argument("--grid", default=os.environ.get("ELO_PMC_GRID", "16x225"),
                help="the l0 grid HxW of the cv1 / encode / pool kernels: 16x225 (64x1800 scans) or 32x256 (128x2048: BASELINE configs[4])")
a = ap.parse_args()
dev = torch.device("cuda:0")
ops, fused, tf_util, synth, elo = pkg("_ops"), pkg("fused"), pkg("tf_util"), pkg("synth"), pkg()
B, (H, W), C, Kq = a.batch, map(int, a.grid.split("x")), 16, 6
N = H * W
g = torch.Generator(device="cpu").manual_seed(0)
f1, f2 = synth.frame_pair(B, H, W, seed=5)
xyz1, xyz2 = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
feat1 = torch.randn((B, H, W, C), generator=g).to(dev); feat2 = torch.randn((B, H, W, C), generator=g).to(dev)
order = torch.randperm(11 * 41, generator=g).to(torch.int32).to(dev)
cvb = bench.cost_volume_bytes(N, C, Kq, 4, 2 if a.half else 4)
cast = (lambda x: x.half()) if a.half else (lambda x: x)
if a.kernel == "cv1_recorded":            # stage 1 at l0 of a REAL forward, on that forward's tensors: what bench.py's roofline object times
    L = bench.recorded_cost_volume(dev, B, 64, 1800, a.half)["l0"]
    run, call = L["run1"], L["call1"]
    cb = bench.cost_volume_bytes(L["N"], L["C"], L["Kq"], L["Kp"], 2 if a.half else 4)
    info = {"flops": bench.cv1_flops(L["N"], L["C"], L["Kq"]) * B, "algorithmic_bytes": (cb["A1"] + cb["P1"]) * B}
elif a.kernel == "cv1":
    store = tf_util.VariableStore(dev, seed=0)
    with tf_util.default_store(store), torch.no_grad():
        P = fused.packed_layer
        layers = (P('CV_0', 10 + 2 * C, 128, row_order=fused.cv0_row_order(C)), P('CV_1', 128, 64), P('CV_2', 64, 64),
                  P('CV_xyz', 10, 64), P('sum_CV_0', 128, 128), P('sum_CV_1', 128, 64))
    feat1, feat2 = cast(feat1), cast(feat2)                  # --half: fp16 feature storage (BASELINE configs[2])
    grp = fused.Grouping(order, [11, 41], 1000)
    if a.pregrouped:
        hw = torch.from_numpy(synth.hw_index(B, H, W)).to(dev)
        idx, _, _, m = elo.fused_conv_select_k(xyz1, xyz2, hw, order, H, W, N, 11, 41, Kq, 0, 1000.0, 1, 1, want_valid=False)
        m = m.reshape(B, N, Kq)
        run = lambda: fused.cv_stage1(xyz1.reshape(B, N, 3), feat1.reshape(B, N, C), xyz2, feat2, idx, m, *layers)
    else:
        run = lambda: fused.cv_stage1(xyz1.reshape(B, N, 3), feat1.reshape(B, N, C), xyz2, feat2, None, None, *layers, group=grp, K=Kq)
    info = {"flops": bench.cv1_flops(N, C, Kq) * B, "algorithmic_bytes": (cvb["A1"] + cvb["P1"]) * B}
elif a.kernel in ("pool", "pool2"):
    K = Kq if a.kernel == "pool" else 4
    logits = cast(torch.randn((B, N, K, 64), generator=g).to(dev)); values = cast(torch.randn((B, N, K, 64), generator=g).to(dev))
    mask = (torch.rand((B, N, K), generator=g) > 0.1).float().to(dev)
    call = (ops.masked_softmax_pool, (logits, values, mask), {})
    run = lambda: ops.masked_softmax_pool(logits, values, mask)
    info = {"algorithmic_bytes": cvb["P1" if a.kernel == "pool" else "P2"] * B}
elif a.kernel in ("select32", "select32_l2", "random16"):
    # the stand-alone grouping ops (no valid_* outputs).  select32: every pixel of a 64x1800 grid, 5x35 window, K=32
    # (the window / K of the l2_origin cost volume on BASELINE config 1's grid); select32_l2: that call at its real
    # size (4x57); random16: BASELINE configs[0] (9x15, K=16, d=0.5)
    gh, gw, win, K, dist, op = {"select32": (64, 1800, (5, 35), 32, 1000.0, elo.fused_conv_select_k),
                                "select32_l2": (4, 57, (5, 35), 32, 1000.0, elo.fused_conv_select_k),
                                "random16": (64, 1800, (9, 15), 16, 0.5, elo.fused_conv_random_k)}[a.kernel]
    g1, g2 = synth.frame_pair(B, gh, gw, seed=5)
    gx1, gx2 = torch.from_numpy(g1).to(dev), torch.from_numpy(g2).to(dev)
    hw = torch.from_numpy(synth.hw_index(B, gh, gw)).to(dev)
    KT, n = win[0] * win[1], gh * gw
    order_g = torch.randperm(KT, generator=g).to(torch.int32).to(dev)
    run = lambda: op(gx1, gx2, hw, order_g, gh, gw, n, win[0], win[1], K, 0, dist, 1, 1, want_valid=False)
    info = {"algorithmic_bytes": (n * 12 + n * 8 + n * 12 + KT * 4 + n * K * 16) * B}
elif a.kernel == "encode2":
    Kp = 4
    order2 = torch.randperm(3 * 5, generator=g).to(torch.int32).to(dev)
    hw = torch.from_numpy(synth.hw_index(B, H, W)).to(dev)
    idx, _, _, m = elo.fused_conv_random_k(xyz1, xyz1, hw, order2, H, W, N, 3, 5, Kp, 0, 1000.0, 1, 1, want_valid=False)
    m = m.reshape(B, N, Kp)
    cost = cast(torch.randn((B, H, W, 64), generator=g).to(dev))
    h1 = cast(feat1)
    call = (ops.cv_encode2, (xyz1, h1, cost, idx, m), {})
    run = lambda: ops.cv_encode2(xyz1, h1, cost, idx, m)
    info = {"algorithmic_bytes": cvb["A2"] * B}
else:
    hw = torch.from_numpy(synth.hw_index(B, H, W)).to(dev)
    idx, _, _, m = elo.fused_conv_select_k(xyz1, xyz2, hw, order, H, W, N, 11, 41, Kq, 0, 1000.0, 1, 1, want_valid=False)
    m = m.reshape(B, N, Kq)
    h1, h2 = cast(feat1), cast(feat2)
    call = (ops.cv_encode1, (xyz1.reshape(B, N, 3), h1.reshape(B, N, C), xyz2, h2, idx, m), {})
    run = lambda: ops.cv_encode1(xyz1.reshape(B, N, 3), h1.reshape(B, N, C), xyz2, h2, idx, m)
    info = {"algorithmic_bytes": cvb["A1"] * B}
if a.cold:
    fn, args, kwargs = call
    foot = bench._footprint(args, kwargs, fn(*args, **kwargs))
    ring = max(3, -(-2 * bench.LLC_BYTES // foot) + 1)
    sets = [(args, kwargs)] + [(bench._clone_tensors(args), bench._clone_tensors(kwargs)) for _ in range(ring - 1)]
    passes = max(2, -(-a.reps // ring))
    keep = []
    for p in range(passes):                  # outputs of a pass stay alive: every launch of the pass writes its own
        keep = [fn(*x, **k) for x, k in sets]
    torch.cuda.synchronize()
    info.update(cold=True, ring=ring, footprint_bytes=foot, launches=ring * passes)
    if a.time:
        info["us"] = round(bench._time_ring(fn, args, kwargs, dev)[0] * 1e6, 2)
else:
    for _ in range(a.reps):
        run()
    torch.cuda.synchronize()
    if a.time:
        info["us"] = round(bench._time_launches(run, dev, 200) * 1e6, 2)
print(json.dumps({"kernel": a.kernel, "batch": B, "half": a.half, "grid": a.grid, **info}))
