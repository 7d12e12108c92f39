"""Times (HIP events around a hipGraph of back-to-back launches, bench._time_launches) of cost-volume stage 1 and its
select-k in their different forms, one table per run:
    python tools/cv1_variants.py [--levels l0,l1,l2] [--batches 1,8]
  select   general wave-per-centre select-k (elo_fused_conv_select_k)
  dense4/8/16  the LDS-tiled form with 4 / 8 / 16 waves per tile (elo_fused_conv_select_k_dense)
  cv1        fused stage 1, select-k in-kernel (cv1_kernel: the tile kernel)
  meta       the tile kernel from idx / mask (cv1_meta_kernel)
  rr         the register-resident kernel from idx / mask (cv1_rr_kernel)
  select+rr  wave-per-centre select-k launch + cv1_rr_kernel
  prepass+rr what ELO_CV1_PREPASS=1 runs (select-k form by grid size + cv1_rr_kernel)
  bits       outputs equal bit for bit: [cv1 == prepass+rr, cv1 == meta, cv1 == rr]
"""
import argparse, importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda sub=None: importlib.import_module("efficientlo-net_amd" + ("." + sub if sub else ""))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--levels", default="l0,l1,l2"); ap.add_argument("--batches", default="1,8"); ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--half", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
fused, tf_util, synth, elo, L = pkg("fused"), pkg("tf_util"), pkg("synth"), pkg(), pkg("_lib")
SH = {"l0": (16, 225, 16, (11, 41)), "l1": (8, 113, 32, (7, 25)), "l2": (4, 57, 64, (5, 15)), "hr0": (32, 256, 16, (11, 41)),
      "full": (64, 1800, 16, (11, 41)), "full725": (64, 1800, 16, (7, 25))}
store = tf_util.VariableStore(dev, seed=0)
for lv in a.levels.split(","):
    H, W, C, win = SH[lv]
    N, Kq = H * W, 6
    for B in [int(b) for b in a.batches.split(",")]:
        g = torch.Generator(device="cpu").manual_seed(0)
        f1, f2 = synth.frame_pair(B, H, W, seed=5)
        x1, x2 = torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev)
        cast = (lambda t: t.half()) if a.half else (lambda t: t)
        ft1, ft2 = (cast(torch.randn((B, H, W, C), generator=g).to(dev)) for _ in range(2))
        order = torch.randperm(win[0] * win[1], generator=g).to(torch.int32).to(dev)
        hw = torch.from_numpy(synth.hw_index(B, H, W)).to(dev)
        sel = lambda dense: elo.fused_conv_select_k(x1, x2, hw, order, H, W, N, win[0], win[1], Kq, 0, 1000.0, 1, 1, want_valid=False, dense=dense)
        row = {"level": lv, "batch": B, "grid": "%dx%d" % (H, W), "window": "%dx%d" % win}
        row["select"] = round(bench._time_launches(lambda: sel(False), dev, a.reps) * 1e6, 2)
        ref = sel(False)
        for P in (4, 8, 16):
            L.lib().elo_debug_select_dense_waves(P)
            row["dense%d" % P] = round(bench._time_launches(lambda: sel(True), dev, a.reps) * 1e6, 2)
            got = sel(True)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[3], ref[3]), (lv, B, P)
        L.lib().elo_debug_select_dense_waves(0)
        if lv in ("full", "full725"):
            print(json.dumps(row), flush=True)
            continue
        with tf_util.default_store(store), torch.no_grad(), tf_util.variable_scope("cv_" + lv):
            Pk = fused.packed_layer
            layers = (Pk('CV_0', 10 + 2 * C, 128, row_order=fused.cv0_row_order(C)), Pk('CV_1', 128, 64), Pk('CV_2', 64, 64),
                      Pk('CV_xyz', 10, 64), Pk('sum_CV_0', 128, 128), Pk('sum_CV_1', 128, 64))
        grp = fused.Grouping(order, list(win), 1000)
        idx, m = ref[0], ref[3].reshape(B, N, Kq)
        x1f = x1.reshape(B, N, 3)
        ft1f = ft1.reshape(B, N, C)
        def run(prepass):
            pkg("tuning").set_host("cv_prepass", int(prepass))
            return fused.cv_stage1(x1f, ft1f, x2, ft2, None, None, *layers, group=grp, K=Kq)
        row["cv1"] = round(bench._time_launches(lambda: run("0"), dev, a.reps) * 1e6, 2)
        pre = lambda: fused.cv_stage1(x1f, ft1f, x2, ft2, idx, m, *layers)
        L.lib().elo_debug_cv1_rr(0)
        row["meta"] = round(bench._time_launches(pre, dev, a.reps) * 1e6, 2)
        o2 = pre()
        L.lib().elo_debug_cv1_rr(1)
        row["rr"] = round(bench._time_launches(pre, dev, a.reps) * 1e6, 2)
        o3 = pre()
        def sel_rr():
            i2, _, _, m2 = sel(False)
            return fused.cv_stage1(x1f, ft1f, x2, ft2, i2, m2.reshape(B, N, Kq), *layers)
        row["select+rr"] = round(bench._time_launches(sel_rr, dev, a.reps) * 1e6, 2)
        row["prepass+rr"] = round(bench._time_launches(lambda: run("1"), dev, a.reps) * 1e6, 2)
        o0, o1 = run("0"), run("1")
        L.lib().elo_debug_cv1_rr(-1)
        row["bits"] = [bool(torch.equal(o0, o)) for o in (o1, o2, o3)]
        row["maxdiff_rr"] = float((o0.float() - o3.float()).abs().max())
        print(json.dumps(row), flush=True)
