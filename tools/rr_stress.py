"""Repeats cost-volume stage 1 on the register-resident kernel and compares every run with the tile kernel's output bit for
bit, in both products modes, on shapes with more workgroups than CUs (a race shows up as a run that differs):
    python tools/rr_stress.py [runs] [--where] [--half-only]
--where: for a run that differs, which points differ and what they are in the kernel's terms -- workgroup (21 points x 6
neighbour rows = 126 of its 128 rows), the waves that own those rows (16 rows each), how many of the 64 channels."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda sub=None: importlib.import_module("efficientlo-net_amd" + ("." + sub if sub else ""))
fused, tf_util, synth, elo, lib = pkg("fused"), pkg("tf_util"), pkg("synth"), pkg(), pkg("_lib")
DEV = torch.device("cuda:0")
ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
RUNS = int(ARGS[0]) if ARGS else 40
WHERE, HALF_ONLY = "--where" in sys.argv, "--half-only" in sys.argv


def where(got, tile, K):
    """(B,N,64) outputs -> text: the workgroups / waves / channels of the rows that differ."""
    g, w = got.reshape(-1, 64), tile.reshape(-1, 64)
    diff = (g != w)
    pts = torch.nonzero(diff.any(1)).flatten().tolist()
    P = 128 // K
    by_wg = {}
    for p in pts:
        by_wg.setdefault(p // P, []).append(p % P)
    lines = ["  %d points differ in %d workgroups (of %d)" % (len(pts), len(by_wg), (g.shape[0] + P - 1) // P)]
    for wg, pps in list(by_wg.items())[:6]:
        waves = sorted({r // 16 for pp in pps for r in range(pp * K, pp * K + K)})
        nch = int(diff[[wg * P + pp for pp in pps]].sum(1).float().mean())
        err = float((g[[wg * P + pp for pp in pps]] - w[[wg * P + pp for pp in pps]]).abs().max())
        lines.append("    tile %d: points %s -> rows of waves %s; %d of 64 channels on average, max |diff| %.3g" % (wg, pps, waves, nch, err))
    return "\n".join(lines)


t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for mode in (("half",) if HALF_ONLY else ("split", "half")):
    for (B, H, W, C, win) in [(2, 16, 225, 16, (11, 41)), (8, 16, 225, 16, (11, 41)), (8, 8, 113, 32, (7, 25)), (4, 16, 225, 64, (11, 41))]:
        with fused.products(mode):
            f1, f2 = synth.frame_pair(B, H, W, seed=H * W + C)
            rng = np.random.default_rng(C)
            fa, fb = (rng.normal(0, 1, (B, H, W, C)).astype(np.float32) for _ in range(2))
            N, K = H * W, 6
            perm = rng.permutation(win[0] * win[1]).astype(np.int32)
            store = tf_util.VariableStore(DEV, seed=3)
            with tf_util.default_store(store), torch.no_grad():
                P = fused.packed_layer
                layers = (P("c0", 10 + 2 * C, 128, row_order=fused.cv0_row_order(C)), P("c1", 128, 64), P("c2", 64, 64),
                          P("cx", 10, 64), P("s0", 128, 128), P("s1", 128, 64))
                hw = t(synth.hw_index(B, H, W))
                idx, _, _, m = elo.fused_conv_select_k(t(f1), t(f2), hw, t(perm), H, W, N, win[0], win[1], K, 0, 1000.0, 1, 1, want_valid=False)
                m = m.reshape(B, N, K)
                run = lambda: fused.cv_stage1(t(f1).reshape(B, N, 3), t(fa).reshape(B, N, C), t(f2), t(fb), idx, m, *layers)
                lib.lib().elo_debug_cv1_rr(0); tile = run()
                lib.lib().elo_debug_cv1_rr(1)
                bad, shown = 0, 0
                for _ in range(RUNS):
                    got = run()
                    if not torch.equal(got, tile):
                        bad += 1
                        if WHERE and shown < 2:
                            shown += 1
                            print(where(got, tile, K), flush=True)
                lib.lib().elo_debug_cv1_rr(-1)
        print(mode, (B, H, W, C), "runs differing from the tile kernel: %d of %d" % (bad, RUNS), flush=True)
