"""Repeats cost-volume stage 1 on the register-resident kernel and compares every run with the tile kernel's output bit for
bit, in both products modes, on shapes with more workgroups than CUs (a race shows up as a run that differs):
    python tools/rr_stress.py [runs]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda sub=None: importlib.import_module("efficientlo-net_amd" + ("." + sub if sub else ""))
fused, tf_util, synth, elo, lib = pkg("fused"), pkg("tf_util"), pkg("synth"), pkg(), pkg("_lib")
DEV = torch.device("cuda:0")
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for mode in ("split", "half"):
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
                bad = sum(int(not torch.equal(run(), tile)) for _ in range(RUNS))
                lib.lib().elo_debug_cv1_rr(-1)
        print(mode, (B, H, W, C), "runs differing from the tile kernel: %d of %d" % (bad, RUNS), flush=True)
