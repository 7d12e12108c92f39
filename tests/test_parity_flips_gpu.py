"""Free-running parity of the whole pyramid as a MEASURED statement ("Next round" 2).

north_star: "within 1e-4 for fp32 features/pose".  The pyramid takes DISCRETE decisions downstream of a pose (a warped
point's projection cell, the per-cell winner, the neighbours of a grouping op), so two correct fp32 implementations
whose coarse poses differ in the 7th digit can take one of them differently and then differ by 1e-3.  This file does not
hide that behind a loose tolerance; per (seed, pair, level) it demands

  * teacher-forced (the oracle warps a level by the PRODUCT's coarse pose): |product - oracle| <= 1e-4, OR the product's
    own projection kernel put a warped point in another cell than the oracle did FROM THE SAME POSE (the product's cell
    of every point is read back from its scratch, pwclo_model.PROJECTION_TAP: a point on a cell border is decided by the
    last place of atan2f / asinf -- round 4; until then no seed of the suite had hit one at the forced level);
  * free-running: |product - oracle| <= 1e-4, OR a counted discrete flip explains the miss: the oracle's own
    free-running and teacher-forced runs -- same code, same weights, coarse poses that differ by the product's
    deviation (<= 1e-4 by the first check, ~1e-6 measured) -- took a different discrete decision at that level or a
    coarser one (oracle/ops_np.discrete_trace: every neighbour index / mask, every point -> cell assignment and cell
    winner, and -- with fp16 feature storage -- every stored half: a feature that rounds to the neighbouring fp16 value
    moves by 5e-4 relative, five times the tolerance).  A miss with NO flip anywhere upstream would be accumulated
    arithmetic error: that fails;
  * the share of (pair, level) outputs within 1e-4 free-running stays above a floor measured on the GPU.
"""
import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import ops_np as O
from util_params import export, randomise, shuffle_fn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4
LEVELS = (3, 2, 1, 0)                       # order of computation: coarse to fine
OUT_OF = {0: (0, 1), 1: (2, 3), 2: (4, 5), 3: (6, 7)}      # level -> positions of (q, t) in the output tuple


def _flips(free_tr, forced_tr, B):
    """{level: int[B]} -- entries that differ between the two traces, per batch element."""
    assert len(free_tr.events) == len(forced_tr.events)
    out = {lvl: np.zeros(B, np.int64) for lvl in LEVELS}
    cell_no = {}
    for (lvl, kind, a), (lvl2, kind2, b) in zip(free_tr.events, forced_tr.events):
        assert lvl == lvl2 and kind == kind2 and a.shape == b.shape
        if lvl is None:
            assert np.array_equal(a, b)      # upstream of every pose: identical by construction
            continue
        if kind == "grid":                   # a projection's output: continuous in the pose (compared with the product's, below)
            continue
        if kind == "cell":                   # one event per batch element, in order
            i = cell_no.get(lvl, 0)
            cell_no[lvl] = i + 1
            out[lvl][i % B] += int((a != b).any(0).sum())
        else:
            diff = (a != b).reshape(B, -1)
            out[lvl] += diff.sum(1)
    return out


def _product_flips(tap, forced_tr, B):
    """{level: int[B]} -- warped points (non-zero ones) that the PRODUCT's projection kernel put in another cell than the
    teacher-forced oracle did from the same pose: float32 arithmetic that differs in the last place (atan2f / asinf on the
    GPU against numpy's) decides a point on a cell border differently.  At 128 x 2048 a column index is ~10^3: one ulp of it
    is 1e-4 of a cell, so a few points per image sit that close to a border."""
    out = {lvl: np.zeros(B, np.int64) for lvl in LEVELS}
    oracle, grids = {}, {}
    for lvl, kind, a in forced_tr.events:
        if kind == "cell" and lvl is not None:
            oracle.setdefault(lvl, []).append(a)
        if kind == "grid" and lvl is not None:
            grids[lvl] = a
    for lvl, scratch, b_, n_, h_, w_, grid in tap:
        cells = scratch[b_ * h_ * w_ + 4 * b_: b_ * h_ * w_ + 4 * b_ + b_ * n_].view(torch.int32).reshape(b_, n_).cpu().numpy()
        events = oracle[lvl][-b_:]                      # the level's projection: one event per batch element
        # a cell whose content differs by more than a millimetre holds ANOTHER point (a different winner of the minimum
        # range, or a point that went to the neighbouring cell): the continuous difference of the two warps is ~1e-5 m
        moved = (np.abs(grid.detach().cpu().numpy() - grids[lvl]).max(-1) > 1e-3).reshape(b_, -1)
        for b in range(b_):
            cell, _same, nonzero = events[b]
            out[lvl][b] += int(((cells[b] != cell) & (nonzero == 1)).sum()) + int(moved[b].sum())
    return out


def statistic(B, seeds, features="f32", H=64, W=1800):
    """Rows of (seed, pair, level, free error, forced error, flips at this level or coarser, product-vs-oracle cell flips)."""
    model, perm, synth = load_pkg("model"), load_pkg("perm"), load_pkg("synth")
    pm = load_pkg("pwclo_model")
    f16 = features == "f16"
    rows = []
    for seed in seeds:
        f1, f2 = synth.frame_pair(B, H, W, seed=seed)
        net = model.PWCLONet(DEV, seed=5, perm_source=perm.PermSource(fn=shuffle_fn),
                             feature_dtype=torch.float16 if f16 else torch.float32)
        both = torch.from_numpy(np.concatenate([f1, f2], 0)).to(DEV)
        net.forward(both[:B], both[B:])
        randomise(net.store, seed=seed + 100)
        pm.PROJECTION_TAP = tap = []
        try:
            got = [x.detach().cpu().numpy() for x in net.forward(both[:B], both[B:])]
        finally:
            pm.PROJECTION_TAP = None
        params = export(net.store)
        with O.feature_storage(np.float16 if f16 else None):
            with O.discrete_trace() as tr_free:
                free = O.get_model_from_projection(params, shuffle_fn, f1, f2)
            with O.discrete_trace() as tr_forced:
                forced = O.get_model_from_projection(params, shuffle_fn, f1, f2,
                                                     coarse_pose={3: (got[6], got[7]), 2: (got[4], got[5]), 1: (got[2], got[3])})
        flips = _flips(tr_free, tr_forced, B)
        pflips = _product_flips(tap, tr_forced, B)
        for b in range(B):
            upstream, p_upstream = 0, 0
            for lvl in LEVELS:
                upstream += int(flips[lvl][b])
                p_upstream += int(pflips[lvl][b])
                # normalised error |got - ref| / (1 + |ref|): <= TOL is the suite's close(atol=1e-4, rtol=1e-4)
                err = lambda ref: max(float((np.abs(got[i][b] - ref[i][b]) / (1.0 + np.abs(ref[i][b]))).max()) for i in OUT_OF[lvl])
                rows.append(dict(seed=seed, pair=b, level=lvl, free=err(free), forced=err(forced), flips=upstream, product_flips=p_upstream))
    return rows


def summarise(rows):
    n = len(rows)
    inside = [r for r in rows if r["free"] <= TOL]
    missed = [r for r in rows if r["free"] > TOL]
    return dict(outputs=n, within_tol=len(inside), share=len(inside) / n,
                misses=len(missed), misses_without_flip=sum(1 for r in missed if r["flips"] == 0),
                worst_forced=max(r["forced"] for r in rows), worst_free=max(r["free"] for r in rows),
                forced_misses=sum(1 for r in rows if r["forced"] > TOL),
                product_cell_flips_in_forced_misses=sorted(r["product_flips"] for r in rows if r["forced"] > TOL),
                worst_forced_without_product_flip=max([r["forced"] for r in rows if r["product_flips"] == 0] or [0.0]),
                worst_free_without_flip=max([r["free"] for r in rows if r["flips"] == 0] or [0.0]),
                flips_in_misses=sorted(r["flips"] for r in missed))


# floors below the shares measured on MI355X; the assertions that carry the parity claim are the first two -- the share
# only documents how often a discrete flip happens with random weights
# measured (tools/parity_flips.py, profiles/r03_parity_flips.txt): fp32 62/64 and 63/64 outputs within 1e-4 free-running,
# fp16 storage 20/32; every miss carries 1..14 k flipped decisions; worst teacher-forced error 4.8e-5 (fp16: 8.3e-5)
@pytest.mark.parametrize("B,seeds,features,floor,H,W", [(1, range(200, 216), "f32", 0.85, 64, 1800), (8, (300, 301), "f32", 0.85, 64, 1800),
                                                        (8, (310,), "f16", 0.45, 64, 1800),
                                                        (8, (320,), "f32", 0.75, 128, 2048), (8, (321,), "f16", 0.35, 128, 2048)])   # configs[4] at batch 8
def test_free_running_misses_are_counted_flips(B, seeds, features, floor, H, W):
    rows = statistic(B, list(seeds), features, H, W)
    s = summarise(rows)
    print("\nparity statistic B=%d %dx%d %s: %s" % (B, H, W, features, s))
    for r in rows:
        # teacher-forced: within the tolerance, unless the PRODUCT itself put a warped point in another projection cell than
        # the oracle did from the same pose (a point on a cell border, decided by the last place of atan2f / asinf)
        assert r["forced"] <= TOL or r["product_flips"] > 0, r
        if r["flips"] == 0 and r["product_flips"] == 0:
            assert r["free"] <= TOL, r                            # no discrete decision differs: plain fp32 agreement
    assert s["share"] >= floor, s
    forced_ok = sum(1 for r in rows if r["forced"] <= TOL)
    assert forced_ok >= 0.9 * len(rows), (forced_ok, len(rows))   # border points are rare: nearly every output agrees teacher-forced
