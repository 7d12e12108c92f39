"""Free-running parity of the whole pyramid as a MEASURED statement ("Next round" 2).

north_star: "within 1e-4 for fp32 features/pose".  The pyramid takes DISCRETE decisions downstream of a pose (a warped
point's projection cell, the per-cell winner, the neighbours of a grouping op), so two correct fp32 implementations
whose coarse poses differ in the 7th digit can take one of them differently and then differ by 1e-3.  This file does not
hide that behind a loose tolerance; per (seed, pair, level) it demands

  * teacher-forced (the oracle warps a level by the PRODUCT's coarse pose): |product - oracle| <= 1e-4, OR the product's
    own projection kernel put a warped point in another cell than the oracle did FROM THE SAME POSE (the product's cell
    of every point is read back from its scratch, pwclo_model.PROJECTION_TAP) AND that point provably sits within
    BORDER_ULPS float32 ulps of a cell border (its coordinate recomputed in float64), or two ranges of a cell tie within one
    ulp -- _product_flips; any other difference between the product's and the oracle's projected grids FAILS the test
    (round 5: until then "the grids differ by more than a millimetre" was accepted as the cause itself);
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
        if kind in ("grid", "warped"):       # a projection's output / input points: continuous in the pose (compared with the product's, below)
            continue
        if kind == "cell":                   # one event per batch element, in order
            i = cell_no.get(lvl, 0)
            cell_no[lvl] = i + 1
            out[lvl][i % B] += int((a != b).any(0).sum())
        else:
            diff = (a != b).reshape(B, -1)
            out[lvl] += diff.sum(1)
    return out


BORDER_ULPS = 4          # float32 ulps OF THE INDEX within which two correct implementations may truncate a coordinate differently


def _near_integer(v64, scale):
    """|v - nearest integer| in float32 ulps at magnitude `scale` (the largest float32 intermediate the coordinate went through)."""
    return abs(v64 - np.rint(v64)) / float(np.spacing(np.float32(max(abs(scale), 1.0))))


def _product_flips(tap, forced_tr, B):
    """{level: int[B]} -- projection decisions the PRODUCT took differently from the teacher-forced oracle (same pose), each
    one ATTRIBUTED to float rounding; anything else fails here, so that a wrong projection / scatter kernel cannot supply
    its own excuse:
      * a non-zero warped point whose product cell (read back from the projection's scratch) is not the oracle's counts only
        if the two warps agree (1e-4 relative), the cells are neighbours along the coordinate(s) that differ, and that
        coordinate -- recomputed in float64 (oracle/ops_np.projection_coordinates64) from the product's OR the oracle's warped
        point -- lies within BORDER_ULPS float32 ulps of the index from an integer: a point on a cell border, decided by the
        last place of atan2f / asinf / the warp;
      * a cell whose content differs by more than a millimetre (the two warps differ by ~1e-5 m) without such a point entering
        or leaving it counts only if its two smallest ranges tie within one ulp (another winner of the minimum range);
      * any other difference of the two grids raises."""
    out = {lvl: np.zeros(B, np.int64) for lvl in LEVELS}
    oracle, grids, warped = {}, {}, {}
    for lvl, kind, a in forced_tr.events:
        if lvl is None:
            continue
        if kind == "cell":
            oracle.setdefault(lvl, []).append(a)
        if kind == "warped":
            warped.setdefault(lvl, []).append(a)
        if kind == "grid":
            grids[lvl] = a
    for lvl, scratch, b_, n_, h_, w_, grid, pts in tap:
        base = b_ * h_ * w_ + 4 * b_
        cells = scratch[base: base + b_ * n_].view(torch.int32).reshape(b_, n_).cpu().numpy()
        events, owarp = oracle[lvl][-b_:], warped[lvl][-b_:]    # the level's projection: one event per batch element
        G = grid.detach().cpu().numpy().reshape(b_, h_ * w_, 3)
        OG = grids[lvl].reshape(b_, h_ * w_, 3)
        P = pts.detach().cpu().numpy().reshape(b_, n_, 3)
        _az, _vres, voff = O.projection_constants(h_, w_)
        for b in range(b_):
            cell, _same, nonzero = events[b]
            op, orng = owarp[b][:, :3], owarp[b][:, 3]
            assert np.abs(P[b] - op).max() <= 1e-4 * (1.0 + np.abs(op).max()), "the two warps differ beyond float rounding"
            differ = np.nonzero((cells[b] != cell) & (nonzero == 1))[0]
            explained = set()
            if len(differ):
                pc, pt = O.projection_coordinates64(P[b][differ], h_, w_)
                oc, ot = O.projection_coordinates64(op[differ], h_, w_)
            for j, p in enumerate(differ):
                (prow, pcol), (orow, ocol) = divmod(int(cells[b][p]), w_), divmod(int(cell[p]), w_)
                why = []
                if pcol != ocol:
                    # (the column coordinate is (pi - atan2f) / az: a DIFFERENCE of two numbers of magnitude pi, so its float32 error is
                    #  an ulp of pi / az = an ulp of W / 2 in index units, also where the index itself is small)
                    margin = min(_near_integer(pc[j], max(abs(pc[j]), w_ / 2.0)), _near_integer(oc[j], max(abs(oc[j]), w_ / 2.0)))
                    # neighbours along the row -- or across the SEAM: the column coordinate (pi - atan2(y, x)) / az runs from 0 (azimuth
                    # +pi) to W (azimuth -pi, clipped to W - 1): a point on the negative x axis is in column 0 or W - 1 by the sign of a zero
                    assert abs(pcol - ocol) in (1, w_ - 1) and margin <= BORDER_ULPS, \
                        ("level %d pair %d point %d: columns %d / %d, column coordinate %.9f is %.1f ulps from a border" % (lvl, b, p, pcol, ocol, pc[j], margin))
                    why.append("col %.1f ulp" % margin)
                if prow != orow:
                    margin = min(_near_integer(pt[j], max(abs(pt[j]), float(voff))), _near_integer(ot[j], max(abs(ot[j]), float(voff))))
                    assert abs(prow - orow) == 1 and margin <= BORDER_ULPS, \
                        ("level %d pair %d point %d: rows %d / %d, row coordinate %.9f is %.1f ulps from a border" % (lvl, b, p, prow, orow, pt[j], margin))
                    why.append("row %.1f ulp" % margin)
                explained.update((int(cells[b][p]), int(cell[p])))
                print("  attributed cell flip: level %d pair %d point %d (%s)" % (lvl, b, p, ", ".join(why)))
            moved = np.nonzero(np.abs(G[b] - OG[b]).max(-1) > 1e-3)[0]
            ties = 0
            for c in moved:
                if int(c) in explained:
                    continue
                rr = np.sort(orng[(cell == c) & (nonzero == 1)])
                assert len(rr) >= 2 and rr[1] - rr[0] <= np.spacing(rr[0]), \
                    ("level %d pair %d cell %d holds another point although no border point entered or left it and its ranges do "
                     "not tie: %s" % (lvl, b, c, rr[:3]))
                ties += 1
                print("  attributed winner flip: level %d pair %d cell %d (ranges %.9g / %.9g)" % (lvl, b, c, rr[0], rr[1]))
            out[lvl][b] += len(differ) + ties
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
