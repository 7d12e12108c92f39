"""The parity statistic's excuse for a teacher-forced miss -- "the product put a warped point in another projection cell" --
must be ATTRIBUTED (tests/test_parity_flips_gpu.py::_product_flips): checked here on the CPU with a stand-in for the product's
tap built from the oracle's own projection, then corrupted the way a wrong projection / scatter kernel would corrupt it."""
import numpy as np
import pytest
import torch

from oracle import ops_np as O
from test_parity_flips_gpu import BORDER_ULPS, _product_flips

H, W, B, N = 8, 113, 2, 904


def _case(seed=0):
    rng = np.random.default_rng(seed)
    az = rng.uniform(-np.pi, np.pi, (B, N))
    el = np.deg2rad(rng.uniform(-24.0, 1.5, (B, N)))
    r = rng.uniform(3.0, 40.0, (B, N))
    P = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], -1).astype(np.float32)
    P[:, :7] = 0                                               # a few zero points (padding): never counted
    with O.discrete_trace() as tr:
        tr.level = 2
        grid, _ = O.ProjectPC2SphericalRing(P, None, H, W)
    cells = np.stack([e[2][0] for e in tr.events if e[1] == "cell"]).astype(np.int32)          # (B,N)
    return P, grid, cells, tr


def _tap(P, grid, cells):
    scratch = torch.zeros(B * H * W + 4 * B + 2 * B * N, dtype=torch.int32)
    scratch[B * H * W + 4 * B: B * H * W + 4 * B + B * N] = torch.from_numpy(cells.reshape(-1))
    return [(2, scratch, B, N, H, W, torch.from_numpy(grid.copy()), torch.from_numpy(P.copy()))]


def test_identical_projection_counts_nothing():
    P, grid, cells, tr = _case()
    flips = _product_flips(_tap(P, grid, cells), tr, B)
    assert all(int(v.sum()) == 0 for v in flips.values())


def test_a_point_in_a_wrong_cell_is_not_an_excuse():
    P, grid, cells, tr = _case()
    col64, _ = O.projection_coordinates64(P[0], H, W)
    far = int(np.argmax(np.where(P[0].any(1), np.abs(col64 - np.rint(col64)), 0)))      # a point in the middle of its cell
    bad = cells.copy()
    bad[0, far] += 1 if bad[0, far] % W < W - 1 else -1
    with pytest.raises(AssertionError, match="ulps from a border"):
        _product_flips(_tap(P, grid, bad), tr, B)


def test_a_cell_holding_another_point_is_not_an_excuse():
    P, grid, cells, tr = _case()
    g = grid.copy().reshape(B, H * W, 3)
    c = int(cells[1, 100])
    g[1, c] += 0.5                                              # the scatter wrote something else there; every point's cell agrees
    with pytest.raises(AssertionError, match="holds another point"):
        _product_flips(_tap(P, g.reshape(grid.shape), cells), tr, B)


def test_a_border_point_is_counted_with_its_margin():
    P, grid, cells, tr = _case()
    az32, _, _ = O.projection_constants(H, W)
    # move one point onto a column border: azimuth such that (pi - atan2) / az is an integer + 1 ulp
    k, rr, el = 40, 12.0, np.deg2rad(-10.0)
    theta = np.float64(np.float32(np.pi)) - np.float64(az32) * k * (1 + 1e-9)
    P2 = P.copy()
    P2[0, 50] = np.array([rr * np.cos(el) * np.cos(theta), rr * np.cos(el) * np.sin(theta), rr * np.sin(el)], np.float32)
    with O.discrete_trace() as tr2:
        tr2.level = 2
        grid2, _ = O.ProjectPC2SphericalRing(P2, None, H, W)
    cells2 = np.stack([e[2][0] for e in tr2.events if e[1] == "cell"]).astype(np.int32)
    col64, _ = O.projection_coordinates64(P2[0, 50:51], H, W)
    margin = abs(col64[0] - np.rint(col64[0])) / np.spacing(np.float32(max(col64[0], W / 2.0)))
    if margin > BORDER_ULPS:
        pytest.skip("float32 rounding of the constructed point left it %.1f ulps from the border" % margin)
    other = cells2.copy()                                       # the "product" truncated the other way
    other[0, 50] += -1 if int(np.rint(col64[0])) <= col64[0] else 1
    g = grid2.copy().reshape(B, H * W, 3)                       # ... and its grid follows: the point sits in the neighbouring cell
    for c in (int(cells2[0, 50]), int(other[0, 50])):
        g[0, c] = 0
    # rebuild the two touched cells the way scatter_min_range would with the point moved
    r = np.sqrt((P2[0].astype(np.float32) ** 2).sum(1)).astype(np.float32)
    for c in (int(cells2[0, 50]), int(other[0, 50])):
        members = np.nonzero(other[0] == c)[0]
        if len(members):
            m = r[members].min()
            g[0, c] = P2[0, members[r[members] == m]].sum(0)
    flips = _product_flips(_tap(P2, g.reshape(grid2.shape), other), tr2, B)
    assert int(flips[2][0]) >= 1 and int(flips[2][1]) == 0


def test_the_seam_of_the_cylinder_is_a_border_too():
    """Column 0 and column W - 1 are neighbours: the column coordinate runs from 0 (azimuth +pi) to W (azimuth -pi, clipped), so a
    point on the negative x axis is decided by the sign of a rounding error in y.  A claim of the far column is accepted for a
    point on the seam and refused for a point in the middle of column 0."""
    P, grid, cells, tr = _case(seed=3)
    P2 = P.copy()
    P2[0, 60] = np.array([-12.0, 1e-9, -1.0], np.float32)          # azimuth pi - 8e-11: column coordinate ~ 1e-9
    with O.discrete_trace() as tr2:
        tr2.level = 2
        grid2, _ = O.ProjectPC2SphericalRing(P2, None, H, W)
    cells2 = np.stack([e[2][0] for e in tr2.events if e[1] == "cell"]).astype(np.int32)
    assert cells2[0, 60] % W == 0
    other = cells2.copy()
    other[0, 60] += W - 1                                           # the "product" saw azimuth -pi: column W - 1 of the same row
    g = grid2.copy().reshape(B, H * W, 3)
    r = np.sqrt((P2[0].astype(np.float32) ** 2).sum(1)).astype(np.float32)
    for c in (int(cells2[0, 60]), int(other[0, 60])):
        members = np.nonzero(other[0] == c)[0]
        g[0, c] = P2[0, members[r[members] == r[members].min()]].sum(0) if len(members) else 0
    flips = _product_flips(_tap(P2, g.reshape(grid2.shape), other), tr2, B)
    assert int(flips[2][0]) >= 1
    P3 = P.copy()
    P3[0, 60] = np.array([-12.0, 0.2, -1.0], np.float32)            # well inside column 0
    with O.discrete_trace() as tr3:
        tr3.level = 2
        grid3, _ = O.ProjectPC2SphericalRing(P3, None, H, W)
    cells3 = np.stack([e[2][0] for e in tr3.events if e[1] == "cell"]).astype(np.int32)
    bad = cells3.copy()
    bad[0, 60] += W - 1
    with pytest.raises(AssertionError, match="ulps from a border"):
        _product_flips(_tap(P3, grid3, bad), tr3, B)
