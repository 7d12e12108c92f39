"""CPU, world_size 2 over gloo: the N>1 paths of efficientlo-net_amd/distributed.py (pair sharding,
pose all_gather in global order, single flat-bucket gradient all-reduce) and pose chaining."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_pairs, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import importlib
    D = importlib.import_module("efficientlo-net_amd.distributed")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = D.shard_range(n_pairs, rank, world)
        # every "pose" encodes its global pair index, so order and ownership are checkable
        idx = torch.arange(lo, hi, dtype=torch.float32)
        q = torch.stack([idx, idx + 0.25, idx + 0.5, idx + 0.75], -1)
        t = torch.stack([idx * 10, idx * 10 + 1, idx * 10 + 2], -1)
        for n_total in (None, n_pairs):
            all_p = D.gather_poses(q, t, n_total)
            assert all_p.shape == (n_pairs, 7)
            assert torch.equal(all_p[:, 0], torch.arange(n_pairs, dtype=torch.float32))
            assert torch.equal(all_p[:, 4], torch.arange(n_pairs, dtype=torch.float32) * 10)
        # one all-reduce over one flat buffer: mean of rank-dependent gradients
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7)),
                  torch.nn.Parameter(torch.randn(2, 2, 2))]
        bucket = D.FlatGradBucket(params)
        assert bucket.flat.numel() == 15 + 7 + 8
        loss = sum(((rank + 1) * (i + 1)) * p.sum() for i, p in enumerate(params))
        loss.backward()
        assert params[0].grad.data_ptr() == bucket.flat.data_ptr()        # grads live inside the bucket
        bucket.all_reduce_mean()
        mean_scale = sum(r + 1 for r in range(world)) / world
        for i, p in enumerate(params):
            assert torch.allclose(p.grad, torch.full_like(p, mean_scale * (i + 1)))
        if world >= 8:      # BASELINE configs[3]'s exchange at its real size: ONE all-reduce of the 899 134-float gradient bucket
            big = [torch.nn.Parameter(torch.zeros(899134))]
            bucket8 = D.FlatGradBucket(big)
            (big[0].sum() * float(rank + 1)).backward()
            bucket8.all_reduce_mean()
            assert bucket8.flat.numel() == 899134 and torch.equal(big[0].grad, torch.full_like(big[0], (world + 1) / 2.0))
        if rank == 0:
            out.put("ok")
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, out)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get() == "ok"


def test_world_size_8_gloo_on_the_stream_of_configs3():
    """Eight ranks (the node configs[3] names) over gloo: the 15 237 pairs of KITTI seq 00-06 (kitti_dataset.py:28) shard
    UNEVENLY (15237 = 8 x 1904 + 5: five ranks own one pair more), the pose gather returns them in global order on every rank
    with and without the total known, and the full-size flat gradient bucket goes through ONE all-reduce."""
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, 15237, out)) for r in range(8)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get() == "ok"
    D = load_pkg("distributed")
    sizes = [D.shard_range(15237, r, 8)[1] - D.shard_range(15237, r, 8)[0] for r in range(8)]
    assert sizes == [1905] * 5 + [1904] * 3


def test_shard_range_covers_everything_once():
    D = load_pkg("distributed")
    for n, world in ((15237, 8), (7, 2), (3, 8), (0, 4)):
        spans = [D.shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_chain_poses_matches_direct_products():
    D = load_pkg("distributed")
    rng = np.random.default_rng(0)
    q = rng.normal(0, 1, (5, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.normal(0, 1, (5, 3))
    Tr = np.eye(4); Tr[:3, :3] = D.quat2mat(rng.normal(0, 1, 4)); Tr[:3, 3] = [0.1, -0.2, 0.3]
    rows = D.chain_poses(np.concatenate([q, t], 1), Tr)
    assert rows.shape == (5, 12)                                     # one row per sample, no prepended identity (main.py:557-572)
    T = np.eye(4)
    for i in range(5):
        M = np.eye(4); M[:3, :3] = D.quat2mat(q[i]); M[:3, 3] = t[i]
        T = T @ Tr @ M @ np.linalg.inv(Tr)
        assert np.allclose(rows[i], T[:3].reshape(12))
    assert np.array_equal(rows, load_pkg("evaluate").pose_rows(q, t, Tr))      # the one chaining convention
    R = D.quat2mat([np.cos(0.3), 0, 0, np.sin(0.3)])               # rotation about z by 0.6 rad
    assert np.allclose(R, [[np.cos(0.6), -np.sin(0.6), 0], [np.sin(0.6), np.cos(0.6), 0], [0, 0, 1]])
