"""Data-parallel plumbing: one process per GPU, torch.distributed ("nccl" == RCCL over xGMI on ROCm,
"gloo" in the CPU tests).  The reference is single-GPU (main.py:1-3, :58); this is new work.

Frame pairs are independent, so ranks own contiguous blocks of the pair stream and there is no
collective on the data path.  The exchanges that exist:
  * inference: all_gather of the (n,7) [q|t] poses so that rank 0 can chain them in order
    (main.py:557-572);
  * training: ONE all_reduce per step over a single flat fp32 bucket holding every gradient
    (899 134 floats = 3.6 MB for the full model): at this size a ring all-reduce over xGMI is
    latency-bound, so one call on one contiguous buffer -- no per-tensor calls, no bucketing hooks.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: the first n_items % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_poses(q, t, n_total=None):
    """q (n_local,4), t (n_local,3) on every rank -> (n_total,7) on every rank, in global pair order
    (ranks own contiguous blocks, see shard_range).  Uneven blocks are padded for the collective."""
    local = torch.cat([q, t], -1).contiguous()
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if n_total is None:
        counts = torch.tensor([local.shape[0]], device=local.device)
        all_counts = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts)
        sizes = [int(c.item()) for c in all_counts]
    else:
        sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    width = max(sizes)
    padded = local.new_zeros((width, 7))
    padded[:local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], 0)


def quat2mat(q):
    """Rotation matrix of a (w,x,y,z) quaternion in numpy float64 (what main.py:401-434 computes: the quaternion need not be
    unit length, a near-zero one gives the identity).  Written from the outer product P = (2/|q|^2) q q^T:
    R = I + [vector part of P, symmetrised off the diagonal] with the w-row of P supplying the antisymmetric part."""
    quat = np.asarray([float(v) for v in q], np.float64)
    norm2 = float(quat @ quat)
    if norm2 < 1e-8:
        return np.eye(3)
    P = np.outer(quat, quat) * (2.0 / norm2)
    sym = P[1:, 1:]                                   # 2 v v^T / |q|^2
    wv = P[0, 1:]                                     # 2 w v / |q|^2
    skew = np.array([[0.0, -wv[2], wv[1]], [wv[2], 0.0, -wv[0]], [-wv[1], wv[0], 0.0]])
    return np.eye(3) * (1.0 - np.trace(sym)) + sym + skew


def chain_poses(poses_n7, Tr=None):
    """Ordered running product of the per-sample transforms, main.py:557-572: TT = Tr [R|t] Tr^-1, T_final <- T_final TT
    (T_final = TT for sample 0).  Returns (n, 12) rows, ONE per sample -- row 0 is sample 0's own prediction, not a
    prepended identity -- exactly what the reference writes to *_pred.txt and what evaluate.pose_rows returns (this is the
    same function on the gathered (n,7) [q | t] log of distributed.gather_poses)."""
    from .evaluate import pose_rows
    p = np.asarray(poses_n7, dtype=np.float64).reshape(-1, 7)
    return pose_rows(p[:, :4], p[:, 4:7], np.eye(4) if Tr is None else Tr)


class FlatGradBucket:
    """All gradients of `params` in ONE contiguous fp32 buffer; .grad of every parameter is a view into it,
    so the backward pass fills the bucket and all_reduce_mean() is a single collective."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros((n,), dtype=torch.float32, device=self.params[0].device)
        off = 0
        self.views = []
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.attach()

    def attach(self):
        """.grad of every parameter = its view of the bucket (the state the optimiser and the collective work on)."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self):
        self.flat.zero_()

    def release(self):
        """.grad = None before a backward pass: autograd then HANDS OVER each gradient instead of launching one
        `grad += new` kernel per parameter into the views (~870 launches of a few hundred bytes per training step)."""
        for p in self.params:
            p.grad = None

    def collect(self):
        """The gradients autograd produced since release() -> their views (one multi-tensor copy), views attached again.
        A parameter the loss did not reach keeps the zeros of zero()."""
        have = [(v, p.grad) for p, v in zip(self.params, self.views) if p.grad is not None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self.attach()

    @staticmethod
    def world_size():
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    force_collective = False      # True: the all-reduce also runs on a world of ONE rank (bench.py ELO_BENCH_FORCE_DIST=1: RCCL's code path on a 1-GPU box)

    @classmethod
    def has_collective(cls):
        """Whether a training step has its exchange step: a process group of more than one rank (or the one-rank rehearsal)."""
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or cls.force_collective)

    def all_reduce_mean(self):
        if self.has_collective():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())
        return self.flat
