"""Training step for the hot path: forward (batch-statistics BN, dropout, fresh visiting orders) ->
get_loss -> backward -> ONE flat-bucket all-reduce over RCCL -> Adam.  Mirrors main.py:120-176,344-397
(the reference's single-GPU loop) with the data-parallel exchange of SURVEY.md section 5 added.
Also here: the reference's training-time input augmentation (main.py:259-297) and checkpoint save / restore
(variables + Adam state as .npz, or the variables as a TensorFlow bundle, tf_checkpoint.py).  The epoch driver
(evaluation cadence, best-model directory policy, main.py:185-249) is out of scope.
"""
import math
import os

import numpy as np
import torch

from . import perm, pwclo_model, tf_checkpoint, tf_util
from .distributed import FlatGradBucket
from .model import graph_capture

BASE_LEARNING_RATE, DECAY_STEP, DECAY_RATE = 0.001, 200000, 0.7          # main.py:46-51
BN_INIT_DECAY, BN_DECAY_DECAY_RATE, BN_DECAY_CLIP = 0.5, 0.5, 0.99       # main.py:62-65


def learning_rate(step, batch_size):
    """main.py:120-128: staircase exponential decay, floored at 1e-5."""
    return max(BASE_LEARNING_RATE * DECAY_RATE ** ((step * batch_size) // DECAY_STEP), 0.00001)


def bn_decay(step, batch_size):
    """main.py:130-138."""
    momentum = BN_INIT_DECAY * BN_DECAY_DECAY_RATE ** ((step * batch_size) // DECAY_STEP)
    return min(BN_DECAY_CLIP, 1 - momentum)


def data_augmentation(rng=None):
    """main.py:259-297 DataAugmentation: a random near-identity rigid transform T_trans (4,4) applied to frame 2 in
    training -- rotations about x, y (sigma 0.01, clipped to +-0.02) and z (sigma 0.05, clipped to +-0.1), each times
    pi/4, composed Rx.Ry.Rz; translation sigma (0.5, 0.1, 0.05) clipped to (1.0, 0.2, 0.15).  The reference's scale
    matrix is uniform(1,1) = identity.  `rng`: a numpy Generator (the reference uses the global numpy state)."""
    rng = np.random.default_rng() if rng is None else rng
    draw = lambda sigma, bound: float(np.clip(sigma * rng.standard_normal(), -bound, bound))
    ax, ay, az = (draw(0.01, 0.02) * math.pi / 4, draw(0.01, 0.02) * math.pi / 4, draw(0.05, 0.1) * math.pi / 4)

    def rot(i, j, angle):                              # rotation in the (i, j) coordinate plane
        R = np.eye(3)
        R[i, i] = R[j, j] = math.cos(angle)
        R[i, j], R[j, i] = -math.sin(angle), math.sin(angle)
        return R

    T = np.eye(4)
    T[:3, :3] = rot(1, 2, ax) @ rot(2, 0, ay) @ rot(0, 1, az)
    T[:3, 3] = (draw(0.5, 1.0), draw(0.1, 0.2), draw(0.05, 0.15))
    return T


class FlatAdam:
    """Adam (betas 0.9 / 0.999, eps 1e-8; main.py:171-176) as ONE launch per step: the parameters are re-seated as views of
    one flat fp32 buffer (as FlatGradBucket does for the gradients), the two moment buffers are flat as well, and
    `elo_adam_flat` walks all 899 134 floats at once -- torch's foreach Adam is ~46 launches for the 382 tensors.  The
    step's scalars go to the device in ONE small async copy (`hyper`), which also makes the optimiser capturable: a
    hipGraph reads them, the host rewrites them before every replay -- from a RING of pinned slots, each guarded by an event
    (a captured step never synchronises, so the host runs steps ahead: with one pinned buffer it overwrote step t's scalars
    while step t's copy was still queued).
    `epsilon`: where eps sits.  "tf" (default) is tf.train.AdamOptimizer's, the optimiser the reference trains with
    (main.py:174): p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps); "torch" is torch.optim.Adam's:
    p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  They differ where sqrt(v) is of the order of eps (early steps
    of the small pose-head gradients, w_x / w_q).  Every parameter handed over is stepped, also one the loss did not reach
    (zero gradient: its moments decay) -- torch.optim.Adam skips those."""
    RING = 32

    def __init__(self, params, bucket, lr=BASE_LEARNING_RATE, betas=(0.9, 0.999), eps=1e-8, epsilon="tf"):
        from . import _lib
        self._lib = _lib
        if epsilon not in ("tf", "torch"):
            raise ValueError("epsilon is 'tf' (tf.train.AdamOptimizer, the reference's) or 'torch' (torch.optim.Adam)")
        self.epsilon = epsilon
        self.params, self.bucket, self.lr, self.betas, self.eps = [p for p in params if p.requires_grad], bucket, lr, betas, eps
        if sum(p.numel() for p in self.params) != bucket.flat.numel():
            raise ValueError("FlatAdam and its FlatGradBucket must be built from the same parameter list (%d against %d values)"
                             % (sum(p.numel() for p in self.params), bucket.flat.numel()))
        dev = self.params[0].device
        n = bucket.flat.numel()
        self.flat = torch.empty((n,), dtype=torch.float32, device=dev)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.views_m, self.views_v = [], []
        off = 0
        with torch.no_grad():
            for p in self.params:                              # the variables move into the flat buffer (same values, same objects)
                k = p.numel()
                view = self.flat[off:off + k].view_as(p)
                view.copy_(p)
                p.data = view
                self.views_m.append(self.exp_avg[off:off + k].view_as(p))
                self.views_v.append(self.exp_avg_sq[off:off + k].view_as(p))
                off += k
        self.t = 0                                             # steps taken (the bias corrections' exponent)
        self.hyper = torch.zeros((4,), dtype=torch.float32, device=dev)
        self._host = torch.zeros((self.RING, 4), dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros((self.RING, 4))
        self._copied = [None] * self.RING                      # event behind the async copy that last read slot i
        self._args = _lib.AdamFlatArgs(n, self.flat.data_ptr(), bucket.flat.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self.hyper.data_ptr(), betas[0], betas[1], 1.0 - betas[0], 1.0 - betas[1])

    def set_hyper(self):
        """The scalars of step t + 1 to the device (call before step(), or before replaying a graph that recorded it)."""
        t = self.t + 1
        slot = t % self.RING
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()                   # the copy that read this slot RING steps ago has run
        host = self._host[slot]
        bc1, bc2 = 1.0 - self.betas[0] ** t, 1.0 - self.betas[1] ** t
        if self.epsilon == "tf":                               # lr_t = lr sqrt(bc2) / bc1, plain sqrt(v) + eps
            host[0], host[1] = self.lr * math.sqrt(bc2) / bc1, 1.0
        else:
            host[0], host[1] = self.lr / bc1, 1.0 / math.sqrt(bc2)
        host[2] = self.eps
        self.hyper.copy_(host, non_blocking=True)
        if self.hyper.is_cuda:
            ev = self._copied[slot] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.hyper.device))
            self._copied[slot] = ev
        self.t = t

    def launch(self):
        """The update itself (the launch a captured training step records)."""
        if self.params[0].data_ptr() != self.flat.data_ptr():
            raise RuntimeError("the variables no longer live in this FlatAdam's flat buffer (another FlatAdam / Trainer re-seated "
                               "them): its update would step an orphaned copy -- keep ONE optimiser per set of variables")
        self._lib.call("elo_adam_flat", self._args, self.flat)

    def step(self):
        self.set_hyper()
        self.launch()

    def state_dict(self):
        """{"state": {i: {"step", "exp_avg", "exp_avg_sq"}}} as torch.optim's, the moments as views of the flat buffers."""
        return {"state": {i: {"step": torch.tensor(float(self.t)), "exp_avg": m, "exp_avg_sq": v}
                          for i, (m, v) in enumerate(zip(self.views_m, self.views_v))}}

    def load_state(self, moments_m, moments_v, t):
        with torch.no_grad():
            for i, (m, v) in enumerate(zip(self.views_m, self.views_v)):
                if i in moments_m:
                    m.copy_(torch.as_tensor(moments_m[i]).reshape(m.shape))
                    v.copy_(torch.as_tensor(moments_v[i]).reshape(v.shape))
        self.t = int(t)


class Trainer:
    def __init__(self, net, capturable=False):
        """`capturable`: Adam keeps its step counts and learning rate on the device, so that a whole optimisation step
        can be recorded into a hipGraph (`capture` / `step_graph`)."""
        self.net = net
        self.capturable = capturable
        self._graph = None
        pwclo_model.create_variables(net.store)
        dev = net.device
        self.w_x = torch.nn.Parameter(torch.tensor(0.0, device=dev))     # main.py:151
        self.w_q = torch.nn.Parameter(torch.tensor(-2.5, device=dev))    # main.py:152
        self.params = net.store.parameters() + [self.w_x, self.w_q]      # 382 tensors, 899 134 values
        self.bucket = FlatGradBucket(self.params)
        self.opt = FlatAdam(self.params, self.bucket, lr=BASE_LEARNING_RATE)   # main.py:174; one launch per step, capturable as it is
        self.step_count = 0

    def _global_batch(self, B):
        """Samples per step over all ranks: the staircase schedules of main.py:120-138 count samples seen."""
        return B * self.bucket.world_size()

    def _set_lr(self, B):
        self.opt.lr = learning_rate(self.step_count, self._global_batch(B))     # (reaches the device with the step's other scalars)

    def _gradients(self, xyz_f1_proj, xyz_f2_proj, q_gt, t_gt, decay):
        """zero grads -> forward -> loss -> backward: this rank's gradients, in the flat bucket."""
        self.bucket.zero()
        self.bucket.release()
        with torch.enable_grad():
            with tf_util.default_store(self.net.store), perm.default_perm_source(self.net.perms):
                out = pwclo_model.get_model_from_projection(xyz_f1_proj, xyz_f2_proj, True, decay)
            loss = pwclo_model.get_loss(*out[:8], q_gt, t_gt, self.w_x, self.w_q)
            loss.backward()
        self.bucket.collect()
        return loss.detach()

    def _body(self, xyz_f1_proj, xyz_f2_proj, q_gt, t_gt, decay):
        """gradients -> all-reduce -> Adam: everything of a step that runs on the GPU."""
        loss = self._gradients(xyz_f1_proj, xyz_f2_proj, q_gt, t_gt, decay)
        self.bucket.all_reduce_mean()                                    # the one collective of a training step
        self.opt.launch()                                                # (its scalars: opt.set_hyper(), outside a captured graph)
        return loss

    def step(self, xyz_f1_proj, xyz_f2_proj, q_gt, t_gt):
        """One optimisation step on this rank's batch; returns the (local) loss."""
        B = xyz_f1_proj.shape[0]
        self._set_lr(B)
        self.opt.set_hyper()
        self.net.perms.reshuffle()                                       # tf.random_shuffle draws per step
        loss = self._body(xyz_f1_proj, xyz_f2_proj, q_gt, t_gt, bn_decay(self.step_count, self._global_batch(B)))
        self.net.store.invalidate()                                      # folded / packed inference weights are stale
        self.step_count += 1
        return loss

    # -- the step as ONE hipGraph ----------------------------------------------
    def capture(self, xyz_f1_proj, xyz_f2_proj, q_gt, t_gt, warmup=3):
        """Record `_body` for this batch shape (needs `capturable=True`).  An eager step is ~4800 launches and is bound
        by the host; the replay is bound by the GPU.  The sample batch is used for the warm-up steps (they are real
        optimisation steps).  Learning rate and visiting orders change between replays through device tensors the
        graph reads; the BN decay is baked in, so `step_graph` re-captures when the schedule moves it (every 200 000
        samples, main.py:130-138)."""
        if not self.capturable:
            raise RuntimeError("Trainer(net, capturable=True) is needed to capture a training step")
        dev = self.net.device
        self._static = [torch.empty_like(x) for x in (xyz_f1_proj, xyz_f2_proj, q_gt, t_gt)]
        for s, x in zip(self._static, (xyz_f1_proj, xyz_f2_proj, q_gt, t_gt)):
            s.copy_(x)
        B = xyz_f1_proj.shape[0]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(*self._static)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._decay = bn_decay(self.step_count, self._global_batch(B))
        self._set_lr(B)
        self.net.perms.reshuffle()
        # With several ranks the collective stays OUTSIDE the graphs: [gradients] -> eager RCCL all-reduce of the flat
        # bucket -> [Adam] (two graphs); one rank: the whole step is one graph.
        self._split = self.bucket.has_collective()
        self._graph = torch.cuda.CUDAGraph()
        with graph_capture(self._graph):
            self._loss = (self._gradients if self._split else self._body)(*self._static, self._decay)
        self._graph_opt = None
        if self._split:
            self._graph_opt = torch.cuda.CUDAGraph()
            with graph_capture(self._graph_opt):
                self.opt.launch()
        # the graphs hold raw device pointers into the module-level index / decoded-order caches, which evict when they grow:
        # keep what they point at alive for as long as the graphs exist (model._cached_tensors, as the inference lanes do)
        from . import model
        self._keep = model._cached_tensors()
        self.net.store.invalidate()
        self.step_count += 1                                             # the capture itself does not run the step...
        self._replay()                                                   # ... this replay does
        return self

    def _replay(self):
        self.opt.set_hyper()
        self._graph.replay()
        if self._split:
            self.bucket.all_reduce_mean()
            self._graph_opt.replay()

    def step_graph(self, xyz_f1_proj, xyz_f2_proj, q_gt, t_gt):
        """`step` through the captured graph (same batch shape as `capture`)."""
        B = xyz_f1_proj.shape[0]
        if self._graph is None or bn_decay(self.step_count, self._global_batch(B)) != self._decay:
            self.capture(xyz_f1_proj, xyz_f2_proj, q_gt, t_gt, warmup=1 if self._graph is not None else 3)
            return self._loss
        for s, x in zip(self._static, (xyz_f1_proj, xyz_f2_proj, q_gt, t_gt)):
            s.copy_(x, non_blocking=True)
        self._set_lr(B)
        self.net.perms.reshuffle()
        self._replay()
        self.net.store.invalidate()
        self.step_count += 1
        return self._loss

    # -- checkpoints -----------------------------------------------------------
    def save(self, path, tf_bundle=False):
        """Variables (+ moving statistics), loss weights, step count and Adam moments -> `path` (.npz); with
        `tf_bundle` the variables, w_x and w_q are written as a TensorFlow bundle <path>.index/.data instead
        (what main.py:233 `saver.save` produces, minus the optimiser slots)."""
        state = {k: v.detach().cpu().numpy() for k, v in self.net.store.state_dict().items()}
        state["w_x"], state["w_q"] = self.w_x.detach().cpu().numpy(), self.w_q.detach().cpu().numpy()
        if tf_bundle:        # variables in the shapes TensorFlow stores them ([1,1,cin,cout] / [1,cin,cout] kernels): Saver.restore-able
            shapes = self.net.store.tf_shapes
            return tf_checkpoint.save_checkpoint(path, {k: v.reshape(shapes.get(k, v.shape)) for k, v in state.items()})
        opt = self.opt.state_dict()["state"]
        for i, p in enumerate(self.opt.params):
            state["adam_m/%d" % i] = opt[i]["exp_avg"].cpu().numpy()
            state["adam_v/%d" % i] = opt[i]["exp_avg_sq"].cpu().numpy()
            state["adam_t/%d" % i] = np.asarray(float(opt[i]["step"]))
        state["step_count"] = np.asarray(self.step_count)
        np.savez(path, **state)
        return sorted(state)

    def load(self, path):
        """Inverse of save(): `path` is the .npz file, or the prefix of a TensorFlow bundle (variables only)."""
        if os.path.exists(path + ".index"):
            tf_checkpoint.load_into(self.net.store, path)
            extra = tf_checkpoint.load_checkpoint(path, [k for k in ("w_x", "w_q")
                                                         if k in tf_checkpoint.read_index(path)[1]])
            state = {}
        else:
            state = dict(np.load(path if path.endswith(".npz") else path + ".npz"))
            extra = {k: state.pop(k) for k in ("w_x", "w_q") if k in state}
            adam = {k: state.pop(k) for k in list(state) if k.startswith("adam_")}
            self.step_count = int(state.pop("step_count", 0))
            self.net.store.load_state_dict(state)
            idx = [i for i in range(len(self.opt.params)) if "adam_m/%d" % i in adam]
            self.opt.load_state({i: adam["adam_m/%d" % i] for i in idx}, {i: adam["adam_v/%d" % i] for i in idx},
                                max([float(adam["adam_t/%d" % i]) for i in idx], default=0.0))
        with torch.no_grad():
            if "w_x" in extra:
                self.w_x.copy_(torch.as_tensor(extra["w_x"]).reshape(()))
            if "w_q" in extra:
                self.w_q.copy_(torch.as_tensor(extra["w_q"]).reshape(()))
        return self
