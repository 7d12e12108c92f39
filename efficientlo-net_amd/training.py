"""Training step for the hot path: forward (batch-statistics BN, dropout, fresh visiting orders) ->
get_loss -> backward -> ONE flat-bucket all-reduce over RCCL -> Adam.  Mirrors main.py:120-176,344-397
(the reference's single-GPU loop) with the data-parallel exchange of SURVEY.md section 5 added.
The full training driver (epochs, evaluation, checkpoint policy) is out of scope.
"""
import torch

from . import perm, pwclo_model, tf_util
from .distributed import FlatGradBucket

BASE_LEARNING_RATE, DECAY_STEP, DECAY_RATE = 0.001, 200000, 0.7          # main.py:46-51
BN_INIT_DECAY, BN_DECAY_DECAY_RATE, BN_DECAY_CLIP = 0.5, 0.5, 0.99       # main.py:62-65


def learning_rate(step, batch_size):
    """main.py:120-128: staircase exponential decay, floored at 1e-5."""
    return max(BASE_LEARNING_RATE * DECAY_RATE ** ((step * batch_size) // DECAY_STEP), 0.00001)


def bn_decay(step, batch_size):
    """main.py:130-138."""
    momentum = BN_INIT_DECAY * BN_DECAY_DECAY_RATE ** ((step * batch_size) // DECAY_STEP)
    return min(BN_DECAY_CLIP, 1 - momentum)


class Trainer:
    def __init__(self, net):
        self.net = net
        pwclo_model.create_variables(net.store)
        dev = net.device
        self.w_x = torch.nn.Parameter(torch.tensor(0.0, device=dev))     # main.py:151
        self.w_q = torch.nn.Parameter(torch.tensor(-2.5, device=dev))    # main.py:152
        self.params = net.store.parameters() + [self.w_x, self.w_q]      # 382 tensors, 899 134 values
        self.bucket = FlatGradBucket(self.params)
        self.opt = torch.optim.Adam(self.params, lr=BASE_LEARNING_RATE)  # main.py:174
        self.step_count = 0

    def step(self, xyz_f1_proj, xyz_f2_proj, q_gt, t_gt):
        """One optimisation step on this rank's batch; returns the (local) loss."""
        B = xyz_f1_proj.shape[0]
        for g in self.opt.param_groups:
            g["lr"] = learning_rate(self.step_count, B)
        self.net.perms.reshuffle()                                       # tf.random_shuffle draws per step
        self.bucket.zero()
        with torch.enable_grad():
            with tf_util.default_store(self.net.store), perm.default_perm_source(self.net.perms):
                out = pwclo_model.get_model_from_projection(xyz_f1_proj, xyz_f2_proj, True,
                                                            bn_decay(self.step_count, B))
            loss = pwclo_model.get_loss(*out[:8], q_gt, t_gt, self.w_x, self.w_q)
            loss.backward()
        self.bucket.all_reduce_mean()                                    # the one collective of a training step
        self.opt.step()
        self.net.store.invalidate()                                      # folded / packed inference weights are stale
        self.step_count += 1
        return loss.detach()
