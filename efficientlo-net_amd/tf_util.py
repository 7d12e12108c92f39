"""Layer wrappers with the reference's names and argument lists (utils/tf_util.py),
on PyTorch-ROCm: conv2d (:120-185), conv1d (:52-115), batch norm (:512-531,
tf.contrib.layers.batch_norm, eps 1e-3) and the variable helpers (:10-49).

Only what the hot path uses is here: 1x1 convolutions.  A 1x1 conv over an
NHWC tensor is a dense contraction (rows = B*N*K, Cin x Cout): it goes to
hipBLASLt through torch.addmm -- it is NOT one of the hand-written memory-bound
kernels.

Variables.  The reference creates variables by name under tf.variable_scope and
shares them between the two frames with scope.reuse_variables()
(pwclo_model.py:117,143).  Here a VariableStore plays the TF graph's variable
collection: `variable_scope(name)` pushes a name, and a layer does
get-or-create on '<scope>/weights', '<scope>/biases',
'<scope>/bn/{gamma,beta,moving_mean,moving_variance}' -- exactly the names of
the shipped checkpoint index, so the Siamese sharing
falls out and a TF checkpoint can be imported by name.

Inference (is_training False) folds BN's moving statistics and the bias into
the GEMM: y = relu(x @ W' + b') is ONE hipBLASLt launch (bias + ReLU epilogue).  Folded weights
are cached per scope until VariableStore.invalidate().
"""
import contextlib
import math
import os

import torch
import torch.nn.functional as Fnn

from . import _ops, tuning

# tuning "train_kernels" (ELO_TRAIN_KERNELS; False: torch's batch norm + GEMM weight gradients, the comparison) is read at the point of use
BN_EPS = 1e-3            # tf.contrib.layers.batch_norm default (utils/tf_util.py:526-531)
_DEFAULT_BN_DECAY = 0.9  # utils/tf_util.py:525


class VariableStore:
    """Named parameters + BN moving statistics + the current scope stack."""

    def __init__(self, device="cuda", seed=0):
        self.device = torch.device(device)
        self.params = {}          # name -> torch.nn.Parameter
        self.buffers = {}         # name -> tensor (moving_mean / moving_variance)
        self.tf_shapes = {}       # name -> shape as the TF checkpoint stores it
        self._scope = []
        self._folded = {}
        self.generation = 0       # bumped by invalidate(): a captured hipGraph holds pointers to the folded / packed tensors
        self._gen = torch.Generator(device="cpu")
        self._gen.manual_seed(seed)

    # -- scopes -----------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name):
        self._scope.append(name)
        try:
            yield _Scope(self)
        finally:
            self._scope.pop()

    def full_name(self, name):
        return "/".join(self._scope + [name])

    # -- variables ----------------------------------------------------------
    def get_variable(self, name, shape, init, tf_shape=None, trainable=True):
        """tf.get_variable: create on first use, reuse afterwards (shape-checked)."""
        full = self.full_name(name)
        table = self.params if trainable else self.buffers
        if full in table:
            if tuple(table[full].shape) != tuple(shape):
                raise ValueError("variable %s exists with shape %s, requested %s"
                                 % (full, tuple(table[full].shape), tuple(shape)))
            return table[full]
        value = init(shape).to(self.device)
        table[full] = torch.nn.Parameter(value) if trainable else value
        self.tf_shapes[full] = tuple(tf_shape if tf_shape is not None else shape)
        return table[full]

    def xavier(self, fan_in, fan_out):
        """tf.contrib.layers.xavier_initializer(): uniform(+-sqrt(6/(fan_in+fan_out)))."""
        limit = math.sqrt(6.0 / (fan_in + fan_out))
        return lambda shape: (torch.rand(shape, generator=self._gen, dtype=torch.float32) * 2 - 1) * limit

    def parameters(self):
        return list(self.params.values())

    def state_dict(self):
        out = {k: v.detach() for k, v in self.params.items()}
        out.update(self.buffers)
        return out

    def load_state_dict(self, state):
        with torch.no_grad():
            for k, v in state.items():
                target = self.params.get(k, self.buffers.get(k))
                if target is None:
                    raise KeyError("unknown variable " + k)
                target.copy_(torch.as_tensor(v).reshape(target.shape))
        self.invalidate()

    def invalidate(self):
        """Drop folded inference weights (call after any parameter update).  Graphs captured before this call read the
        dropped tensors' memory: PWCLONet compares `generation` and refuses to replay them."""
        self._folded.clear()
        self.generation += 1

    # -- inference folding --------------------------------------------------
    def folded(self, scope_name, W, b, bn):
        key = scope_name
        hit = self._folded.get(key)
        if hit is None:
            with torch.no_grad():
                if bn is None:
                    hit = (W.detach().contiguous(), b.detach().contiguous())
                else:
                    gamma, beta, mean, var = bn
                    s = gamma / torch.sqrt(var + BN_EPS)
                    hit = ((W * s).contiguous(), ((b - mean) * s + beta).contiguous())
            self._folded[key] = hit
        return hit


class _Scope:
    def __init__(self, store):
        self.store = store

    def reuse_variables(self):
        """No-op: variables are always get-or-create (pwclo_model.py:143)."""


_current = []


@contextlib.contextmanager
def default_store(store):
    """Make `store` the variable collection the layer functions below write to."""
    _current.append(store)
    try:
        yield store
    finally:
        _current.pop()


def get_store():
    if not _current:
        raise RuntimeError("no VariableStore is active: wrap the model in `with tf_util.default_store(store):`")
    return _current[-1]


def variable_scope(name):
    return get_store().variable_scope(name)


def scope_name():
    """Full name of the active variable scope, e.g. 'sa1/layer0'."""
    return "/".join(get_store()._scope)


def relu(x):
    return torch.relu(x)


def dense_variables(scope, cin, num_output_channels, tf_kernel_dims=(1, 1), bn=True):
    """get-or-create the variables of one conv layer; returns (full scope name, W, b, bn_vars|None)."""
    store = get_store()
    with store.variable_scope(scope):
        name = store.full_name("")[:-1]
        W = store.get_variable("weights", (cin, num_output_channels), store.xavier(cin, num_output_channels),
                               tf_shape=tuple(tf_kernel_dims) + (cin, num_output_channels))
        b = store.get_variable("biases", (num_output_channels,), torch.zeros)
        bn_vars = None
        if bn:
            with store.variable_scope("bn"):
                gamma = store.get_variable("gamma", (num_output_channels,), torch.ones)
                beta = store.get_variable("beta", (num_output_channels,), torch.zeros)
                mean = store.get_variable("moving_mean", (num_output_channels,), torch.zeros, trainable=False)
                var = store.get_variable("moving_variance", (num_output_channels,), torch.ones, trainable=False)
            bn_vars = (gamma, beta, mean, var)
    return name, W, b, bn_vars


def folded_variables(scope, cin, num_output_channels, tf_kernel_dims=(1, 1), bn=True):
    """Inference weights of one conv layer with BN's moving statistics and the bias folded in."""
    name, W, b, bn_vars = dense_variables(scope, cin, num_output_channels, tf_kernel_dims, bn)
    return get_store().folded(name, W, b, bn_vars)


_bn_groups = 1


@contextlib.contextmanager
def bn_groups(n):
    """`with tf_util.bn_groups(2):` -- inside, a TRAINING batch-norm layer treats its batch as n equal blocks with their own batch
    statistics, and updates the moving averages once per block in order: what calling the layer once per block with shared variables
    does (the reference's Siamese pyramid, pwclo_model.py:117-143), in one set of launches."""
    global _bn_groups
    prev, _bn_groups = _bn_groups, int(n)
    try:
        yield
    finally:
        _bn_groups = prev


def _dense(inputs, num_output_channels, scope, tf_kernel_dims, activation_fn, bn, bn_decay, is_training):
    cin = inputs.shape[-1]
    name, W, b, bn_vars = dense_variables(scope, cin, num_output_channels, tf_kernel_dims, bn)
    x2 = inputs.reshape(-1, cin)
    training = bool(is_training) if is_training is not None else False
    if training:                                 # batch statistics, autograd (the caller's is_training decides, not the autograd mode)
        decay = _DEFAULT_BN_DECAY if bn_decay is None else float(bn_decay)
        if (bn and tuning.get("train_kernels") and (activation_fn is relu or activation_fn is None)
                and _ops.dense_bn_supported(x2, num_output_channels)):
            # the row reductions of conv -> batch norm -> ReLU on hand-written kernels (csrc/elo_train.hip)
            y = _ops.dense_bn(x2, W, b, bn_vars[0], bn_vars[1], bn_vars[2], bn_vars[3], 1.0 - decay, BN_EPS,
                              activation_fn is relu, groups=_bn_groups)
            return y.reshape(inputs.shape[:-1] + (num_output_channels,))
        y = torch.addmm(b, x2, W)
        if bn:
            y = torch.cat([Fnn.batch_norm(part, bn_vars[2], bn_vars[3], bn_vars[0], bn_vars[1], training=training,
                                          momentum=1.0 - decay, eps=BN_EPS) for part in y.chunk(_bn_groups)]) if _bn_groups > 1 else \
                Fnn.batch_norm(y, bn_vars[2], bn_vars[3], bn_vars[0], bn_vars[1], training=training, momentum=1.0 - decay, eps=BN_EPS)
        if activation_fn is not None:
            y = activation_fn(y)
    else:                                        # inference: BN + bias folded into the GEMM
        Wf, bf = get_store().folded(name, W, b, bn_vars)
        if activation_fn is relu:
            y = torch._addmm_activation(bf, x2, Wf)      # bias + ReLU in the hipBLASLt epilogue: one kernel
        else:
            y = torch.addmm(bf, x2, Wf)
            if activation_fn is not None:
                y = activation_fn(y)
    return y.reshape(inputs.shape[:-1] + (num_output_channels,))


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=[1, 1], padding='SAME', data_format='NHWC',
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn=relu, bn=False, bn_decay=None,
           is_training=None):
    """utils/tf_util.py:120-185 for the 1x1 / stride-1 / NHWC case the model uses."""
    if list(kernel_size) != [1, 1] or list(stride) != [1, 1] or data_format != 'NHWC':
        raise NotImplementedError("only 1x1, stride 1, NHWC convolutions are on the hot path")
    if not use_xavier or weight_decay is not None:
        raise NotImplementedError("the model uses xavier init without weight decay")
    return _dense(inputs, num_output_channels, scope, (1, 1), activation_fn, bn, bn_decay, is_training)


def conv1d(inputs, num_output_channels, kernel_size, scope, stride=1, padding='SAME', data_format='NHWC',
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn=relu, bn=False, bn_decay=None,
           is_training=None):
    """utils/tf_util.py:52-115 for kernel_size 1."""
    if kernel_size != 1 or stride != 1 or data_format != 'NHWC':
        raise NotImplementedError("only kernel 1, stride 1, NHWC 1-D convolutions are on the hot path")
    return _dense(inputs, num_output_channels, scope, (1,), activation_fn, bn, bn_decay, is_training)
