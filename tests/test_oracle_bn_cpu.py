"""CPU: the oracle's training-mode layer (oracle/ops_np.conv_train: batch statistics + moving-average update,
utils/tf_util.py:512-531) against torch's own batch norm on the CPU -- two independent statements of the same layer."""
import numpy as np
import torch

from oracle import ops_np as O


def test_conv_train_is_batch_norm_with_batch_statistics():
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1, (3, 40, 5, 7)).astype(np.float32)
    p = {"l/weights": rng.normal(0, 0.4, (7, 16)).astype(np.float32), "l/biases": rng.normal(0, 0.1, 16).astype(np.float32),
         "l/bn/gamma": rng.normal(1, 0.2, 16).astype(np.float32), "l/bn/beta": rng.normal(0, 0.2, 16).astype(np.float32),
         "l/bn/moving_mean": rng.normal(0, 0.3, 16).astype(np.float32), "l/bn/moving_variance": rng.uniform(0.5, 2, 16).astype(np.float32)}
    y, mm, mv = O.conv_train(p, "l", x, bn_decay=0.9)
    z = torch.from_numpy(x).reshape(-1, 7) @ torch.from_numpy(p["l/weights"]) + torch.from_numpy(p["l/biases"])
    rm, rv = torch.from_numpy(p["l/bn/moving_mean"]).clone(), torch.from_numpy(p["l/bn/moving_variance"]).clone()
    t = torch.relu(torch.nn.functional.batch_norm(z, rm, rv, torch.from_numpy(p["l/bn/gamma"]), torch.from_numpy(p["l/bn/beta"]),
                                                  training=True, momentum=0.1, eps=float(O.BN_EPS)))
    assert np.allclose(y.reshape(-1, 16), t.numpy(), atol=2e-5, rtol=1e-5)
    assert np.allclose(mm, rm.numpy(), atol=1e-6) and np.allclose(mv, rv.numpy(), atol=1e-6)
    # inference with the updated averages is the oracle's own conv()
    q = dict(p, **{"l/bn/moving_mean": mm, "l/bn/moving_variance": mv})
    assert O.conv(q, "l", x).shape == y.shape
