"""CPU: host-side logic that needs no GPU -- variable table vs the shipped checkpoint index,
pyramid sizes, the C-ABI library exporting every declared symbol, loud failure on CPU tensors,
the synthetic generator, the visiting-order source."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, load_pkg


def test_variable_table_matches_checkpoint_index():
    """382 trainable tensors / 899 134 values incl. the two loss scalars."""
    with open(os.path.join(GOLDEN, "ckpt_index_shapes.json")) as f:
        ckpt = json.load(f)
    pm, tf_util = load_pkg("pwclo_model"), load_pkg("tf_util")
    store = pm.create_variables(tf_util.VariableStore("cpu", seed=0))
    mine = dict(store.tf_shapes)
    theirs = {k: tuple(v) for k, v in ckpt.items() if k not in ("Variable", "w_x", "w_q")}
    assert set(mine) == set(theirs), (sorted(set(mine) ^ set(theirs))[:8])
    for k in mine:
        assert tuple(mine[k]) == theirs[k], k
    n_train = sum(int(np.prod(p.shape)) for p in store.params.values())
    assert len(store.params) + 2 == 382 and n_train + 2 == 899134
    assert sum(int(np.prod(b.shape)) for b in store.buffers.values()) == 14912


def test_pyramid_sizes():
    pm = load_pkg("pwclo_model")
    assert pm.pyramid_sizes(64, 1800) == ([64, 64, 16, 8, 4, 4], [1800, 1800, 225, 113, 57, 29])
    assert pm.pyramid_sizes(128, 2048) == ([128, 128, 32, 16, 8, 8], [2048, 2048, 256, 128, 64, 32])


def test_library_exports_every_declared_symbol():
    """Every `int elo_*(` / `long elo_*(` / `const char *elo_*(` prototype in include/elo.h resolves in libelo_hip.so,
    and the ctypes table binds exactly that set (no compute calls: there is no GPU here)."""
    L = load_pkg("_lib")
    header = open(os.path.join(ROOT, "include", "elo.h")).read()
    declared = set(re.findall(r"^(?:int|long|const char \*)\s*(elo_\w+)\s*\(", header, flags=re.M))
    assert declared == {name for name, _, _ in L.SYMBOLS}
    lib = L.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.elo_abi_version() == L.ABI_VERSION == 26
    assert lib.elo_last_error() == b"" or isinstance(lib.elo_last_error(), bytes)


def test_argument_validation_without_a_gpu():
    """The C entry points validate before launching: bad arguments come back as ELO_ERR_ARG/-LIMIT
    with the reference op's wording, without touching a device."""
    L = load_pkg("_lib")
    lib = L.lib()
    a = L.GroupArgs(1, 4, 8, 4, 8, 0, 3, 5, 4, 0, 1.0, 1, 1, None, None, None, None, None, None, None, None)
    assert lib.elo_fused_conv_random_k(ctypes.byref(a), None) == -1
    assert b"positive npoints" in lib.elo_last_error()
    a.npoints, a.kernel_h, a.kernel_w = 2, 71, 71
    assert lib.elo_fused_conv_select_k(ctypes.byref(a), None) == -2
    assert b"exceeds 5000" in lib.elo_last_error()
    a.kernel_h, a.kernel_w = 3, 19                     # kernel_w/2 = 9 > W2 = 8
    assert lib.elo_fused_conv_select_k(ctypes.byref(a), None) == -2
    a.kernel_w = 5
    assert lib.elo_fused_conv_select_k(ctypes.byref(a), None) == -1 and b"null tensor" in lib.elo_last_error()


def test_hot_path_fails_loudly_on_cpu_tensors():
    elo, ops = load_pkg(), load_pkg("_ops")
    x = torch.zeros(1, 4, 8, 3)
    idx = torch.zeros(1, 2, 2, dtype=torch.int32)
    perm = torch.arange(15, dtype=torch.int32)
    with pytest.raises(Exception, match="no CPU fallback"):
        elo.fused_conv_select_k(x, x, idx, perm, 4, 8, 2, 3, 5, 4, 0, 1.0, 1, 1)
    with pytest.raises(Exception, match="no CPU fallback"):
        ops.masked_maxpool(torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "efficientlo-net_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text and "libelo_oracle" not in text, fn


def test_synthetic_scene_properties():
    synth = load_pkg("synth")
    f1, f2 = synth.frame_pair(2, 64, 1800, seed=3)
    assert f1.shape == f2.shape == (2, 64, 1800, 3) and f1.dtype == np.float32
    empty = (f1 == 0).all(-1)
    assert 0.03 < empty.mean() < 0.6
    assert (np.hypot(f1[..., 0], f1[..., 1]) <= 35.0 + 1e-3).all()          # 35 m crop, model_util.py:380-383
    g1, _ = synth.frame_pair(2, 64, 1800, seed=3)
    assert np.array_equal(f1, g1)                                           # seeded
    idx = synth.hw_index(2, 4, 5)
    assert idx.shape == (2, 20, 2) and idx[0, 7].tolist() == [1, 2]
    assert synth.strided_index(1, 2, 3, 4, 8)[0].tolist() == [[0, 0], [0, 8], [0, 16], [4, 0], [4, 8], [4, 16]]


def test_perm_source_fixed_and_reshuffle():
    perm = load_pkg("perm")
    src = perm.PermSource(seed=1)
    a = src.get("s", "t", 45, "cpu")
    assert sorted(a.tolist()) == list(range(45)) and a.dtype == torch.int32
    assert src.get("s", "t", 45, "cpu") is a                                # fixed: same buffer
    before, ptr = a.clone(), a.data_ptr()
    src.reshuffle()
    assert a.data_ptr() == ptr and sorted(a.tolist()) == list(range(45)) and not torch.equal(a, before)
    hooked = perm.PermSource(fn=lambda scope, tag, KT: np.arange(KT)[::-1])
    assert hooked.get("x", "y", 5, "cpu").tolist() == [4, 3, 2, 1, 0]


def test_torch_library_registration_schema_and_shape_function():
    """torch.ops.elo.fused_conv_{random,select}_k: same attributes/inputs/outputs as the TF REGISTER_OP
    (fused_conv.cpp:15-63); GPU-only like REGISTER_KERNEL_BUILDER(DEVICE_GPU) (:176)."""
    load_pkg("torch_ops")
    for name in ("fused_conv_random_k", "fused_conv_select_k"):
        op = getattr(torch.ops.elo, name)
        schema = str(op.default._schema)
        for arg in ("xyz1", "xyz2", "idx_n2", "random_hw", "int H", "int W", "int npoints", "int kernel_size_H",
                    "int kernel_size_W", "int K", "int flag_copy", "float distance", "int stride_h", "int stride_w"):
            assert arg in schema, (name, arg)
        x = torch.zeros(2, 4, 8, 3, device="meta")
        idx = torch.zeros(2, 5, 2, dtype=torch.int32, device="meta")
        perm = torch.zeros(15, dtype=torch.int32, device="meta")
        out = op(x, x, idx, perm, 4, 8, 5, 3, 5, 6, 0, 1.0, 1, 1)
        assert [tuple(t.shape) for t in out] == [(2, 5, 6, 3), (2, 5, 15, 1), (2, 5, 15, 1), (2, 5, 6, 1)]
        assert out[0].dtype == torch.int32 and out[3].dtype == torch.float32
        with pytest.raises(NotImplementedError):
            op(torch.zeros(1, 4, 8, 3), torch.zeros(1, 4, 8, 3), torch.zeros(1, 2, 2, dtype=torch.int32),
               torch.zeros(15, dtype=torch.int32), 4, 8, 2, 3, 5, 4, 0, 1.0, 1, 1)


def test_tf_bundle_reader_round_trip_and_crc(tmp_path):
    """tf_checkpoint: crc32c known answer, writer -> reader round trip with every crc verified (table blocks and
    tensor bytes), scalars / int tensors / multi-block tables, VariableStore restore by TF variable name."""
    tc = load_pkg("tf_checkpoint")
    assert tc.crc32c(b"123456789") == 0xE3069283                     # RFC 3720 check value
    rng = np.random.default_rng(0)
    tensors = {"sa1/layer0/conv0/weights": rng.standard_normal((1, 1, 6, 8)).astype(np.float32),
               "w_q": np.array(-2.5, np.float32), "Variable": np.array(7, np.int32),
               "ids": np.arange(5, dtype=np.int64)}
    tensors.update({"v/%03d" % i: rng.standard_normal((i % 7 + 1, 3)).astype(np.float32) for i in range(40)})
    prefix = str(tmp_path / "model.ckpt")
    tc.save_checkpoint(prefix, tensors)
    header, entries = tc.read_index(prefix, verify=True)
    assert header["num_shards"] == 1 and set(entries) == set(tensors)
    back = tc.load_checkpoint(prefix, verify=True)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v)
    assert "Variable" not in tc.model_variables(entries)
    blob = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    blob[3] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(blob))
    with pytest.raises(ValueError, match="crc32c"):
        tc.load_checkpoint(prefix, verify=True)

    tf_util, pm = load_pkg("tf_util"), load_pkg("pwclo_model")
    a, b = tf_util.VariableStore("cpu", seed=1), tf_util.VariableStore("cpu", seed=2)
    pm.create_variables(a), pm.create_variables(b)
    state = {k: v.cpu().numpy() for k, v in a.state_dict().items()}
    tc.save_checkpoint(str(tmp_path / "full.ckpt"), state)
    loaded = tc.load_into(b, str(tmp_path / "full.ckpt"), verify=False)
    assert len(loaded) == len(state)
    for k, v in b.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), state[k])


@pytest.mark.skipif(not os.path.exists("/root/reference/pretrained_model/pretrained_model.ckpt.index"),
                    reason="the reference checkout is only present in the authoring container")
def test_tf_bundle_reader_parses_the_reference_index():
    """The reference ships the INDEX of its checkpoint (no data shard): every table block passes its crc32c, the
    entries tile the data shard without gaps, and names/shapes equal the committed fixture."""
    tc = load_pkg("tf_checkpoint")
    header, entries = tc.read_index("/root/reference/pretrained_model/pretrained_model.ckpt", verify=True)
    assert header["num_shards"] == 1 and len(entries) == 1327
    spans = sorted((e.offset, e.size) for e in entries.values())
    assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:]))
    with open(os.path.join(GOLDEN, "ckpt_index_shapes.json")) as f:
        fixture = json.load(f)
    model = tc.model_variables(entries) + ["Variable"]
    assert {k: list(entries[k].shape) for k in model} == fixture
    for e in entries.values():
        assert e.size == int(np.prod(e.shape, dtype=np.int64)) * np.dtype(tc.DTYPES[e.dtype]).itemsize
    with pytest.raises(FileNotFoundError, match="ships only the .index"):
        tc.load_checkpoint("/root/reference/pretrained_model/pretrained_model.ckpt", ["w_x"])


def test_data_augmentation_is_a_clipped_rigid_transform():
    """main.py:259-297: rotation part orthonormal with det 1, angles and offsets inside the reference's clip bounds."""
    tr = load_pkg("training")
    rng = np.random.default_rng(3)
    for _ in range(200):
        T = tr.data_augmentation(rng)
        R, t = T[:3, :3], T[:3, 3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
        assert np.array_equal(T[3], [0, 0, 0, 1])
        assert abs(t[0]) <= 1.0 and abs(t[1]) <= 0.2 and abs(t[2]) <= 0.15
        angle = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
        assert angle <= (0.02 + 0.02 + 0.1) * np.pi / 4 + 1e-9
    # first-order check of the composition order Rx.Ry.Rz against an explicit product
    T = tr.data_augmentation(np.random.default_rng(0))
    r = np.random.default_rng(0)
    draw = lambda s, b: float(np.clip(s * r.standard_normal(), -b, b))
    ax, ay, az = draw(0.01, 0.02) * np.pi / 4, draw(0.01, 0.02) * np.pi / 4, draw(0.05, 0.1) * np.pi / 4
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    assert np.allclose(T[:3, :3], Rx @ Ry @ Rz, atol=1e-15)


def test_trainer_tf_bundle_has_the_reference_checkpoint_shapes(tmp_path):
    """Trainer.save(tf_bundle=True) writes every variable in the shape TensorFlow stores it (1x1 kernels as
    [1,1,cin,cout], conv1d kernels as [1,cin,cout]): the written index equals the reference checkpoint's index fixture
    name for name and shape for shape (a TF Saver.restore checks exactly that), and loads back bit for bit."""
    model, training, tc = load_pkg("model"), load_pkg("training"), load_pkg("tf_checkpoint")
    net = model.PWCLONet("cpu", seed=3)
    tr = training.Trainer(net)
    prefix = str(tmp_path / "model.ckpt")
    tr.save(prefix, tf_bundle=True)
    _, entries = tc.read_index(prefix, verify=True)
    with open(os.path.join(GOLDEN, "ckpt_index_shapes.json")) as f:
        fixture = {k: v for k, v in json.load(f).items() if k != "Variable"}          # the global step is the trainer's own
    assert {k: list(e.shape) for k, e in entries.items()} == fixture
    before = {k: v.clone() for k, v in net.store.state_dict().items()}
    with torch.no_grad():
        for p in net.store.parameters():
            p.add_(1.0)
    training.Trainer(net).load(prefix)
    for k, v in net.store.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_store_and_perm_generations_track_invalidation():
    """What a captured inference graph depends on is versioned: VariableStore.invalidate() (checkpoint load, training
    step) and PermSource.reshuffle() bump a generation that PWCLONet compares before every replay."""
    tf_util, perm = load_pkg("tf_util"), load_pkg("perm")
    store, src = tf_util.VariableStore("cpu"), perm.PermSource(seed=1)
    g0, p0 = store.generation, src.generation
    store.invalidate()
    src.get("s", "t", 15, "cpu")
    src.reshuffle()
    assert store.generation == g0 + 1 and src.generation == p0 + 1
    store.load_state_dict({})
    assert store.generation == g0 + 2


def test_the_regime_switches_of_a_forward_are_tuning_values():
    """Which levels of a forward merge their cost volume with the set-upconvs (tile kernels' heterogeneous launch), from how many rows
    a stage takes the register-resident kernels, and the two round-5 rides, are fields of tuning.snapshot() -- hashed into every
    captured graph -- with the measured defaults (profiles/r05_batch1_regimes.txt, r05_small_tile_units.txt, r05_sv_ride.txt)."""
    tuning, pu, fused = load_pkg("tuning"), load_pkg("pointnet_util"), load_pkg("fused")
    for name, default in (("merge_points", 8192), ("merge_level_points", 2000), ("sv_ride", True), ("chain_pair", True)):
        assert tuning.get(name) == default, name
    # a 64 x 1800 pair at batch 1: l2 (228 centres) and l1 (904) merge, l0 (3600) does not; from batch 3 on nothing merges
    assert [pu.merge_branches(False, 3600, n) for n in (228, 904, 3600)] == [True, True, False]
    assert not pu.merge_branches(False, 3 * 3600, 3 * 228) and not pu.merge_branches(True, 3600, 228)
    with tuning.override(merge_level_points=10 ** 9):
        assert pu.merge_branches(False, 3600, 3600)
    # stage 1 of the l0 cost volume (3600 x 6 rows) takes the pre-pass + chain kernel below the throughput batch, stage 2 does not
    assert fused._prepass_rows(1, 1) <= 3600 * 6 < fused._prepass_rows(2, 1) and fused._prepass_rows(1, 8) == fused._prepass_rows(2, 8) == 8192
    before = tuning.digest()
    with tuning.override(sv_ride=False):
        assert tuning.digest() != before
    assert tuning.digest() == before


def test_kitti_density_profile_of_the_synthetic_scene():
    """synth.range_image(profile="kitti"): the density of a projected HDL-64 scan after the 35 m crop (kitti_dataset.py:38-103,
    model_util.py:380-383) -- about half of the grid valid, whole dead beam rows, a sector without returns -- and frame_pair's starved
    last batch element; the default ("dense") scene is unchanged by the new arguments."""
    synth = load_pkg("synth")
    f1, f2 = synth.frame_pair(3, 64, 1800, seed=40, profile="kitti")
    valid = (f1 != 0).any(-1)
    assert 0.45 < valid[0].mean() < 0.65 and 0.45 < valid[1].mean() < 0.65
    assert (valid[0].sum(1) == 0).sum() >= 2                       # dead beam rows
    assert (valid[0].sum(0) == 0).sum() >= 30                      # the 7-degree sector: 35 columns of 1800
    assert 1 <= valid[2].sum() <= 6 and 1 <= (f2[2] != 0).any(-1).sum() <= 6       # the starved element
    r = np.hypot(f1[0][..., 0], f1[0][..., 1])
    assert r.max() <= 35.0 + 1e-3
    near, far = valid[0][r < 10].mean() if (r < 10).any() else 1.0, valid[0][(r > 20)].mean()
    g1, _ = synth.frame_pair(1, 64, 1800, seed=40, profile="kitti", starved=True)
    assert (g1[0] != 0).any(-1).sum() <= 6
    d1, d2 = synth.frame_pair(2, 64, 1800, seed=40)
    assert (d1 != 0).any(-1).mean() > 0.93 and (d1[1] != 0).any(-1).sum() > 100000     # dense default: nobody starves
    with pytest.raises(ValueError):
        synth.range_image(profile="nope")


def test_bench_ring_helpers_clone_tensors_and_count_bytes():
    """bench._time_ring's bookkeeping (the COLD roofline reading): tensor arguments are cloned to new addresses, anything else is
    shared, and the footprint counts every distinct storage once."""
    import bench
    a, b = torch.zeros(1000), torch.zeros(10, 10)
    view = a[:10]
    args = (a, view, [b, 3], {"k": b, "s": "x"})
    assert bench._footprint(args, {}, None) == a.numel() * 4 + b.numel() * 4          # `view` shares a's storage, b counted once
    cl = bench._clone_tensors(args)
    assert cl[0].data_ptr() != a.data_ptr() and cl[2][0].data_ptr() != b.data_ptr() and cl[2][1] == 3 and cl[3]["s"] == "x"
    assert torch.equal(cl[0], a) and isinstance(cl[2], list) and isinstance(cl, tuple)
    assert bench.LLC_BYTES == 256 * 1024 * 1024


def test_bn_groups_on_the_torch_path_is_one_call_per_block():
    """tf_util.bn_groups(2) (the Siamese training batch) on the torch comparison path (CPU: no kernels): ONE call on the 2B batch equals the
    layer called once per frame with shared variables -- outputs, the moving averages (frame 1's update, then frame 2's) and the
    gradients -- the contract the grouped kernels of csrc/elo_train.hip are tested against on the GPU."""
    tf_util = load_pkg("tf_util")
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(2, 300, 1, 6, generator=g) * torch.tensor([1.0, 3.0]).view(2, 1, 1, 1) + torch.tensor([0.0, 1.5]).view(2, 1, 1, 1)
    gy = torch.randn(600, 8, generator=g)
    res = []
    for joint in (False, True):
        store = tf_util.VariableStore("cpu", seed=3)
        x = x0.clone().requires_grad_(True)
        layer = lambda inp: tf_util.conv2d(inp, 8, [1, 1], scope="layer", bn=True, is_training=True, bn_decay=0.7, activation_fn=tf_util.relu)
        with tf_util.default_store(store):
            if joint:
                with tf_util.bn_groups(2):
                    y = layer(x.reshape(1, 600, 1, 6)).reshape(600, 8)
            else:
                y = torch.cat([layer(x[0:1]), layer(x[1:2])], 1).reshape(600, 8)
            (y * gy).sum().backward()
        P = store.params
        res.append([y.detach(), store.buffers["layer/bn/moving_mean"], store.buffers["layer/bn/moving_variance"], x.grad,
                    P["layer/weights"].grad, P["layer/bn/gamma"].grad, P["layer/bn/beta"].grad])
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 1e-5 * (float(a.abs().max()) + 1e-3)
    # and the two blocks really were normalised apart: each half of the output has its own zero mean before the ReLU's cut shows
    assert abs(float(res[1][1].mean()) - float(res[0][1].mean())) < 1e-6


def test_the_lane_range_counter_is_stamped_into_the_fused_argument_blocks():
    """_lib.set_range_counter(ptr): every launch helper writes the lane's device word into the argument blocks that have the field
    (ABI 26); without it the blocks keep NULL = the process-wide counter."""
    L = load_pkg("_lib")
    a, b, m = L.SetconvArgs(), L.Cv1Args(), L.MlpArgs()
    L._stamp(a, b, None, m)
    assert not a.range_counter and not b.range_counter and not m.range_counter
    prev = L.set_range_counter(0x1000)
    try:
        L._stamp(a, b, None, m, L.Cv2Args(), L.WeightGradArgs())
        assert a.range_counter == 0x1000 and b.range_counter == 0x1000 and m.range_counter == 0x1000
    finally:
        assert L.set_range_counter(prev) == 0x1000
