"""CPU: the C restatement (oracle/elo_oracle.c) against the committed golden
vectors, which were produced by the reference's own kernel bodies
(tests/golden/make_golden.py).  Bit-exact on all four outputs."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, expand_prefix, load_pkg
from oracle import grouping as G


def _call(c, blobs, impl="oracle", threads=1):
    k = c["name"]
    fn = G.fused_conv_random_k if c["op"] == "random" else G.fused_conv_select_k
    idx = blobs[k + "/idx_n2"]
    return fn(blobs[k + "/xyz1"], blobs[k + "/xyz2"], idx, blobs[k + "/random_hw"], c["H"], c["W"],
              idx.shape[1], c["window"][0], c["window"][1], c["K"], c["flag_copy"], c["distance"],
              c["stride"][0], c["stride"][1], impl=impl, threads=threads)


def test_oracle_matches_every_golden_case(golden_cases):
    meta, blobs = golden_cases
    assert len(meta) >= 50
    for c in meta:
        k = c["name"]
        sel, valid, indis, mask = _call(c, blobs)
        KT = c["window"][0] * c["window"][1]
        assert np.array_equal(sel, blobs[k + "/sel"]), k
        assert np.array_equal(mask, blobs[k + "/mask"]), k
        assert np.array_equal(valid, expand_prefix(blobs[k + "/n_valid"], KT)), k
        assert np.array_equal(indis, expand_prefix(blobs[k + "/n_indis"], KT)), k


def test_oracle_threaded_equals_scalar(golden_cases):
    meta, blobs = golden_cases
    for c in meta[::7]:
        a = _call(c, blobs, threads=1)
        b = _call(c, blobs, threads=4)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_kat1_reference_demo_values(golden_cases):
    """Known answers of the reference's __main__ demo with the identity permutation
    (fused_conv_random_k.py:100-127)."""
    meta, blobs = golden_cases
    by = {c["name"]: c for c in meta}
    sel, valid, indis, mask = _call(by["kat1_random"], blobs)
    assert not sel[0, 0].any() and not mask[0, 0].any()          # centre 0 is the zero point
    assert sel[0, 1, :4].tolist() == [[0, 0, 6], [0, 0, 1], [0, 0, 2], [0, 0, 3]]
    assert mask[0, 1, :, 0].tolist() == [1, 1, 1, 1, 0, 0, 0, 0]
    assert valid[0, 1, :, 0].tolist() == [1, 1, 1, 1, 0] == indis[0, 1, :, 0].tolist()
    sel, _, _, mask = _call(by["kat1_select"], blobs)
    assert sel[0, 1, :4].tolist() == [[0, 0, 1], [0, 0, 2], [0, 0, 3], [0, 0, 6]]
    sel, _, _, mask = _call(by["kat1_select_copy"], blobs)
    assert sel[0, 1].tolist() == [[0, 0, 1]] * 8 and mask[0, 1, :, 0].tolist() == [1] * 8


def _digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("i", range(4))
def test_oracle_large_digests(i):
    """Full 64x1800 / 128x2048 cases: digests of the reference outputs only."""
    with open(os.path.join(GOLDEN, "large_digests.json")) as f:
        d = json.load(f)[i]
    synth = load_pkg("synth")
    f1, f2 = synth.frame_pair(1, d["H"], d["W"], seed=d["seed"])
    idx = synth.hw_index(1, d["H"], d["W"])
    KT = d["window"][0] * d["window"][1]
    perm = np.random.default_rng(d["seed"]).permutation(KT).astype(np.int32)
    if _digest(f1, f2, idx, perm) != d["inputs_sha256"]:
        pytest.skip("numpy RNG stream differs from the authoring container")
    fn = G.fused_conv_random_k if d["op"] == "random" else G.fused_conv_select_k
    sel, valid, indis, mask = fn(f1, f2, idx, perm, d["H"], d["W"], idx.shape[1], d["window"][0],
                                 d["window"][1], d["K"], 0, d["distance"], 1, 1, threads=8)
    assert _digest(sel) == d["sel_sha256"]
    assert _digest(mask) == d["mask_sha256"]
    assert _digest(valid.sum(2).astype(np.int32), indis.sum(2).astype(np.int32)) == d["counts_sha256"]
