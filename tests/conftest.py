import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg(sub=None):
    """The package directory is named 'efficientlo-net_amd' (hyphen): import by name."""
    name = "efficientlo-net_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


@pytest.fixture(scope="session")
def elo():
    return load_pkg()


@pytest.fixture(scope="session")
def golden_cases():
    with open(os.path.join(GOLDEN, "grouping_cases.json")) as f:
        meta = json.load(f)
    blobs = np.load(os.path.join(GOLDEN, "grouping_cases.npz"))
    return meta, blobs


def expand_prefix(counts, KT):
    """valid_idx / valid_in_dis_idx are prefix-ones masks (Appendix A.2 of SURVEY.md):
    fixtures store the per-centre counts; rebuild the (B,N,KT,1) float masks."""
    ar = np.arange(KT, dtype=np.int32)[None, None, :]
    return (ar < counts[..., None]).astype(np.float32)[..., None]
