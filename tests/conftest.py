import ast
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def csr_from(g, prefix):
    n = int(g[prefix + ".n"])
    return sp.csr_matrix((g[prefix + ".data"], g[prefix + ".indices"], g[prefix + ".indptr"]), shape=(n, n))


def golden_args(g, key):
    return ast.literal_eval(str(g[f"{key}.args"]))


@pytest.fixture(scope="session")
def influence_golden():
    return load_golden("influence.npz")


@pytest.fixture(scope="session")
def forward_golden():
    return load_golden("forward.npz")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no HIP device is visible")
    return torch.device("cuda:0")
