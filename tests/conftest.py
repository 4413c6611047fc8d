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


# ---- recorded fp32-noise gates -----------------------------------------------------------------------------------
# `full` / `sparse` evaluate the reference's fp32 finite difference (f(X + d) - f(X)) / 1e-4 itself, so their distance
# to the fp64 evaluation is rounding noise of the reference's own class -- not a quantity a tolerance can be derived
# for.  The kernels are deterministic (fixed-order chains, no atomics), so every test case's measured ratio
# |ours - ref64| / |ref32 - ref64| (and raw AUC / AP deviations) is a constant of the source tree: it is RECORDED in
# tests/golden/fp32_noise_ratios.json by a GPU run with LT_RECORD_RATIOS=1 (written to gpurun_out/, copied into
# tests/golden/ by hand) and every later run asserts `measured <= recorded * 1.10` -- plus a hard ceiling no recorded
# value may exceed.  A missing key fails: new cases must be recorded.
RATIO_FILE = os.path.join(GOLDEN, "fp32_noise_ratios.json")
RATIO_CEILING = 2.5     # sanity cap on any recorded value (measured over all cases: 0.33 .. 2.06; whole-matrix fixtures <= 1.0)
_recorded = {}


def _ratio_table():
    import json
    try:
        with open(RATIO_FILE) as fh:
            return json.load(fh)
    except FileNotFoundError:
        return {}


def noise_gate(key, measured, ceiling=RATIO_CEILING):
    """measured = our error expressed in units of the reference's own fp32 error for the same case."""
    measured = float(measured)
    assert ceiling is None or measured <= ceiling, f"{key}: {measured:.3f} x the reference's own fp32 error (ceiling {ceiling})"
    if os.environ.get("LT_RECORD_RATIOS"):
        _recorded[key] = round(measured, 6)
        return
    table = _ratio_table()
    assert key in table, f"{key}: no recorded value in {RATIO_FILE} (run the GPU suite once with LT_RECORD_RATIOS=1)"
    # (+ 1e-5: the table is rounded to 6 decimals, and two cases compare two of OUR outputs, ratios of a few 1e-6)
    assert measured <= table[key] * 1.10 + 1e-5, f"{key}: measured {measured:.6f}, recorded {table[key]:.6f} (+10 % allowed)"


def pytest_sessionfinish(session, exitstatus):
    if os.environ.get("LT_RECORD_RATIOS") and _recorded:
        import json
        out = os.path.join(REPO, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        merged = _ratio_table()
        merged.update(_recorded)
        with open(os.path.join(out, "fp32_noise_ratios.json"), "w") as fh:
            json.dump(dict(sorted(merged.items())), fh, indent=1)
