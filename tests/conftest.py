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


# ---- fp32-noise gates ------------------------------------------------------------------------------------------------
# `full` / `sparse` evaluate the reference's fp32 finite difference (f(X + d) - f(X)) / 1e-4 itself, so their distance to the
# fp64 evaluation is rounding noise of the reference's own class.  The HARD gate is principled: our error, in units of the
# reference's own fp32 error for the same case, stays below RATIO_CEILING = 2 (two draws from one noise class; BASELINE.md
# section 3's `err(build) <= err(reference fp32)` itself is asserted on WHOLE matrices of the BASELINE configs, where the
# maxima are over 1e5 entries and the ratio is stable: tests/test_gpu_round4.py, tests/golden/fp32_whole_matrix.npz).
# The measured ratios are also kept in tests/golden/fp32_noise_ratios.json (a GPU run with LT_RECORD_RATIOS=1 writes
# gpurun_out/fp32_noise_ratios.json) -- INFORMATIONAL since round 4: a value that moved by more than 10 % against the record
# is reported as a warning (a toolchain bump or a legitimate reorder moves them), never a failure.
RATIO_FILE = os.path.join(GOLDEN, "fp32_noise_ratios.json")
RATIO_CEILING = 2.0
_recorded = {}


def _ratio_table():
    import json
    try:
        with open(RATIO_FILE) as fh:
            return json.load(fh)
    except FileNotFoundError:
        return {}


def noise_gate(key, measured, ceiling=RATIO_CEILING):
    """measured = our error expressed in units of the reference's own fp32 error for the same case (or, with an explicit
    ceiling, any quantity with a bound of its own)."""
    import warnings
    measured = float(measured)
    assert ceiling is None or measured <= ceiling, f"{key}: {measured:.4f} exceeds its bound {ceiling}"
    if os.environ.get("LT_RECORD_RATIOS"):
        _recorded[key] = round(measured, 6)
        return
    rec = _ratio_table().get(key)
    if rec is not None and measured > rec * 1.10 + 1e-5:
        warnings.warn(f"{key}: measured {measured:.6f}, recorded {rec:.6f} (informational: inside its bound {ceiling})")


def pytest_sessionfinish(session, exitstatus):
    if os.environ.get("LT_RECORD_RATIOS") and _recorded:
        import json
        out = os.path.join(REPO, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        merged = _ratio_table()
        merged.update(_recorded)
        with open(os.path.join(out, "fp32_noise_ratios.json"), "w") as fh:
            json.dump(dict(sorted(merged.items())), fh, indent=1)
