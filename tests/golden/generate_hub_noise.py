#!/usr/bin/env python3
"""fp32 self-noise of the REFERENCE on the hub-of-many-segments case of tests/test_gpu_round3.py -> hub_noise.npz.

A 9 500-node graph whose node 0 is adjacent to every other node (a row of 9 499 entries), 100 probes x 100 observed nodes with the
hub observed: the pair (probe, hub) is the hub's ulp-quantised logit difference / 1e-4 in any fp32 evaluation.  The test gates the
HIP `full` mode's error against the reference's OWN fp32 error, both measured from the fp64 evaluation, so the right-hand side is
generated here from the imported reference (its ``get_gradient_eps_mat`` on torch's CPU kernels, one thread) and committed:

    probes, obs       the node lists (the test rebuilds the same graph / features / weights from the same seeds and checks them)
    ref32             the reference's fp32 scores, float32 [100, 100]
    ref64             the fp64 evaluation (oracle.RestrictedOracle, pinned to the reference by tests/test_oracle_golden.py)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_hub_noise.py        (about a minute on one thread)
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LT_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

from linkteller_amd import synth                # noqa: E402  our seeded generators (inputs only)
from oracle import linkteller_oracle as O       # noqa: E402
from generate_golden import fake_worker, make_args, quiet   # noqa: E402  (imports the reference modules too)
import attacker as ref_attacker                 # noqa: E402  /root/reference/attacker.py
from gcn.models import GCN                      # noqa: E402  /root/reference/gcn/models.py

torch.set_num_threads(1)


def hub_case():
    """The inputs of test_full_mode_on_a_hub_of_many_segments, from its seeds."""
    n = 9500
    rng = np.random.RandomState(4)
    r = rng.randint(1, n, 30000)
    c = rng.randint(1, n, 30000)
    keep = r != c
    rows = np.concatenate([np.zeros(n - 1, int), r[keep]])
    cols = np.concatenate([np.arange(1, n), c[keep]])
    a = sp.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.float32).tocsr()
    x = synth.twitch_like_features(n, 200, seed=6, density=0.03)
    w = synth.gcn_weights(200, 256, 2, seed=8)
    probes = np.concatenate([np.arange(1, 71), rng.choice(np.arange(200, n), 30, replace=False)]).astype(np.int64)
    obs = np.concatenate([[0], np.arange(1, 40), rng.choice(np.arange(200, n), 60, replace=False)]).astype(np.int64)
    return a, x, w, probes, obs


def main():
    a, x, w, probes, obs = hub_case()
    model = quiet(GCN, nfeat=x.shape[1], nhid=256, nclass=2, dropout=0.5)
    model.load_state_dict({"gc1.weight": torch.from_numpy(w["W1"]), "gc1.bias": torch.from_numpy(w["b1"]),
                           "gc2.weight": torch.from_numpy(w["W2"]), "gc2.bias": torch.from_numpy(w["b2"])})
    model.eval()
    wk = fake_worker(a, a, x, "FirstOrderGCN")
    atk = ref_attacker.Attacker(make_args(n_test=len(probes)), model, wk)
    ref32 = np.zeros((len(probes), len(obs)), dtype=np.float32)
    with torch.no_grad():
        for k, v in enumerate(probes):
            g = atk.get_gradient_eps_mat(int(v))                        # attacker.py:100-108, fp32
            ref32[k] = g[torch.as_tensor(obs)].norm(dim=1).numpy()      # attacker.py:229 without the per-pair .item()
    ref64 = O.RestrictedOracle(x, O.first_order_gcn(a), w).rows(probes, obs, 1e-4)
    e = np.abs(ref32.astype(np.float64) - ref64)
    print(f"hub case: max score {ref64.max():.4f}; reference fp32 error: row maxima rms {np.sqrt((e.max(axis=1) ** 2).mean()):.5f} "
          f"max {e.max():.5f}; hub column rms {np.sqrt((e[:, 0] ** 2).mean()):.5f}", flush=True)
    np.savez_compressed(os.path.join(HERE, "hub_noise.npz"), probes=probes, obs=obs, ref32=ref32, ref64=ref64)


if __name__ == "__main__":
    main()
