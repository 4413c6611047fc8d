"""The oracle (oracle/linkteller_oracle.py) pinned against vectors produced by the reference itself
(tests/golden/generate_golden.py).  CPU only."""
import ast

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import csr_from, golden_args, load_golden
from oracle import linkteller_oracle as O


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)   # the fixtures were generated single-threaded (fixed reduction order)
    yield
    torch.set_num_threads(n)


def test_normalizers_and_sparse_conversion():
    g = load_golden("normalizer.npz")
    for key in g["keys"]:
        a = csr_from(g, f"{key}.adj")
        for norm in ("FirstOrderGCN", "AugNormAdj"):
            with np.errstate(divide="ignore"):
                res = O.NORMALIZERS[norm](a)
            assert np.array_equal(res.row, g[f"{key}.{norm}.row"])
            assert np.array_equal(res.col, g[f"{key}.{norm}.col"])
            assert np.array_equal(res.data, g[f"{key}.{norm}.data"])          # float64, bit-exact
            t = O.to_torch_sparse(res)
            assert np.array_equal(t._indices().numpy(), g[f"{key}.{norm}.t_indices"])
            assert np.array_equal(t._values().numpy(), g[f"{key}.{norm}.t_values"])
            assert t._values().dtype == torch.float32 and t._indices().dtype == torch.int64


def _params(g, key, dtype=torch.float32, names=(("W1", "gc1.weight"), ("b1", "gc1.bias"), ("W2", "gc2.weight"), ("b2", "gc2.bias"))):
    return {k: torch.from_numpy(g[f"{key}.sd.{n}"]).to(dtype) for k, n in names}


@pytest.mark.parametrize("key", ["n64", "n600", "n200c7"])
def test_gcn_forward(forward_golden, key):
    g = forward_golden
    adj = O.to_torch_sparse(O.first_order_gcn(csr_from(g, f"{key}.adj")))
    x = torch.from_numpy(g[f"{key}.x"])
    out32 = O.gcn_forward(x, adj, _params(g, key)).numpy()
    assert np.array_equal(out32, g[f"{key}.logits32"])
    out64 = O.gcn_forward(x.double(), adj.double(), _params(g, key, torch.float64)).numpy()
    assert np.array_equal(out64, g[f"{key}.logits64"])


def test_gcn3_forward(forward_golden):
    g = forward_golden
    names = (("W1", "gc1.weight"), ("b1", "gc1.bias"), ("W2", "gc2.weight"), ("b2", "gc2.bias"),
             ("W3", "gc3.weight"), ("b3", "gc3.bias"))
    adj = O.to_torch_sparse(O.first_order_gcn(csr_from(g, "gcn3.adj")))
    x = torch.from_numpy(g["gcn3.x"])
    assert np.array_equal(O.gcn3_forward(x, adj, _params(g, "gcn3", names=names)).numpy(), g["gcn3.logits32"])
    assert np.array_equal(O.gcn3_forward(x.double(), adj.double(), _params(g, "gcn3", torch.float64, names)).numpy(),
                          g["gcn3.logits64"])


def test_sampler_all_combinations():
    g = load_golden("sampler.npz")
    a = csr_from(g, "adj")
    for tag in g["combos"]:
        ds, st, seed = str(tag).split(".")
        dataset = ds.replace("_", "/")
        np.random.seed(int(seed))
        (ex, nex), nodes = O.sample_subgraph_pairs(dataset, st, a, 24)
        assert np.array_equal(np.asarray(nodes), g[f"{tag}.nodes"])
        assert np.array_equal(np.asarray(ex, dtype=np.int64).reshape(-1, 2), g[f"{tag}.exist"])
        assert np.array_equal(np.asarray(nex, dtype=np.int64).reshape(-1, 2), g[f"{tag}.nonexist"])


@pytest.mark.parametrize("key", ["er300", "pl600", "pl600hi", "lap600", "rand400"])
def test_influence_loop_scores_metrics(influence_golden, key):
    g = influence_golden
    args = golden_args(g, key)
    a = csr_from(g, f"{key}.adj")
    served = csr_from(g, f"{key}.served") if f"{key}.served.n" in g else a
    adj = O.to_torch_sparse(O.NORMALIZERS[args["norm"]](served))
    x = torch.from_numpy(g[f"{key}.x"])
    np.random.seed(args["sample_seed"])
    (ex, nex), nodes = O.sample_subgraph_pairs(args["dataset"], args["sample_type"], a, args["n_test"])
    assert np.array_equal(nodes, g[f"{key}.ref32.test_nodes"])
    infl32 = O.influence_matrix(x, adj, _params(g, key), nodes, args["influence"])
    assert np.array_equal(infl32, g[f"{key}.ref32.influence_val"])            # bit-exact fp32 op sequence
    infl64 = O.influence_matrix(x.double(), adj.double(), _params(g, key, torch.float64), nodes, args["influence"])
    assert np.array_equal(infl64, g[f"{key}.ref64.influence_val"])
    ne, nn = O.pair_scores(infl32, nodes, ex, nex)
    assert np.array_equal(np.asarray(ne), g[f"{key}.ref32.norm_exist"])
    assert np.array_equal(np.asarray(nn), g[f"{key}.ref32.norm_nonexist"])
    m = O.attack_metrics(ne, nn)
    assert m["auc"] == float(g[f"{key}.ref32.auc"]) and m["ap"] == float(g[f"{key}.ref32.ap"])
    for k_ in ("fpr", "tpr", "thresholds", "precision", "recall"):
        assert np.array_equal(m[k_], g[f"{key}.ref32.{k_}"])
    assert np.array_equal(m["thresholds_2"], g[f"{key}.ref32.pr_thresholds"])
    assert np.array_equal(np.asarray(m["y"]), g[f"{key}.ref32.y"])
    fn = O.result_filename(args["dataset"], args["mode"], args["attack_mode"], args["sample_type"], args["n_test"],
                           args["sample_seed"], args["perturb_type"], args["epsilon"], args["noise_seed"])
    assert fn == str(g[f"{key}.ref32.filename"])
    # partial loop (the bounded cpu_baseline sample) fills exactly the requested rows
    part = O.influence_matrix(x, adj, _params(g, key), nodes, args["influence"], probe_range=range(2, 5))
    assert np.array_equal(part[2:5], infl32[2:5]) and not part[:2].any() and not part[5:].any()


def test_dp_adjacency_generators():
    g = load_golden("dp_adjacency.npz")
    a = csr_from(g, "adj")
    for perturb, eps in (("continuous", 5.0), ("continuous", 1.0), ("discrete", 4.0), ("discrete", 7.0)):
        fn = O.perturb_adj_continuous if perturb == "continuous" else O.perturb_adj_discrete
        res = sp.csr_matrix(fn(sp.csr_matrix(a), eps, 42))
        res.sort_indices()
        tag = f"{perturb}.eps{eps:g}"
        assert np.array_equal(res.indptr, g[f"{tag}.indptr"])
        assert np.array_equal(res.indices, g[f"{tag}.indices"])
        assert np.array_equal(np.asarray(res.data, dtype=np.float64), g[f"{tag}.data"])
