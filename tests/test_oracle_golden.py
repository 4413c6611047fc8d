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


def test_next_rows_baseline_balanced_gcn3():
    g = load_golden("next_rows.npz")
    a = csr_from(g, "adj")
    x = torch.from_numpy(g["x"])
    adj = O.to_torch_sparse(O.first_order_gcn(a))
    P = {k: torch.from_numpy(g[f"sd.{n}"]) for k, n in (("W1", "gc1.weight"), ("b1", "gc1.bias"), ("W2", "gc2.weight"), ("b2", "gc2.bias"))}
    for mode in ("baseline", "baseline-feat"):
        np.random.seed(42)
        (ex, nex), nodes = O.sample_subgraph_pairs("twitch/ES/RU", "unbalanced", a, 40)
        assert np.array_equal(nodes, g[f"{mode}.test_nodes"])
        vec = O.baseline_vectors(mode, x, adj, P, "twitch/ES/RU")
        ne, nn = O.baseline_attack(vec, nodes, ex, nex)
        assert np.array_equal(np.asarray(ne), g[f"{mode}.norm_exist"])
        assert np.array_equal(np.asarray(nn), g[f"{mode}.norm_nonexist"])
        m = O.attack_metrics(ne, nn)
        assert m["auc"] == float(g[f"{mode}.auc"]) and m["ap"] == float(g[f"{mode}.ap"])
        assert O.result_filename("twitch/ES/RU", "vanilla-clean", mode, "unbalanced", 40, 42) == str(g[f"{mode}.filename"])
    # balanced-full
    ab = csr_from(g, "bf.adj")
    xb = torch.from_numpy(g["bf.x"])
    adjb = O.to_torch_sparse(O.first_order_gcn(ab))
    np.random.seed(82)
    (ex, nex), nodes = O.sample_balanced_full(ab)
    assert nodes == list(range(ab.shape[0]))
    assert np.array_equal(np.asarray(ex, dtype=np.int64).reshape(-1, 2), g["bf.exist"])
    assert np.array_equal(np.asarray(nex, dtype=np.int64).reshape(-1, 2), g["bf.nonexist"])
    for dt, tag in ((torch.float32, "ref32"), (torch.float64, "ref64")):
        Pd = {k: v.to(dt) for k, v in P.items()}
        ne, nn = O.efficient_balanced_scores(xb.to(dt), adjb.to(dt), Pd, ab.shape[0], ex, nex, 1e-4)
        assert np.array_equal(np.asarray(ne), g[f"bf.{tag}.norm_exist"])
        assert np.array_equal(np.asarray(nn), g[f"bf.{tag}.norm_nonexist"])
    m = O.attack_metrics(g["bf.ref32.norm_exist"].tolist(), g["bf.ref32.norm_nonexist"].tolist())
    assert m["auc"] == float(g["bf.ref32.auc"])
    assert O.result_filename("twitch/ES/RU", "vanilla-clean", "efficient", "balanced-full", ab.shape[0], 82) == str(g["bf.ref32.filename"])
    vec = O.baseline_vectors("baseline", xb, adjb, P, "twitch/ES/RU")
    ne, nn = O.baseline_attack_balanced(vec, ex, nex)
    assert np.array_equal(np.asarray(ne), g["bf.baseline.norm_exist"])
    assert np.array_equal(np.asarray(nn), g["bf.baseline.norm_nonexist"])
    # GCN3 under the efficient attack
    names = (("W1", "gc1.weight"), ("b1", "gc1.bias"), ("W2", "gc2.weight"), ("b2", "gc2.bias"), ("W3", "gc3.weight"), ("b3", "gc3.bias"))
    P3 = {k: torch.from_numpy(g[f"gcn3.sd.{n}"]) for k, n in names}
    nodes3 = g["gcn3.ref32.test_nodes"]
    infl = O.influence_matrix(x, adj, P3, nodes3, 1e-4, forward=O.gcn3_forward)
    assert np.array_equal(infl, g["gcn3.ref32.influence_val"])
    infl64 = O.influence_matrix(x.double(), adj.double(), {k: v.double() for k, v in P3.items()}, nodes3, 1e-4,
                                forward=O.gcn3_forward)
    assert np.array_equal(infl64, g["gcn3.ref64.influence_val"])


@pytest.mark.parametrize("key", ["er300", "pl600", "lap600"])
def test_restricted_oracle_equals_the_verbatim_one(key):
    """oracle.RestrictedOracle (baseline once + only the rows a probe can change; used where two full fp64 forwards per
    probe take minutes) gives the quantity of the verbatim restatement -- here against the REFERENCE's own fp64 matrices."""
    from conftest import csr_from, golden_args, load_golden
    from oracle import linkteller_oracle as O
    g = load_golden("influence.npz")
    args = golden_args(g, key)
    served = csr_from(g, f"{key}.served") if f"{key}.served.n" in g else csr_from(g, f"{key}.adj")
    a_hat = O.NORMALIZERS[args["norm"]](served)
    P = {k: g[f"{key}.sd.{n}"] for k, n in (("W1", "gc1.weight"), ("b1", "gc1.bias"), ("W2", "gc2.weight"), ("b2", "gc2.bias"))}
    nodes = g[f"{key}.ref64.test_nodes"] if f"{key}.ref64.test_nodes" in g else g[f"{key}.ref32.test_nodes"]
    ref64 = g[f"{key}.ref64.influence_val"]
    got = O.RestrictedOracle(g[f"{key}.x"], a_hat, P).rows(nodes, nodes, args["influence"])
    assert np.abs(got - ref64).max() <= 1e-9 * ref64.max()
    assert np.array_equal(got == 0, ref64 == 0)
