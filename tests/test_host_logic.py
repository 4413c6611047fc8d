"""Host-side logic of the product (normalisers, CSR packing, sampler, pair scoring, metrics, result
file) against the golden vectors.  The device compute is replaced by the golden influence matrix
here -- the kernels themselves are checked in test_gpu_parity.py."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import csr_from, golden_args, load_golden
from linkteller_amd import graph, sampling
from linkteller_amd.attacker import Attacker


def test_normalizers_match_reference_bits():
    g = load_golden("normalizer.npz")
    for key in g["keys"]:
        a = csr_from(g, f"{key}.adj")
        for norm in ("FirstOrderGCN", "AugNormAdj"):
            res = graph.fetch_normalization(norm)(a)
            ref = sp.csr_matrix((g[f"{key}.{norm}.data"], (g[f"{key}.{norm}.row"], g[f"{key}.{norm}.col"])),
                                shape=a.shape)
            ref.sort_indices()
            assert np.array_equal(res.indptr, ref.indptr) and np.array_equal(res.indices, ref.indices)
            assert np.array_equal(res.data, ref.data)                     # float64 bits
            n, rowptr, col, val = graph.csr_arrays(res)
            tref = sp.csr_matrix((g[f"{key}.{norm}.t_values"], (g[f"{key}.{norm}.t_indices"][0],
                                                              g[f"{key}.{norm}.t_indices"][1])), shape=a.shape)
            tref.sort_indices()
            assert rowptr.dtype == np.int32 and col.dtype == np.int32 and val.dtype == np.float32
            assert np.array_equal(val, tref.data) and np.array_equal(col, tref.indices)
            t = graph.sparse_mx_to_torch_sparse_tensor(res)
            assert t._values().dtype == torch.float32 and t._indices().dtype == torch.int64
    with pytest.raises(NotImplementedError):
        graph.fetch_normalization("nope")


def test_all_six_normalizers_formulas():
    a = csr_from(load_golden("normalizer.npz"), "er50.adj").astype(np.float64)   # float64 in -> float64 math
    d = np.asarray(a.sum(1)).ravel()
    dense = a.toarray().astype(np.float64)
    n = a.shape[0]
    eye = np.eye(n)
    dm = np.diag(d ** -0.5)
    d1 = np.diag((d + 1) ** -0.5)
    want = {
        "FirstOrderGCN": eye + dm @ dense @ dm,
        "NormAdj": dm @ dense @ dm,
        "AugNormAdj": d1 @ (dense + eye) @ d1,
        "BingGeNormAdj": d1 @ (dense + eye) @ d1 + eye,
        "RWalk": np.diag(1 / d) @ dense,
        "AugRWalk": np.diag(1 / (d + 1)) @ (dense + eye),
    }
    for name, w in want.items():
        assert np.allclose(graph.fetch_normalization(name)(a).toarray(), w, rtol=1e-13, atol=1e-15), name


def test_sampler_matches_reference_stream_and_order():
    g = load_golden("sampler.npz")
    a = csr_from(g, "adj")
    for tag in g["combos"]:
        ds, st, seed = str(tag).split(".")
        np.random.seed(int(seed))
        (ex, nex), nodes = sampling.construct_edge_sets_from_random_subgraph(ds.replace("_", "/"), st, a, 24)
        assert np.array_equal(nodes, g[f"{tag}.nodes"])
        assert np.array_equal(ex, g[f"{tag}.exist"]) and np.array_equal(nex, g[f"{tag}.nonexist"])
    with pytest.raises(NotImplementedError):
        sampling.construct_edge_sets_from_random_subgraph("twitch/ES/RU", "bfs", a, 4)
    with pytest.raises(NotImplementedError):
        sampling.degree_bounds("unknown-dataset")


@pytest.mark.parametrize("key", ["er300", "lap600"])
def test_attacker_scoring_metrics_and_result_file(influence_golden, key, tmp_path, monkeypatch, capsys):
    import argparse
    import types
    g = influence_golden
    args = argparse.Namespace(**golden_args(g, key))
    a = csr_from(g, f"{key}.adj")
    worker = types.SimpleNamespace(features_2=None, adj_2=None, adj_ori=a, n_nodes=a.shape[0])
    atk = Attacker(args, model=None, worker=worker)
    atk.prepare_test_data()
    assert np.array_equal(atk.test_nodes, g[f"{key}.ref32.test_nodes"])
    assert np.array_equal(atk.exist_edges, g[f"{key}.ref32.exist"])
    # device compute replaced by the golden matrix: everything downstream must reproduce the reference
    monkeypatch.setattr(atk, "influence_matrix", lambda mode=None: g[f"{key}.ref32.influence_val"].copy())
    monkeypatch.chdir(tmp_path)
    atk.link_prediction_attack_efficient()
    out = capsys.readouterr().out
    assert "time for predicting edges:" in out and "attack results saved to:" in out
    assert atk.auc == float(g[f"{key}.ref32.auc"]) and atk.ap == float(g[f"{key}.ref32.ap"])
    fn = str(g[f"{key}.ref32.filename"])
    assert os.path.exists(fn)
    saved = torch.load(fn, weights_only=False)
    assert repr({k: sorted(v.keys()) for k, v in saved.items()}) == str(g[f"{key}.ref32.schema"])
    assert type(saved["result"]["y"]).__name__ == str(g[f"{key}.ref32.y_type"])
    assert type(saved["result"]["pred"][0]).__name__ == str(g[f"{key}.ref32.pred_type"])
    assert np.array_equal(np.asarray(saved["result"]["pred"]), g[f"{key}.ref32.pred"])
    assert np.array_equal(np.asarray(saved["result"]["y"]), g[f"{key}.ref32.y"])
    for k_, s_ in (("fpr", "auc"), ("tpr", "auc"), ("thresholds", "auc"), ("precision", "pr"), ("recall", "pr")):
        assert np.array_equal(saved[s_][k_], g[f"{key}.ref32.{k_}"])
    assert str(saved["auc"]["fpr"].dtype) == str(g[f"{key}.ref32.fpr_dtype"])


def test_attacker_rejects_what_the_reference_cannot_run():
    import argparse
    import types
    args = argparse.Namespace(dataset="cora", sample_type="bfs", n_test=4, sample_seed=1, influence=1e-4,
                              mode="vanilla-clean", attack_mode="efficient")
    w = types.SimpleNamespace(features=None, adj_full=None, adj_ori=sp.identity(8, format="csr"), n_nodes=8)
    with pytest.raises(NotImplementedError):
        Attacker(args, None, w).prepare_test_data()


def test_balanced_full_sampler_matches_reference():
    g = load_golden("next_rows.npz")
    ab = csr_from(g, "bf.adj")
    np.random.seed(82)
    (ex, nex), nodes = sampling.construct_balanced_edge_sets("twitch/ES/RU", "balanced-full", ab, 7)
    assert nodes == list(range(ab.shape[0]))
    assert np.array_equal(ex, g["bf.exist"]) and np.array_equal(nex, g["bf.nonexist"])


def test_baseline_attacks_host_math(monkeypatch, tmp_path):
    """LSA2 baselines: the correlation math against the reference's scores, with the model forward
    (the only device step) replaced by the oracle's posteriors."""
    import argparse
    import types
    from oracle import linkteller_oracle as O
    g = load_golden("next_rows.npz")
    a = csr_from(g, "adj")
    x = torch.from_numpy(g["x"])
    P = {k: torch.from_numpy(g[f"sd.{n}"]) for k, n in (("W1", "gc1.weight"), ("b1", "gc1.bias"), ("W2", "gc2.weight"), ("b2", "gc2.bias"))}
    adj = O.to_torch_sparse(O.first_order_gcn(a))
    monkeypatch.chdir(tmp_path)
    for mode in ("baseline", "baseline-feat"):
        args = argparse.Namespace(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=40, sample_seed=42,
                                  influence=1e-4, mode="vanilla-clean", attack_mode=mode)
        w = types.SimpleNamespace(features_2=x, adj_2=adj, adj_ori=a, n_nodes=a.shape[0])
        atk = Attacker(args, model=lambda f, ad: O.gcn_forward(f, ad, P), worker=w)
        atk.prepare_test_data()
        atk.baseline_attack()
        ref = np.concatenate([g[f"{mode}.norm_exist"], g[f"{mode}.norm_nonexist"]])
        got = np.asarray(torch.load(str(g[f"{mode}.filename"]), weights_only=False)["result"]["pred"])
        assert np.abs(got - ref).max() <= 2e-6                  # fp32 dot-product order only
        if mode == "baseline-feat":
            # with C = 2 the centred softmax posteriors are collinear, every correlation is +-1 up to
            # rounding and the reference's own AUC is decided by that rounding: scores, not AUC, are the check
            assert abs(atk.auc - float(g[f"{mode}.auc"])) <= 1e-4
    ab = csr_from(g, "bf.adj")
    xb = torch.from_numpy(g["bf.x"])
    adjb = O.to_torch_sparse(O.first_order_gcn(ab))
    args = argparse.Namespace(dataset="twitch/ES/RU", sample_type="balanced-full", n_test=7, sample_seed=82,
                              influence=1e-4, mode="vanilla-clean", attack_mode="baseline")
    w = types.SimpleNamespace(features_2=xb, adj_2=adjb, adj_ori=ab, n_nodes=ab.shape[0])
    atk = Attacker(args, model=lambda f, ad: O.gcn_forward(f, ad, P), worker=w)
    assert args.n_test == ab.shape[0]
    atk.prepare_test_data()
    atk.baseline_attack_balanced()
    ref = np.concatenate([g["bf.baseline.norm_exist"], g["bf.baseline.norm_nonexist"]])
    got = np.asarray(torch.load(str(g["bf.baseline.filename"]), weights_only=False)["result"]["pred"])
    assert np.abs(got - ref).max() <= 2e-6


def test_attacker_walk_cache_follows_the_model():
    """Attacker._walk skips state_dict() while the model's parameters are the objects and storages of the last walk (round 5): an
    in-place update keeps the cache (the baseline's refresh re-reads the borrowed tensors), a replaced parameter, a replaced
    model and a model whose state_dict keys are not attribute paths walk again."""
    import torch
    from linkteller_amd.attacker import Attacker
    from linkteller_amd.gcn import GCN, GCN3
    a = Attacker.__new__(Attacker)
    a.model = GCN(30, 16, 2, 0.5)
    kind, sd = a._walk()
    assert kind == "gcn2" and a._walk()[1] is sd                     # cached
    with torch.no_grad():
        a.model.gc2.weight.mul_(2)
    assert a._walk()[1] is sd and torch.equal(sd["gc2.weight"], a.model.gc2.weight)
    a.model.gc2.weight = torch.nn.Parameter(torch.zeros(16, 2))
    kind2, sd2 = a._walk()
    assert sd2 is not sd and sd2["gc2.weight"].data_ptr() == a.model.gc2.weight.data_ptr()
    a.model = GCN3(30, 16, 8, 2, 0.5)
    assert a._walk()[0] == "gcn3"

    class Odd(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))

        def state_dict(self, *a_, **k_):
            return {"gc1.weight": self.w.detach()}
    a.model = Odd()
    assert a._walk()[0] == "generic" and a._walk_cache is None
