"""CLI flag surface, the twitch Worker on a tiny synthetic MUSAE-format dataset, and the DP
adjacency generators against the reference's outputs.  CPU only."""
import argparse
import json
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import csr_from, load_golden
from linkteller_amd import dp, main as lt_main, synth


def test_cli_accepts_reference_flags_with_reference_defaults():
    a = lt_main.get_arguments([])
    ref_defaults = dict(no_cuda=False, fastmode=False, seed=42, num_epochs=500, lr=0.01, weight_decay=5e-4, hidden=16,
                        hidden1=16, hidden2=16, dropout=0.5, dataset="cora", model_path="", mode="vanilla-clean",
                        init_method="knn", cluster_method="hierarchical", scale="small", break_method="kmeans",
                        norm="AugNormAdj", sample_type="balanced", epsilon=0.1, delta=1e-5, influence=0.0001,
                        train_ratio=0.5, patience=10, n_clusters=10, n_test=100, n_layer=2, break_ratio=1,
                        feature_size=-1, k=1, approx=False, attack=False, test=False, break_down=False,
                        display=False, same_size=False, eval_degree=False, trainable=False, early=False,
                        fnormalize=False, noise_seed=42, sample_seed=42, cluster_seed=42, knn=-1,
                        noise_type="laplace", perturb_type="discrete", attack_mode="efficient", coeff=1, degree=2,
                        assign_seed=42)                       # reference main.py:17-93
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k
    # README command line (README.md:71 uses the --eps prefix of --epsilon)
    a = lt_main.get_arguments("--mode vanilla --dataset twitch/ES/RU --hidden 256 --norm FirstOrderGCN --test "
                              "--model-path m.pt --attack --attack-mode efficient --sample-type unbalanced "
                              "--n-test 500 --eps 5 --perturb-type continuous".split())
    assert a.epsilon == 5 and a.n_test == 500 and a.attack and a.test and a.influence_mode == "full"
    with pytest.raises(NotImplementedError):
        lt_main.main(["--dataset", "twitch/ES/RU"])           # training is refused, not faked


def test_dp_generators_match_reference():
    g = load_golden("dp_adjacency.npz")
    a = csr_from(g, "adj")
    for perturb, eps in (("continuous", 5.0), ("continuous", 1.0), ("discrete", 4.0), ("discrete", 7.0)):
        res = sp.csr_matrix(dp.perturb_adj(sp.csr_matrix(a), perturb, eps, 42))
        res.sort_indices()
        tag = f"{perturb}.eps{eps:g}"
        assert np.array_equal(res.indptr, g[f"{tag}.indptr"]), tag
        assert np.array_equal(res.indices, g[f"{tag}.indices"]), tag
        assert np.array_equal(np.asarray(res.data, dtype=np.float64), g[f"{tag}.data"]), tag


def _write_musae(root, code, adj, n_feat_ids, seed):
    rng = np.random.RandomState(seed)
    d = os.path.join(root, "twitch", code)
    os.makedirs(d)
    n = adj.shape[0]
    feats = {str(i): sorted(rng.choice(n_feat_ids, rng.randint(1, 6), replace=False).tolist()) for i in range(n)}
    json.dump(feats, open(os.path.join(d, f"musae_{code}_features.json"), "w"))
    coo = sp.triu(adj, k=1).tocoo()
    with open(os.path.join(d, f"musae_{code}_edges.csv"), "w") as fh:
        fh.write("from,to\n" + "".join(f"{i},{j}\n" for i, j in zip(coo.row, coo.col)))
    perm = rng.permutation(n)
    with open(os.path.join(d, f"musae_{code}_target.csv"), "w") as fh:
        fh.write("id,days,mature,views,partner,new_id\n" +
                 "".join(f"{1000 + i},1,{bool(i % 3 == 0)},5,False,{i}\n" for i in perm))
    return feats


def test_twitch_worker_reads_musae_layout(tmp_path, monkeypatch):
    from linkteller_amd.worker import Worker
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    a1, a2 = synth.erdos_renyi_graph(40, 90, seed=1), synth.erdos_renyi_graph(30, 60, seed=2)
    _write_musae(str(tmp_path), "ES", a1, 50, 1)
    f2 = _write_musae(str(tmp_path), "RU", a2, 50, 2)
    args = argparse.Namespace(norm="FirstOrderGCN", perturb_type="continuous", epsilon=5.0, noise_seed=42,
                              noise_type="laplace", delta=1e-5)
    w = Worker(args, dataset="twitch/ES/RU", mode="vanilla-clean", data_root=str(tmp_path))
    assert w.transfer and w.n_nodes_1 == 40 and w.n_nodes_2 == 30 and w.n_features == 3170 and w.n_classes == 2
    assert (w.adj_ori != a2).nnz == 0 and w.adj_ori.dtype == np.float32
    assert w.features_2.shape == (30, 3170) and w.features_2.dtype == torch.float32
    assert w.labels_2.tolist() == [int(i % 3 == 0) for i in range(30)]
    # standardised with graph-1 statistics: a feature id never used in ES keeps mean 0 / scale 1
    raw = np.zeros((30, 3170)); [raw.__setitem__((int(i), v), 1) for i, v in f2.items()]
    assert w.adj_2.is_sparse and w.adj_2.shape == (30, 30)
    dense = w.adj_2.to_dense().numpy()
    deg = np.asarray(a2.sum(1)).ravel()
    want = np.eye(30) + a2.toarray() / np.sqrt(np.outer(deg, deg).clip(1e-30))
    assert np.allclose(dense, want, atol=1e-6)
    w2 = Worker(args, dataset="twitch/ES/RU", mode="vanilla", data_root=str(tmp_path))   # LapGraph-served graph
    assert (w2.adj_ori != a2).nnz == 0                       # pairs still come from the clean graph
    assert not np.allclose(w2.adj_2.to_dense().numpy(), dense)
    with pytest.raises(NotImplementedError):
        Worker(args, dataset="cora", mode="vanilla-clean", data_root=str(tmp_path))
