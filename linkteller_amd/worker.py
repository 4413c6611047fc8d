"""``Worker`` for the datasets the hot path is evaluated on: the twitch transfer
setting ``twitch/<TRAIN>/<TEST>`` of reference worker.py:470-496, 549-552, 589-678.

Reads the MUSAE files the reference reads (``./data/twitch/<CC>/musae_<CC>_{features.json,
target.csv,edges.csv}``, utils/load.py:42-93, 452-460), standardises the one-hot features with
statistics of graph 1, optionally perturbs both graphs (edge DP), normalises, and exposes the
attributes ``Attacker`` / ``GCNTrainer`` use: ``features_1/2``, ``adj_1/2`` (normalised, torch sparse
COO like the reference), ``adj_ori`` (clean scipy CSR of graph 2), ``labels_1/2``, sizes.
Other dataset families of the reference have no data on the box and are refused.
"""
from __future__ import annotations

import json
import os

import numpy as np
import scipy.sparse as sp
import torch

from . import dp
from .graph import fetch_normalization, sparse_mx_to_torch_sparse_tensor

TWITCH_N_FEATURES = 3170   # reference utils/load.py:56


def read_musae_features(folder: str, code: str, n_features: int = TWITCH_N_FEATURES):
    """musae_<code>_features.json -> dense 0/1 [n, n_features]; musae_<code>_target.csv -> labels
    ('mature' looked up by 'new_id'), reference utils/load.py:47-69."""
    import pandas as pd
    with open(os.path.join(folder, f"musae_{code}_features.json")) as fh:
        data = json.load(fh)
    n = len(data)
    feats = np.zeros((n, n_features))
    for node, items in data.items():
        feats[int(node), items] = 1
    tgt = pd.read_csv(os.path.join(folder, f"musae_{code}_target.csv"))
    by_id = dict(zip(map(int, tgt["new_id"].values), map(int, tgt["mature"].values)))
    labels = torch.LongTensor([by_id[i] for i in range(n)])
    return feats, labels


def read_musae_edges(folder: str, code: str, n_nodes: int):
    """musae_<code>_edges.csv -> A + A^T as float32 CSR (reference utils/load.py:452-460)."""
    import pandas as pd
    edges = pd.read_csv(os.path.join(folder, f"musae_{code}_edges.csv")).values
    a = sp.csr_matrix((np.ones(edges.shape[0]), (edges[:, 0], edges[:, 1])), shape=(n_nodes, n_nodes),
                      dtype=np.float32)
    return a + a.T


class Worker:
    def __init__(self, args, dataset="", mode="", data_root="./data"):
        self.args = args
        self.dataset = dataset
        self.mode = mode
        self.data_root = data_root
        self.transfer = (dataset.startswith("twitch") and not dataset.startswith("twitch-train")) \
            or dataset.startswith("wikipedia") or dataset.startswith("deezer")
        self.load_data()

    def load_data(self):
        if not (self.dataset.startswith("twitch") and not self.dataset.startswith("twitch-train")):
            raise NotImplementedError(f"dataset = {self.dataset} not implemented! (hot-path scope: twitch/<A>/<B>)")
        family, code_1, code_2 = self.dataset.split("/")        # twitch/ES/RU
        self.dataset1, self.dataset2 = f"{family}/{code_1}", f"{family}/{code_2}"
        f1, self.labels_1 = read_musae_features(os.path.join(self.data_root, family, code_1), code_1)
        f2, self.labels_2 = read_musae_features(os.path.join(self.data_root, family, code_2), code_2)
        from sklearn.preprocessing import StandardScaler
        scaler = StandardScaler().fit(f1)                         # worker.py:486-490
        self.features_1 = torch.FloatTensor(scaler.transform(f1))
        self.features_2 = torch.FloatTensor(scaler.transform(f2))
        self.n_nodes_1, self.n_nodes_2 = len(self.labels_1), len(self.labels_2)
        self.n_nodes = self.n_nodes_2
        self.n_features, self.multi_label, self.n_classes = TWITCH_N_FEATURES, 1, 2
        self.adj_1 = read_musae_edges(os.path.join(self.data_root, family, code_1), code_1, self.n_nodes_1)
        self.adj_2 = read_musae_edges(os.path.join(self.data_root, family, code_2), code_2, self.n_nodes_2)
        self.adj_ori = sp.csr_matrix.copy(self.adj_2)             # clean ground truth, worker.py:552
        self.prepare_data()

    def prepare_data(self):
        a = self.args
        if self.mode == "vanilla":                                # serve the model on DP graphs, worker.py:632-635
            self.adj_1 = dp.perturb_adj(self.adj_1, a.perturb_type, a.epsilon, a.noise_seed, a.noise_type, a.delta)
            self.adj_2 = dp.perturb_adj(self.adj_2, a.perturb_type, a.epsilon, a.noise_seed, a.noise_type, a.delta)
            print("perturbing done!")
        elif self.mode != "vanilla-clean":
            raise NotImplementedError("mode = {} not implemented!".format(self.mode))
        normalizer = fetch_normalization(a.norm)
        self.adj_1 = sparse_mx_to_torch_sparse_tensor(normalizer(self.adj_1))
        self.adj_2 = sparse_mx_to_torch_sparse_tensor(normalizer(self.adj_2))
        print("Normalizing Adj done!")
        if torch.cuda.is_available():                             # worker.py:665-678
            self.features_1, self.features_2 = self.features_1.cuda(), self.features_2.cuda()
            self.labels_1, self.labels_2 = self.labels_1.cuda(), self.labels_2.cuda()
            self.adj_1, self.adj_2 = self.adj_1.cuda(), self.adj_2.cuda()
