"""Node / node-pair sampling for the attack (reference attacker.py:33-48, utils/load.py:304-381).

Host-side integer work.  The node draw must reproduce numpy's *legacy global* stream
(``np.random.seed(sample_seed)`` then ``np.random.choice(..., replace=False)``, attacker.py:45 and
utils/load.py:379), so it stays on ``np.random``; the O(n_test^2) pair enumeration is vectorised
but returns the pairs in the reference's order (i < j over the sampled nodes, row-major).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def degree_bounds(dataset: str):
    """(lo, hi) thresholds of utils/load.py:354-372."""
    if dataset.startswith("twitch"):
        return (5 if "PTBR" not in dataset else 10), 10
    if dataset in ("flickr", "ppi") or dataset.startswith("deezer"):
        return 15, 30
    if dataset in "cora":        # the reference tests substring membership: ( 'cora' ) is a str
        return 3, 4
    if dataset in "citeseer":
        return 3, 3
    if dataset in "pubmed":
        return 10, 10
    raise NotImplementedError(f"lo and hi for dataset = {dataset} not set!")


def edge_sets_among_nodes(adj: sp.csr_matrix, nodes: np.ndarray):
    """All pairs (nodes[i], nodes[j]), i < j, split by structural presence of nodes[j] in row
    nodes[i] of ``adj`` (utils/load.py:304-326).  Returns two int64 arrays of shape [k, 2]."""
    nodes = np.asarray(nodes, dtype=np.int64)
    k = nodes.shape[0]
    pattern = sp.csr_matrix((np.ones(adj.indices.shape[0], dtype=np.int8), adj.indices, adj.indptr),
                            shape=adj.shape)
    sub = pattern[nodes][:, nodes].toarray() != 0
    iu, ju = np.triu_indices(k, k=1)
    present = sub[iu, ju]
    pairs = np.stack([nodes[iu], nodes[ju]], axis=1)
    return pairs[present], pairs[~present]


def construct_edge_sets_from_random_subgraph(dataset, sample_type, adj, n_samples):
    """Same signature/return shape as utils/load.py:338-381: ((edges, non_edges), nodes)."""
    adj = sp.csr_matrix(adj)
    n_nodes = adj.shape[0]
    if sample_type == "unbalanced":
        candidates = np.arange(n_nodes)
    else:
        deg = np.diff(adj.indptr)
        lo, hi = degree_bounds(dataset)
        if sample_type == "unbalanced-lo":
            candidates = np.where(deg <= lo)[0]
        elif sample_type == "unbalanced-hi":
            candidates = np.where(deg >= hi)[0]
        else:
            raise NotImplementedError(f"sample_type = {sample_type} not implemented!")
    print("#indice =", len(candidates))
    nodes = np.random.choice(candidates, n_samples, replace=False)
    edges, non_edges = edge_sets_among_nodes(adj, nodes)
    print("#nodes =", len(nodes))
    print("#edges_set =", len(edges))
    print("#nonedge_set =", len(non_edges))
    return (edges, non_edges), nodes


def construct_balanced_edge_sets(dataset, sample_type, adj, n_samples):
    """``balanced-full`` (reference utils/load.py:219-249): every u < v edge of the graph, plus as many
    random pairs that are adjacent in neither direction.  The non-edge draws are two scalar
    ``np.random.choice(n_nodes)`` calls per candidate, in the reference's order (u == v and repeated
    pairs can occur, as they do there).  Returns ((edges, non_edges), all nodes)."""
    adj = sp.csr_matrix(adj)
    n_nodes = adj.shape[0]
    indptr, indices = adj.indptr, adj.indices
    rows = np.repeat(np.arange(n_nodes, dtype=np.int64), np.diff(indptr))
    upper = indices > rows
    edges = np.stack([rows[upper], indices[upper].astype(np.int64)], axis=1)
    nbr_sets = [set(indices[indptr[u]: indptr[u + 1]].tolist()) for u in range(n_nodes)]
    non_edges = np.empty((edges.shape[0], 2), dtype=np.int64)
    k = 0
    while k < edges.shape[0]:
        u = np.random.choice(n_nodes)
        v = np.random.choice(n_nodes)
        if v not in nbr_sets[u] and u not in nbr_sets[v]:
            non_edges[k] = (u, v)
            k += 1
    print(f"sampling done! len(edge_set) = {len(edges)}, len(nonedge_set) = {len(non_edges)}")
    return (edges, non_edges), list(range(n_nodes))
