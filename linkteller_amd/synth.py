"""Seeded synthetic inputs with the shapes BASELINE.json names.

There is no dataset on the box (the reference reads ``./data/twitch/<CC>/musae_*``,
reference utils/load.py:47,63,455), so every benchmark / parity graph is generated
here.  Nothing in this file mirrors reference code; it only reproduces the *shapes*:

* twitch-shaped: undirected simple 0/1 graph with N nodes / E edges, F one-hot-ish
  Bernoulli features passed through a StandardScaler (reference worker.py:486-490
  standardises the MUSAE one-hot features, which makes them fully dense).
* hub-and-spoke power-law graph (degree skew like real MUSAE graphs).
* R-MAT (a,b,c,d) = (0.57,0.19,0.19,0.05) for the scale-out config.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

# Public MUSAE statistics quoted in SURVEY.md section 8 (not verifiable offline).
TWITCH_SHAPES = {
    "twitch-RU": dict(n=4385, e=37304),
    "twitch-ES": dict(n=4648, e=59382),
}
TWITCH_N_FEATURES = 3170  # reference utils/load.py:56, worker.py:494


def _symmetric_csr(rows: np.ndarray, cols: np.ndarray, n: int) -> sp.csr_matrix:
    """Simple undirected 0/1 CSR (float32) from directed draws; drops self loops / duplicates."""
    keep = rows != cols
    rows, cols = rows[keep], cols[keep]
    lo = np.minimum(rows, cols).astype(np.int64)
    hi = np.maximum(rows, cols).astype(np.int64)
    key = np.unique(lo * n + hi)
    lo, hi = key // n, key % n
    r = np.concatenate([lo, hi])
    c = np.concatenate([hi, lo])
    a = sp.csr_matrix((np.ones(r.shape[0], dtype=np.float32), (r, c)), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    return a


def erdos_renyi_graph(n: int, n_edges: int, seed: int = 0) -> sp.csr_matrix:
    """Undirected G(n, m)-like graph with exactly ``n_edges`` distinct edges."""
    rng = np.random.RandomState(seed)
    have = np.empty(0, dtype=np.int64)
    while have.shape[0] < n_edges:
        m = int((n_edges - have.shape[0]) * 1.2) + 16
        u = rng.randint(0, n, size=m).astype(np.int64)
        v = rng.randint(0, n, size=m).astype(np.int64)
        ok = u != v
        key = np.minimum(u[ok], v[ok]) * n + np.maximum(u[ok], v[ok])
        have = np.unique(np.concatenate([have, key]))
    # deterministic down-select to exactly n_edges
    if have.shape[0] > n_edges:
        have = have[np.sort(rng.choice(have.shape[0], n_edges, replace=False))]
    return _symmetric_csr(have // n, have % n, n)


def powerlaw_graph(n: int, n_edges: int, seed: int = 0, exponent: float = 2.2) -> sp.csr_matrix:
    """Chung-Lu style graph: endpoint i drawn with probability ~ (i+1)^(-1/(exponent-1)).

    Gives a few hubs with degree in the hundreds/thousands like the real MUSAE graphs,
    which is what stresses row-length imbalance in the SpMM.
    """
    rng = np.random.RandomState(seed)
    w = (np.arange(n, dtype=np.float64) + 1.0) ** (-1.0 / (exponent - 1.0))
    w /= w.sum()
    perm = rng.permutation(n)  # hubs are not the low node ids
    have = np.empty(0, dtype=np.int64)
    while have.shape[0] < n_edges:
        m = int((n_edges - have.shape[0]) * 1.5) + 16
        u = perm[rng.choice(n, size=m, p=w)].astype(np.int64)
        v = perm[rng.choice(n, size=m, p=w)].astype(np.int64)
        ok = u != v
        key = np.minimum(u[ok], v[ok]) * n + np.maximum(u[ok], v[ok])
        have = np.unique(np.concatenate([have, key]))
    if have.shape[0] > n_edges:
        have = have[np.sort(rng.choice(have.shape[0], n_edges, replace=False))]
    return _symmetric_csr(have // n, have % n, n)


def rmat_graph(scale: int, n_draws: int, seed: int = 42,
               abcd=(0.57, 0.19, 0.19, 0.05)) -> sp.csr_matrix:
    """R-MAT: ``n_draws`` directed draws on 2^scale nodes, then symmetrise + dedupe."""
    rng = np.random.RandomState(seed)
    a, b, c, _ = abcd
    rows = np.zeros(n_draws, dtype=np.int64)
    cols = np.zeros(n_draws, dtype=np.int64)
    for bit in range(scale):
        r = rng.random_sample(n_draws)
        # quadrant: a -> (0,0), b -> (0,1), c -> (1,0), d -> (1,1)
        right = (r >= a) & (r < a + b) | (r >= a + b + c)
        down = r >= a + b
        rows |= down.astype(np.int64) << bit
        cols |= right.astype(np.int64) << bit
    return _symmetric_csr(rows, cols, 1 << scale)


def rmat_draws(scale: int) -> int:
    """Directed draws for an R-MAT graph of 2^scale nodes: 40 M at scale 21 (BASELINE configs[4]: |V| = 2 M, |E| = 40 M),
    the same 19.07 per node at other scales."""
    return 40_000_000 if scale == 21 else int(round((1 << scale) * 40_000_000 / (1 << 21)))


def standard_scale(x: np.ndarray) -> np.ndarray:
    """Column standardisation with zero-variance columns left at scale 1 (sklearn semantics)."""
    mean = x.mean(axis=0)
    std = x.std(axis=0)
    std[std == 0.0] = 1.0
    return (x - mean) / std


def twitch_like_features(n: int, f: int = TWITCH_N_FEATURES, seed: int = 0,
                         density: float = 0.006) -> np.ndarray:
    """Bernoulli(density) indicator matrix -> standardised dense float32 [n, f]."""
    rng = np.random.RandomState(seed)
    x = (rng.random_sample((n, f)) < density).astype(np.float64)
    return standard_scale(x).astype(np.float32)


def gaussian_features(n: int, f: int, seed: int = 0) -> np.ndarray:
    rng = np.random.RandomState(seed)
    return rng.standard_normal((n, f)).astype(np.float32)


def gcn_weights(f: int, h: int, c: int, seed: int = 42):
    """2-layer GCN parameters with the reference's init law U(-1/sqrt(out), 1/sqrt(out))
    (reference gcn/layers.py:24-28), drawn from numpy so they do not depend on torch's RNG."""
    rng = np.random.RandomState(seed)

    def u(shape, fan_out):
        s = 1.0 / np.sqrt(fan_out)
        return rng.uniform(-s, s, size=shape).astype(np.float32)

    return dict(W1=u((f, h), h), b1=u((h,), h), W2=u((h, c), c), b2=u((c,), c))


def twitch_like_problem(name: str = "twitch-RU", hidden: int = 256, n_classes: int = 2,
                        seed: int = 0, n_features: int = TWITCH_N_FEATURES,
                        powerlaw: bool = False):
    """(adjacency CSR, features [N,F] f32, weights dict) for a BASELINE.json twitch config."""
    shp = TWITCH_SHAPES[name]
    gen = powerlaw_graph if powerlaw else erdos_renyi_graph
    adj = gen(shp["n"], shp["e"], seed=seed)
    x = twitch_like_features(shp["n"], n_features, seed=seed + 1)
    w = gcn_weights(n_features, hidden, n_classes, seed=42)
    return adj, x, w


def write_musae_dataset(root: str, code: str, adj, n_feat_ids: int, seed: int):
    """Writes ``<root>/twitch/<code>/musae_<code>_{features.json,edges.csv,target.csv}`` in the MUSAE layout
    the reference reads (utils/load.py:47, 63, 455): a JSON dict node -> list of feature ids, an edge list with
    one line per undirected edge, and a target table whose rows are in shuffled order (``new_id`` is the node).
    Synthetic content, seeded; returns the feature dict."""
    import json
    import os
    rng = np.random.RandomState(seed)
    d = os.path.join(root, "twitch", code)
    os.makedirs(d)
    n = adj.shape[0]
    feats = {str(i): sorted(rng.choice(n_feat_ids, rng.randint(1, 6), replace=False).tolist()) for i in range(n)}
    with open(os.path.join(d, f"musae_{code}_features.json"), "w") as fh:
        json.dump(feats, fh)
    coo = sp.triu(adj, k=1).tocoo()
    with open(os.path.join(d, f"musae_{code}_edges.csv"), "w") as fh:
        fh.write("from,to\n" + "".join(f"{i},{j}\n" for i, j in zip(coo.row, coo.col)))
    perm = rng.permutation(n)
    with open(os.path.join(d, f"musae_{code}_target.csv"), "w") as fh:
        fh.write("id,days,mature,views,partner,new_id\n" +
                 "".join(f"{1000 + i},1,{bool(i % 3 == 0)},5,False,{i}\n" for i in perm))
    return feats
