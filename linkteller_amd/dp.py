"""Edge-DP graph perturbation that precedes the hot path in the DP evaluation:
EdgeRand (``--perturb-type discrete``, reference worker.py:213-278) and LapGraph
(``--perturb-type continuous``, reference worker.py:281-335).

The random draws are host numpy, consuming numpy's legacy global stream in the reference's draw order so a
given ``--noise-seed`` yields the reference's graph (pinned by tests/golden/dp_adjacency.npz); LapGraph's
O(N^2) add + top-k select run on the GPU when one is visible (``lt_lapgraph_select``, csrc/lt_dp.hip).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def get_noise(noise_type, size, seed, eps=10, delta=1e-5, sensitivity=2):
    """Seeded Laplace / Gaussian noise, reference utils/load.py:27-39."""
    np.random.seed(seed)
    if noise_type == "laplace":
        return np.random.laplace(0, sensitivity / eps, size)
    if noise_type == "gaussian":
        return np.random.normal(0, np.sqrt(2 * np.log(1.25 / delta)) * sensitivity / eps, size)
    raise NotImplementedError("noise {} not implemented!".format(noise_type))


def _symmetric_from_upper(rows, cols, n):
    """0/1 matrix with the (i < j) pairs and their mirror (role of worker.py:178-203)."""
    keep = rows < cols
    m = sp.csr_matrix((np.ones(int(keep.sum()), dtype=np.int64), (rows[keep], cols[keep])), shape=(n, n))
    return m + m.T


def perturb_adj_discrete(adj, epsilon, noise_seed):
    """EdgeRand: every cell is re-drawn with probability s = 2/(e^eps+1); re-drawn cells of the upper
    triangle become 1 or 0 with probability 1/2.  Draw order: one N x N binomial, then one binomial
    per selected cell in row-major order (worker.py:222-233)."""
    s = 2 / (np.exp(epsilon) + 1)
    print(f"s = {s:.4f}")
    n = adj.shape[0]
    np.random.seed(noise_seed)
    rows, cols = np.nonzero(np.random.binomial(1, s, (n, n)))
    coin = np.random.binomial(1, 1 / 2, rows.shape[0])
    add = _symmetric_from_upper(rows[coin == 1], cols[coin == 1], n)
    sub = _symmetric_from_upper(rows[coin == 0], cols[coin == 0], n)
    noisy = adj + add - sub
    noisy.data[noisy.data == -1] = 0          # removed a non-edge: stays absent (explicit zero, as in the reference)
    noisy.data[noisy.data == 2] = 1           # added an existing edge
    return noisy


def _lapgraph_inputs(adj, epsilon, noise_seed, noise_type, delta):
    """The two seeded draws of worker.py:291-304 in the reference's order: the N x N cell noise, then the edge-count noise."""
    n = adj.shape[0]
    n_edges = len(adj.data) // 2
    eps_1 = epsilon * 0.01
    eps_2 = epsilon - eps_1
    noise = get_noise(noise_type, (n, n), noise_seed, eps=eps_2, delta=delta, sensitivity=1)
    n_keep = n_edges + int(get_noise(noise_type, 1, noise_seed, eps=eps_1, delta=delta, sensitivity=1)[0])
    print(f"edge number from {n_edges} to {n_keep}")
    return n, noise, n_keep


def perturb_adj_continuous(adj, epsilon, noise_seed, noise_type="laplace", delta=1e-5, backend="auto"):
    """LapGraph: Laplace(1/eps2) noise on the strict lower triangle, keep the top-(E + noise) cells,
    symmetrise (worker.py:281-335).  eps is split 1 % / 99 % between the edge count and the cells.
    The reference selects the top cells with a 50-way split + argpartition; the selected *set* is the
    plain top-k (ties only among exact zeros, which are never reached).

    backend: "hip" -- the noise (numpy's stream, drawn here) is uploaded and the add + top-k select run on the GPU
    (lt_lapgraph_select); "host" -- numpy argpartition, as the reference; "auto" -- "hip" when a HIP device is visible.
    Both give the same cells (tests/golden/dp_adjacency.npz)."""
    if backend == "auto":
        import torch
        backend = "hip" if torch.cuda.is_available() else "host"
    n, noise, n_keep = _lapgraph_inputs(adj, epsilon, noise_seed, noise_type, delta)
    if backend == "hip":
        top = _lapgraph_select_hip(adj, noise, n_keep)
    elif backend == "host":
        noise *= np.tri(n, n, k=-1, dtype=bool)
        cells = np.asarray(sp.tril(adj, k=-1) + noise).ravel()
        top = np.argpartition(cells, -n_keep)[-n_keep:]
    else:
        raise ValueError(f"backend = {backend!r}")
    mat = sp.csr_matrix((np.ones(n_keep, dtype=np.int32), (top // n, top % n)), shape=(n, n))
    return mat + mat.T


def _lapgraph_select_hip(adj, noise, n_keep):
    """Flat indices of the n_keep largest cells of tril(adj, -1) + noise, selected on the device."""
    import ctypes as C
    import torch
    from . import _lib
    _lib.require_gpu()
    n = adj.shape[0]
    dev = torch.device("cuda", torch.cuda.current_device())
    low = sp.csr_matrix(adj)
    low.sort_indices()
    rowptr = torch.from_numpy(low.indptr.astype(np.int32)).to(dev)
    col = torch.from_numpy(low.indices.astype(np.int32)).to(dev)
    cells = torch.from_numpy(np.ascontiguousarray(noise, dtype=np.float64)).to(dev)
    out = torch.empty(n_keep, dtype=torch.int64, device=dev)
    work = torch.zeros(4096, dtype=torch.uint8, device=dev)
    thr = C.c_double(0.0)
    _lib.check(_lib.lib().lt_lapgraph_select(n, rowptr.data_ptr(), col.data_ptr(), cells.data_ptr(), int(n_keep), out.data_ptr(),
                                             work.data_ptr(), work.numel(), C.byref(thr),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lt_lapgraph_select")
    return out.cpu().numpy()


def perturb_adj(adj, perturb_type, epsilon, noise_seed, noise_type="laplace", delta=1e-5, backend="auto"):
    """Dispatch of worker.py:206-210."""
    if perturb_type == "discrete":
        return perturb_adj_discrete(adj, epsilon, noise_seed)
    return perturb_adj_continuous(adj, epsilon, noise_seed, noise_type, delta, backend=backend)
