"""Host-side graph preparation and the device-resident graph handle.

Mirrors, for the hot path only, what the reference does between reading an adjacency and
handing it to the model:

* ``fetch_normalization(name)`` -- the six ``--norm`` choices of reference utils/load.py:562-627
  (``FirstOrderGCN`` = ``I + D^-1/2 A D^-1/2`` is the one BASELINE.json uses).  Computed on the
  CSR arrays in the reference's multiplication order ``(d_i * a_ij) * d_j`` and in the dtype numpy
  promotion gives the reference (float32 for a float32 adjacency), so the float32 values handed
  to the device are the reference's bit for bit.
* ``sparse_mx_to_torch_sparse_tensor`` (reference utils/load.py:552-559) -- kept for API parity;
  the product path converts to int32 CSR instead (8 B/nnz rather than the reference's 20 B/nnz
  int64 COO) and uploads it once through ``lt_graph_create``.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np
import scipy.sparse as sp

from . import _lib


# ----------------------------------------------------------------------------------------------
# normalisers (host; dtype follows numpy promotion from the input, as in the reference)
# ----------------------------------------------------------------------------------------------
def _canonical_csr(adj) -> sp.csr_matrix:
    # dtype is deliberately NOT forced: the reference's arithmetic follows numpy promotion from
    # the adjacency's dtype (float32 for the MUSAE readers, utils/load.py:456-460 -> the whole
    # normalisation runs in float32; integer for the DP-perturbed graphs -> float64).
    a = sp.csr_matrix(adj, copy=True)
    if a.dtype == np.bool_:
        a = a.astype(np.int64)
    a.sum_duplicates()
    a.sort_indices()
    return a


def _with_identity(a: sp.csr_matrix) -> sp.csr_matrix:
    out = _canonical_csr(a + sp.identity(a.shape[0], dtype=np.float64, format="csr"))  # float64 from here
    return out


def _inv_power(row_sum: np.ndarray, power: float, zero_inf: bool) -> np.ndarray:
    with np.errstate(divide="ignore"):
        d = np.power(row_sum, power)
    if zero_inf:
        d[np.isinf(d)] = 0.0
    return d


def _scale(a: sp.csr_matrix, left: np.ndarray, right=None) -> sp.csr_matrix:
    """(diag(left) @ a) @ diag(right), entry-wise in that association."""
    rows = np.repeat(np.arange(a.shape[0]), np.diff(a.indptr))
    data = left[rows] * a.data
    if right is not None:
        data = data * right[a.indices]
    return sp.csr_matrix((data, a.indices.copy(), a.indptr.copy()), shape=a.shape)


def _row_sums(a: sp.csr_matrix) -> np.ndarray:
    return np.asarray(a.sum(axis=1)).ravel()


def first_order_gcn(adj):
    """``FirstOrderGCN``: I + D^-1/2 A D^-1/2 (reference utils/load.py:572-578)."""
    a = _canonical_csr(adj)
    d = _inv_power(_row_sums(a), -0.5, True)
    return _canonical_csr(sp.identity(a.shape[0], dtype=np.float64, format="csr") + _scale(a, d, d))


def aug_normalized_adjacency(adj):
    """``AugNormAdj``: (D+I)^-1/2 (A+I) (D+I)^-1/2 (reference utils/load.py:562-569)."""
    a = _with_identity(_canonical_csr(adj))
    d = _inv_power(_row_sums(a), -0.5, True)
    return _canonical_csr(_scale(a, d, d))


def bingge_norm_adjacency(adj):
    """``BingGeNormAdj``: (D+I)^-1/2 (A+I) (D+I)^-1/2 + I (reference utils/load.py:581-588)."""
    return _canonical_csr(aug_normalized_adjacency(adj) + sp.identity(adj.shape[0], dtype=np.float64, format="csr"))


def normalized_adjacency(adj):
    """``NormAdj``: D^-1/2 A D^-1/2 (reference utils/load.py:591-597)."""
    a = _canonical_csr(adj)
    d = _inv_power(_row_sums(a), -0.5, True)
    return _canonical_csr(_scale(a, d, d))


def random_walk(adj):
    """``RWalk``: D^-1 A (reference utils/load.py:600-605; infinities are NOT zeroed there)."""
    a = _canonical_csr(adj)
    return _canonical_csr(_scale(a, _inv_power(_row_sums(a), -1.0, False)))


def aug_random_walk(adj):
    """``AugRWalk``: (D+I)^-1 (A+I) (reference utils/load.py:608-614)."""
    a = _with_identity(_canonical_csr(adj))
    return _canonical_csr(_scale(a, _inv_power(_row_sums(a), -1.0, False)))


_NORMALIZERS = {
    "FirstOrderGCN": first_order_gcn,
    "BingGeNormAdj": bingge_norm_adjacency,
    "NormAdj": normalized_adjacency,
    "AugRWalk": aug_random_walk,
    "RWalk": random_walk,
    "AugNormAdj": aug_normalized_adjacency,
}


def fetch_normalization(name: str):
    """Same lookup contract as reference utils/load.py:617-627."""
    try:
        return _NORMALIZERS[name]
    except KeyError:
        raise NotImplementedError(f"normalization {name!r} not implemented") from None


def sparse_mx_to_torch_sparse_tensor(sparse_mx):
    """scipy -> torch sparse COO, float32 values / int64 indices (reference utils/load.py:552-559)."""
    import torch
    m = sp.coo_matrix(sparse_mx).astype(np.float32)
    idx = torch.from_numpy(np.vstack((m.row, m.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(m.data), torch.Size(m.shape))


# ----------------------------------------------------------------------------------------------
# device graph handle
# ----------------------------------------------------------------------------------------------
def csr_arrays(mat):
    """(n, rowptr int32, col int32, val float32) of a square sparse matrix, canonical form."""
    a = sp.csr_matrix(mat)
    if a.shape[0] != a.shape[1]:
        raise ValueError(f"adjacency must be square, got {a.shape}")
    a = a.astype(np.float32)       # the reference rounds to f32 BEFORE any duplicate could be summed
    a.sum_duplicates()
    a.sort_indices()
    if a.nnz >= 2**31 - 1:
        raise ValueError("nnz does not fit int32 row pointers")
    return (a.shape[0], np.ascontiguousarray(a.indptr, dtype=np.int32),
            np.ascontiguousarray(a.indices, dtype=np.int32), np.ascontiguousarray(a.data, dtype=np.float32))


class HipGraph:
    """Normalised adjacency resident in HBM as CSR (+ CSC) behind an ``lt_graph`` handle."""

    def __init__(self, mat):
        _lib.require_gpu()
        n, rowptr, col, val = csr_arrays(mat)
        h = C.c_void_p()
        _lib.check(_lib.lib().lt_graph_create(n, int(col.shape[0]), rowptr.ctypes.data, col.ctypes.data,
                                              val.ctypes.data, C.byref(h)), "lt_graph_create")
        import torch
        self._h = h
        self.n = n
        self.nnz = int(col.shape[0])
        self.device_index = torch.cuda.current_device()      # lt_graph_create uploads to the current device
        self._finalizer = weakref.finalize(self, _lib.lib().lt_graph_destroy, h)

    @property
    def handle(self):
        return self._h

    @property
    def max_row_nnz(self) -> int:
        m = C.c_int32()
        _lib.check(_lib.lib().lt_graph_info(self._h, None, None, C.byref(m)), "lt_graph_info")
        return m.value

    @classmethod
    def from_torch_sparse(cls, t):
        """Accepts what the reference feeds its model: an (uncoalesced) sparse COO float tensor."""
        t = t.detach().cpu().coalesce()
        idx = t.indices().numpy()
        m = sp.coo_matrix((t.values().numpy().astype(np.float32), (idx[0], idx[1])), shape=tuple(t.shape))
        return cls(m)


_GRAPH_CACHE: "dict[int, tuple]" = {}


def as_hip_graph(adj) -> HipGraph:
    """HipGraph for whatever the caller holds (HipGraph / torch sparse / scipy), cached per object."""
    if isinstance(adj, HipGraph):
        return adj
    key = id(adj)
    hit = _GRAPH_CACHE.get(key)
    if hit is not None and hit[0]() is adj:
        return hit[1]
    import torch
    if isinstance(adj, torch.Tensor):
        if not adj.is_sparse:
            raise TypeError("adj must be a sparse tensor, a scipy sparse matrix or a HipGraph")
        g = HipGraph.from_torch_sparse(adj)
    else:
        g = HipGraph(adj)
    try:
        ref = weakref.ref(adj, lambda _r, k=key: _GRAPH_CACHE.pop(k, None))
        _GRAPH_CACHE[key] = (ref, g)
    except TypeError:
        pass
    return g
