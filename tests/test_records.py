"""The incidence records of `delta`'s fused route (lt_graph_create builds them; lt_graph_records_host exposes the same host code
without a device) against a plain restatement: node v's items are the rows r of column v of A_hat, and every entry (u, r) of a
row u that holds an item is an incidence (A_hat[u, r], item, position of the entry in row u) -- grouped by u ascending, a node's
entries in entry order.  CPU only."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp


def _records(lt, a, capacity=None, want_rec=True):
    a = sp.csr_matrix(a, dtype=np.float32)
    a.sort_indices()
    n = a.shape[0]
    rp, ci, va = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float32)
    meta = np.zeros(4 * n, dtype=np.int32)
    words = C.c_int64(-1)
    h = lt.lib()
    rc = h.lt_graph_records_host(n, a.nnz, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, meta.ctypes.data, None, 0, C.byref(words))
    if rc != 0 or not want_rec:
        return rc, meta.reshape(n, 4), None, words.value
    cap = words.value if capacity is None else capacity
    rec = np.zeros(max(cap, 1), dtype=np.int32)
    rc = h.lt_graph_records_host(n, a.nnz, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, meta.ctypes.data, rec.ctypes.data, cap,
                                 C.byref(words))
    return rc, meta.reshape(n, 4), rec, words.value


def _restate(a):
    """(items, nodes, entries) per node, straight from the definition."""
    a = sp.csr_matrix(a, dtype=np.float32)
    a.sort_indices()
    csc = a.tocsc()
    csc.sort_indices()
    n = a.shape[0]
    out = []
    for v in range(n):
        rows = csc.indices[csc.indptr[v]:csc.indptr[v + 1]]
        vals = csc.data[csc.indptr[v]:csc.indptr[v + 1]]
        items = [(int(r), np.float32(x)) for r, x in zip(rows, vals)]
        inc = []
        for item, (r, _) in enumerate(items):
            for u in csc.indices[csc.indptr[r]:csc.indptr[r + 1]]:
                row_cols = a.indices[a.indptr[u]:a.indptr[u + 1]]
                k = int(np.searchsorted(row_cols, r))
                assert row_cols[k] == r
                inc.append((int(u), k, item, np.float32(a.data[a.indptr[u] + k])))
        inc.sort(key=lambda t: (t[0], t[1]))
        nodes, entries = [], []
        for u, k, item, x in inc:
            if not nodes or nodes[-1][0] != u:
                nodes.append([u, len(entries), 0])
            nodes[-1][2] += 1
            entries.append((x, (item << 16) | k))
        out.append((items, nodes, entries))
    return out


def _bits(x):
    return int(np.float32(x).view(np.int32))


@pytest.fixture(scope="module")
def lt():
    from linkteller_amd import _lib
    import os
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


@pytest.mark.parametrize("seed,directed", [(1, False), (2, False), (3, True)])
def test_records_match_the_definition(lt, seed, directed):
    rng = np.random.RandomState(seed)
    n = 180
    a = sp.random(n, n, density=0.03, random_state=rng, format="lil", dtype=np.float32)
    if not directed:
        a = a + a.T
    a = sp.lil_matrix(a)
    a.setdiag(1.0)
    a[7, :] = 0          # an empty row ...
    a[:, 11] = 0         # ... and an empty column (node 11 has no items)
    a = sp.csr_matrix(a)
    a.eliminate_zeros()
    a.data = rng.uniform(0.05, 0.9, a.nnz).astype(np.float32)
    rc, meta, rec, words = _records(lt, a)
    assert rc == 0, lt.lib().lt_last_error()
    want = _restate(a)
    at = 0
    for v, (items, nodes, entries) in enumerate(want):
        off, cnt, tu, t = (int(x) for x in meta[v])
        assert (off, cnt, tu, t) == (at, len(items), len(nodes), len(entries)), v
        r = rec[off: off + 2 * (cnt + tu + t)].reshape(-1, 2)
        assert [(int(p), int(q)) for p, q in r[:cnt]] == [(rr, _bits(x)) for rr, x in items]
        assert [(int(p), int(q)) for p, q in r[cnt:cnt + tu]] == [(u, s | (c << 16)) for u, s, c in nodes]
        assert [(int(p), int(q)) for p, q in r[cnt + tu:]] == [(_bits(x), ik) for x, ik in entries]
        at += 2 * (cnt + tu + t)
    assert words == at
    assert meta[11, 1] == 0 and meta[11, 3] == 0


def test_graphs_without_records_and_argument_errors(lt):
    h = lt.lib()
    n = 400
    ring = sp.lil_matrix((n, n), dtype=np.float32)
    for i in range(n):
        ring[i, i] = 1
        ring[i, (i + 1) % n] = 1
        ring[(i + 1) % n, i] = 1
    assert _records(lt, ring)[0] == 0
    hub = ring.copy()
    for i in range(1, 140):                     # a row of more than 128 entries: a hub row
        hub[0, i] = 1
        hub[i, 0] = 1
    rc = _records(lt, hub, want_rec=False)[0]
    assert rc == -3 and b"no incidence records" in h.lt_last_error()
    dense = ring.copy()
    k = 72                                       # a 72-clique: 72 x ~74 incidences per member, beyond the cap of 4096
    for u in range(k):
        for v in range(k):
            dense[u, v] = 1
    assert _records(lt, dense, want_rec=False)[0] == -3
    small = ring.copy()
    for u in range(20):
        for v in range(20):
            small[u, v] = 1
    rc, meta, rec, words = _records(lt, small)
    assert rc == 0 and meta[:, 3].max() <= 4096 and meta[5, 3] >= 20 * 20
    # too little room, NULL arguments
    assert _records(lt, small, capacity=words - 2)[0] == -4
    assert h.lt_graph_records_host(4, 0, None, None, None, None, None, 0, None) == -1
