"""GPU tests added in round 5: the matrix leaving the device once as float64 (lt_export_rows_f64), node ids checked on the
device (LT_ERR_INDEX / lt_node_check), the drop-in API's hot path (one state_dict walk, cached node lists)."""
import argparse
import ctypes as C
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(w, dev):
    return [torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")]


@pytest.mark.parametrize("rows,cols,ld", [(500, 500, 500), (7, 5, 5), (64, 33, 40), (1, 1, 1), (3, 1, 6), (129, 1000, 1000)])
def test_export_rows_f64_to_pinned_host_and_to_device(gpu, rows, cols, ld):
    """lt_export_rows_f64: dst[i, j] = (double)src[i, j] for odd widths, leading dimensions larger than the row, a device
    destination and a pinned-host destination (written by the kernel over PCIe, valid after the stream wait)."""
    from linkteller_amd import _lib, engine
    src = torch.randn((rows, ld), device=gpu)
    view = src[:, :cols]
    want = view.cpu().numpy().astype(np.float64)
    got = engine.export_rows_f64(view)
    assert got.dtype == np.float64 and got.shape == (rows, cols) and np.array_equal(got, want)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # device destination with its own leading dimension (unaligned rows take the scalar stores)
    dst = torch.full((rows, cols + 3), -1.0, dtype=torch.float64, device=gpu)
    _lib.check(_lib.lib().lt_export_rows_f64(src.data_ptr(), ld, rows, cols, dst.data_ptr(), cols + 3, st))
    torch.cuda.synchronize()
    d = dst.cpu().numpy()
    assert np.array_equal(d[:, :cols], want) and np.all(d[:, cols:] == -1.0)


def test_export_rows_f64_refuses_pageable_memory(gpu):
    from linkteller_amd import _lib
    src = torch.randn((4, 4), device=gpu)
    pageable = np.zeros((4, 4))
    rc = _lib.lib().lt_export_rows_f64(src.data_ptr(), 4, 4, 4, pageable.ctypes.data, 4, None)
    msg = _lib.lib().lt_last_error().lower()
    assert rc == -1 and (b"pinned" in msg or b"pageable" in msg), (rc, msg)
    torch.cuda.synchronize()


def _graphs():
    from linkteller_amd import graph, synth
    er = graph.first_order_gcn(synth.twitch_like_problem("twitch-RU", hidden=32, n_classes=2, seed=0)[0])       # no hub rows: records
    pl = graph.first_order_gcn(synth.powerlaw_graph(1500, 9000, seed=2))                                        # hub rows: item kernels
    return {"er": er, "pl": pl}


@pytest.mark.parametrize("which", ["er", "pl"])
@pytest.mark.parametrize("mode", ["delta", "sparse", "full"])
@pytest.mark.parametrize("bad_list", ["probe", "observed"])
def test_out_of_range_node_id_on_the_device_fast_path(gpu, which, mode, bad_list):
    """int32 CUDA node lists skip the host check (engine._as_nodes); the reference raises IndexError at grad_mat[test_nodes[j]]
    (attacker.py:226-229).  The kernels check the ids themselves: no out-of-bounds access (the call completes), IndexError
    from engine.node_check() after the sync -- or from the NEXT call's entry -- and the handle serves a valid call afterwards,
    bit-equal to one made before the bad call."""
    from linkteller_amd import engine, graph, synth
    a_hat = _graphs()[which]
    n = a_hat.shape[0]
    f, h = 96, 32
    x = torch.from_numpy(synth.twitch_like_features(n, f, seed=3, density=0.05)).to(gpu)
    w = synth.gcn_weights(f, h, 2, seed=4)
    base = engine.Baseline(graph.HipGraph(a_hat), x, *_params(w, gpu))
    rng = np.random.RandomState(5)
    probes = rng.choice(n, 24, replace=False).astype(np.int32)
    obs = rng.choice(n, 40, replace=False).astype(np.int32)
    good = base.influence_rows(torch.from_numpy(probes).to(gpu), torch.from_numpy(obs).to(gpu), 1e-4, mode).clone()
    torch.cuda.synchronize()
    engine.node_check()                                   # nothing pending
    for bad in (n, n + 12345, -1, 2 ** 31 - 1):
        p2, o2 = probes.copy(), obs.copy()
        (p2 if bad_list == "probe" else o2)[7] = bad
        out = base.influence_rows(torch.from_numpy(p2).to(gpu), torch.from_numpy(o2).to(gpu), 1e-4, mode)
        torch.cuda.synchronize()                          # the call ran to completion: nothing faulted
        assert torch.isfinite(out).all()
        with pytest.raises(IndexError):
            engine.node_check()
        engine.node_check()                               # the flag is cleared by the report
    # reported by the next call's entry when nobody asked in between
    p2 = probes.copy()
    p2[0] = n
    base.influence_rows(torch.from_numpy(p2).to(gpu), torch.from_numpy(obs).to(gpu), 1e-4, mode)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        base.influence_rows(torch.from_numpy(probes).to(gpu), torch.from_numpy(obs).to(gpu), 1e-4, mode)
    again = base.influence_rows(torch.from_numpy(probes).to(gpu), torch.from_numpy(obs).to(gpu), 1e-4, mode)
    torch.cuda.synchronize()
    engine.node_check()
    assert torch.equal(again, good)


def test_out_of_range_node_id_three_layers(gpu):
    from linkteller_amd import engine, graph, synth
    n, f = 400, 48
    a_hat = graph.first_order_gcn(synth.powerlaw_graph(n, 1800, seed=2))
    x = torch.from_numpy(synth.gaussian_features(n, f, seed=3)).to(gpu)
    rs = np.random.RandomState(0)
    mk = lambda *s: torch.from_numpy((rs.randn(*s) * 0.2).astype(np.float32)).to(gpu)
    base = engine.Baseline3(graph.HipGraph(a_hat), x, mk(f, 32), mk(32), mk(32, 16), mk(16), mk(16, 2), mk(2))
    nodes = np.arange(20, dtype=np.int32)
    for mode in ("sparse", "delta"):
        good = base.influence_rows(torch.from_numpy(nodes).to(gpu), torch.from_numpy(nodes).to(gpu), 1e-4, mode).clone()
        bad = nodes.copy()
        bad[3] = n + 5
        base.influence_rows(torch.from_numpy(bad).to(gpu), torch.from_numpy(nodes).to(gpu), 1e-4, mode)
        torch.cuda.synchronize()
        with pytest.raises(IndexError):
            engine.node_check()
        assert torch.equal(base.influence_rows(torch.from_numpy(nodes).to(gpu), torch.from_numpy(nodes).to(gpu), 1e-4, mode), good)


def _attacker(gpu, n_test=90, mode="delta", hidden=64):
    from linkteller_amd import graph, synth
    from linkteller_amd.attacker import Attacker
    from linkteller_amd.gcn import GCN
    n, f = 700, 300
    adj = synth.powerlaw_graph(n, 3500, seed=3)
    x = torch.from_numpy(synth.twitch_like_features(n, f, seed=4, density=0.03)).to(gpu)
    a_hat = graph.first_order_gcn(adj)
    adj_t = graph.sparse_mx_to_torch_sparse_tensor(a_hat).to(gpu)
    w = synth.gcn_weights(f, hidden, 2, seed=5)
    model = GCN(f, hidden, 2, 0.5)
    model.load_state_dict({"gc1.weight": torch.from_numpy(w["W1"]), "gc1.bias": torch.from_numpy(w["b1"]),
                           "gc2.weight": torch.from_numpy(w["W2"]), "gc2.bias": torch.from_numpy(w["b2"])})
    model.to(gpu).eval()
    wk = types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=adj.tocsr(), n_nodes=n)
    args = argparse.Namespace(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=n_test, sample_seed=42, influence=1e-4,
                              mode="vanilla-clean", attack_mode="efficient", influence_mode=mode)
    atk = Attacker(args, model, wk)
    atk.prepare_test_data()
    return atk, model, a_hat, x, w


@pytest.mark.parametrize("mode", ["delta", "sparse"])
def test_api_hot_path_matrix_equals_the_primitive_and_is_a_fresh_float64_array(gpu, mode):
    """Attacker.influence_matrix (what `time for predicting edges` brackets, attacker.py:213-231) through the round-5 path:
    float64 [n_test, n_test] on the host, equal to the primitive's rows widened; every call returns an array of its own
    (an earlier result is not overwritten by a later attack); in-place weight updates are seen; a new node list is re-uploaded."""
    from linkteller_amd import engine, graph
    atk, model, a_hat, x, w = _attacker(gpu, mode=mode)
    m1 = atk.influence_matrix()
    assert m1.dtype == np.float64 and m1.shape == (90, 90) and m1.flags["C_CONTIGUOUS"]
    base = engine.Baseline(graph.HipGraph(a_hat), x, *_params(w, gpu))
    nodes = np.asarray(atk.test_nodes)
    want = base.influence_rows(nodes, nodes, 1e-4, mode).cpu().numpy().astype(np.float64)
    assert np.array_equal(m1, want)
    keep = m1.copy()
    m2 = atk.influence_matrix()
    assert m2 is not m1 and not np.shares_memory(m1, m2) and np.array_equal(m1, keep) and np.array_equal(m2, keep)
    # an in-place weight update (same storage) must be seen by the next attack
    with torch.no_grad():
        model.gc2.weight.mul_(2.0)
    m3 = atk.influence_matrix()
    assert np.array_equal(m1, keep)                       # the old array still holds the old scores
    assert not np.array_equal(m3, keep) and np.abs(m3 - 2.0 * keep).max() <= 1e-3 * keep.max() + (0 if mode == "delta" else 0.05 * keep.max())
    # another node list: the cached device lists are keyed by content
    atk.test_nodes = nodes[::-1].copy()
    m4 = atk.influence_matrix()
    assert np.array_equal(m4, m3[::-1, ::-1])


def test_api_out_of_range_test_node_raises_index_error(gpu):
    atk, *_ = _attacker(gpu)
    atk.test_nodes = np.asarray(atk.test_nodes).copy()
    atk.test_nodes[5] = 700
    with pytest.raises(IndexError):
        atk.influence_matrix()


@pytest.mark.parametrize("mode", ["delta", "sparse"])
def test_wide_baseline_probe_chunks_keep_every_bit(gpu, mode, monkeypatch):
    """engine.WideBaseline serves the probes in chunks so that the slices' difference vectors stay inside a fixed budget
    (ADVICE r4); a budget of a few probes per chunk must give the bits of one chunk."""
    from linkteller_amd import engine, graph, synth
    n, f, h, c = 300, 80, 320, 12
    a_hat = graph.first_order_gcn(synth.powerlaw_graph(n, 1400, seed=3))
    x = torch.from_numpy(synth.gaussian_features(n, f, seed=4)).to(gpu)
    w = synth.gcn_weights(f, h, c, seed=5)
    rng = np.random.RandomState(1)
    probes, obs = rng.choice(n, 37, replace=False), rng.choice(n, 50, replace=False)
    base = engine.baseline_for(graph.HipGraph(a_hat), x, *_params(w, gpu))
    assert isinstance(base, engine.WideBaseline)
    whole = base.influence_rows(probes, obs, 1e-4, mode).clone()
    monkeypatch.setattr(engine.WideBaseline, "VEC_BUDGET_BYTES", 2 * 50 * 8 * 4 * 5)      # 5 probes per chunk (2 hidden slices)
    base2 = engine.baseline_for(graph.HipGraph(a_hat), x, *_params(w, gpu))
    chunked = base2.influence_rows(probes, obs, 1e-4, mode)
    assert base2._buf["key"][0] == 5 and torch.equal(chunked, whole)


@pytest.mark.parametrize("bits", ["per-probe", "big-probes-only"])
def test_observed_hub_search_forms_keep_every_bit(gpu, bits):
    """The observed-hub blocks of large calls choose per (probe, hub) pair between the probe's own two-level search (up to four
    rounds), the whole block searching the probe's R_v, and the whole block passing over the row through a bitmap (round 5,
    DESIGN 5.2b).  A power-law graph whose biggest hubs are BOTH probed and observed, next to hubs of under and over 1024 entries
    (rows that sit in LDS whole / sampled rows), drives all three; with the short-side forms forced on (`hub_short_side` = 1) and
    off the matrices must be equal bit for bit, in `delta` and in `sparse`, with a bitmap row per probe and with rows for the big
    probes only -- and `delta` must meet the fp64 oracle."""
    from linkteller_amd import _lib, engine, graph, synth
    from oracle import linkteller_oracle as O
    n = 12000
    a_hat = graph.first_order_gcn(synth.powerlaw_graph(n, 90000, seed=11, exponent=1.7))
    deg = np.diff(a_hat.indptr)
    order = np.argsort(-deg)
    assert deg[order[0]] > 2048 and (deg > 1024).sum() >= 2 and ((deg > 128) & (deg <= 1024)).sum() >= 10
    f, h, c = 48, 64, 2
    x = synth.gaussian_features(n, f, seed=3)
    w = synth.gcn_weights(f, h, c, seed=4)
    rng = np.random.RandomState(2)
    mid = np.flatnonzero((deg > 128) & (deg <= 1024))
    probes = np.concatenate([order[:5], rng.choice(mid, 6, replace=False), rng.choice(n, 120, replace=False)])
    obs = np.concatenate([order[:4], rng.choice(mid, 12, replace=False), rng.choice(n, 300, replace=False)])
    assert (deg[probes] > 256).sum() >= 5                      # probes the block searches for (more than four rounds of their own)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu))
    knobs = ("hub_short_side", "bits_max_bytes", "pair_marks")
    try:
        _lib.set_tuning("pair_marks", -1)
        _lib.set_tuning("bits_max_bytes", (1 << 27) if bits == "per-probe" else 1)
        got = {}
        for short in (0, 1):
            _lib.set_tuning("hub_short_side", short)
            for m in ("delta", "sparse"):
                got[(short, m)] = base.influence_rows(probes, obs, 1e-4, m).clone()
        for m in ("delta", "sparse"):
            assert torch.equal(got[(0, m)], got[(1, m)]), (bits, m, float((got[(0, m)] - got[(1, m)]).abs().max()))
    finally:
        for k in knobs:
            _lib.set_tuning(k, None)
    # the fp64 oracle on the hub probes' rows
    P64 = {k: torch.from_numpy(w[k]).double() for k in ("W1", "b1", "W2", "b2")}
    adj_o, x64 = O.to_torch_sparse(a_hat).double(), torch.from_numpy(x).double()
    d = got[(1, "delta")].cpu().numpy().astype(np.float64)
    with torch.no_grad():
        for i in (0, 1, 4, 7, 20):
            r64 = O.get_gradient_eps_mat(x64, adj_o, P64, int(probes[i]), 1e-4)[torch.as_tensor(obs.astype(np.int64))].norm(dim=1).numpy()
            assert np.abs(d[i] - r64).max() <= 1e-5 * max(r64.max(), 1e-6), (i, np.abs(d[i] - r64).max(), r64.max())
