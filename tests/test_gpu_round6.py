"""Round-6 GPU tests: the generic (> 3 layers / wide GCN3) path through the drop-in API, the ring form of the product rows,
the hub-row sharding of large calls."""
import argparse
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, csr_from

pytestmark = pytest.mark.gpu


def _stack_reference(adj_hat, x, layers, probes, observe, delta):
    """attacker.py:100-108, 220-229 for a GraphConvolution stack of any depth, in fp64 (gcn/layers.py:30-36 per layer,
    relu between the layers as gcn/models.py:19-24, 39-45 place it)."""
    import scipy.sparse as sp
    a = sp.csr_matrix(adj_hat).astype(np.float64)

    def forward(xx):
        h = xx
        for li, (w, b) in enumerate(layers):
            h = a @ (h @ w) + b
            if li < len(layers) - 1:
                h = np.maximum(h, 0.0)
        return h

    base = forward(x)
    out = np.zeros((len(probes), len(observe)))
    for i, v in enumerate(probes):
        xp = x.copy()
        xp[v] = x[v] + x[v] * delta
        out[i] = np.linalg.norm(((forward(xp) - base) / delta)[observe], axis=1)
    return out


def test_generic_stack_through_influence_matrix(gpu):
    """Attacker.influence_matrix() hands _rows() the cached int32 DEVICE node lists; a model on the generic
    path (four layers here) reached _rows_generic with them and np.asarray raised.  The API result must equal the direct
    numpy-list call bit for bit and sit in the fp32 finite difference's noise class of the fp64 evaluation."""
    from linkteller_amd import graph
    from linkteller_amd.attacker import Attacker
    from linkteller_amd.gcn import GraphConvolution

    g = np.load(os.path.join(GOLDEN, "next_rows.npz"), allow_pickle=False)
    a = csr_from(g, "adj")
    x = torch.from_numpy(g["x"]).to(gpu)
    adj_hat = graph.first_order_gcn(a)
    adj_t = graph.sparse_mx_to_torch_sparse_tensor(adj_hat).to(gpu)
    w = types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=a, n_nodes=a.shape[0])

    class Stack4(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(5)
            self.gc1 = GraphConvolution(x.shape[1], 24)
            self.gc2 = GraphConvolution(24, 16)
            self.gc3 = GraphConvolution(16, 12)
            self.gc4 = GraphConvolution(12, 3)

        def forward(self, xx, adj):
            h = self.gc1(xx, adj, relu=True)
            h = self.gc2(h, adj, relu=True)
            h = self.gc3(h, adj, relu=True)
            return self.gc4(h, adj)

    model = Stack4().to(gpu).eval()
    args = argparse.Namespace(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=16, sample_seed=42,
                              influence=1e-4, mode="vanilla-clean", attack_mode="efficient")
    atk = Attacker(args, model, w)
    atk.prepare_test_data()
    assert atk._walk()[0] == "generic"
    got = atk.influence_matrix()                      # device node lists -> _rows -> _rows_generic
    nodes = np.asarray(atk.test_nodes, dtype=np.int64)
    direct = atk._rows_generic(nodes, nodes).cpu().numpy().astype(np.float64)
    assert got.dtype == np.float64 and got.shape == (16, 16)
    assert np.array_equal(got, direct)
    layers = [(getattr(model, f"gc{i}").weight.detach().cpu().numpy().astype(np.float64),
               getattr(model, f"gc{i}").bias.detach().cpu().numpy().astype(np.float64)) for i in (1, 2, 3, 4)]
    ref = _stack_reference(adj_hat, g["x"].astype(np.float64), layers, nodes, nodes, 1e-4)
    # fp32 finite difference at delta = 1e-4: absolute noise ~ ulp(logit) / 1e-4, a few 1e-3 of the largest score
    assert np.abs(got - ref).max() <= 3e-2 * ref.max(), np.abs(got - ref).max() / ref.max()
    with pytest.raises(IndexError):
        atk._rows_generic(np.array([0, a.shape[0]]), nodes)


def _params(w, dev):
    return [torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")]


@pytest.mark.parametrize("f,h,hub", [(3170, 256, False), (2300, 64, False), (3170, 128, True)])
def test_ring_and_flag_forms_of_the_product_rows(gpu, f, h, hub):
    """k_s1d_feature_ring (persistent LDS-ring form, "feature_ring") and the flag-bit list of k_s1d_feature_rows ("feature_flags")
    against the default form of the feature-difference product: `delta` within 1e-5 of the fp64 oracle on every form, the
    forms within 1e-6 of the largest score of each other (they differ in fp64 summation order only), repeatable bit for bit,
    dense rows among the sparse ones served (and the route retired by the hint), the record route / the item kernels behind it."""
    from test_gpu_parity import _oracle_matrix
    from linkteller_amd import _lib, engine, graph, synth
    n = 1300
    adj = synth.powerlaw_graph(n, 6000, seed=3) if hub else synth.erdos_renyi_graph(n, 5000, seed=3)
    a_hat = graph.first_order_gcn(adj)
    x = synth.twitch_like_features(n, f, seed=6, density=0.006)
    x[5, ::7] = 3.25                      # a row with ~450 differing columns: beyond a wave's list (the piecewise path), sets the hint
    x[n - 1, f - 1] = 2.5                 # the last value of the matrix; the first one
    x[0, 0] = -1.5
    w = synth.gcn_weights(f, h, 2, seed=7)
    hg = graph.HipGraph(a_hat)
    xt = torch.from_numpy(x).to(gpu)
    rng = np.random.RandomState(1)
    probes = np.concatenate([[n - 1, 0, 5], rng.choice(np.arange(6, n - 1), 29, replace=False)])
    observe = np.concatenate([[n - 1, 0, 5], rng.choice(np.arange(6, n - 1), 150, replace=False)])
    ref64 = _oracle_matrix(a_hat, x, w, probes[:8], observe, 1e-4, torch.float64)

    def run(**knobs):
        for k, v in knobs.items():
            _lib.set_tuning(k, v)
        try:
            base = engine.Baseline(hg, xt, *_params(w, gpu))
            _lib.set_tuning("feature_delta", 1)          # (the dense row would retire the route at the probe of enable_fp64)
            base.enable_fp64()
            assert base.fp64_route() == 1
            a = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
            base.refresh()
            b = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
            assert np.array_equal(a, b)                  # a refresh re-forms the rows: the same bits (claimed rows, any wave)
            return a
        finally:
            for k in list(knobs) + ["feature_delta"]:
                _lib.set_tuning(k, None)

    plain = run()
    ring = run(feature_ring=1)
    flags = run(feature_flags=1)
    both = run(feature_ring=1, defer_cref=0)
    top = ref64.max()
    for name, got in (("plain", plain), ("ring", ring), ("flags", flags), ("ring, cref first", both)):
        assert np.abs(got[:8] - ref64).max() <= 1e-5 * top, name
        assert np.abs(got - plain).max() <= 1e-6 * top, name
        assert np.all(got[:8][ref64 == 0] == 0), name


@pytest.mark.parametrize("graph_kind,mode", [("er", "delta"), ("er", "sparse"), ("pl", "delta"), ("er", "full")])
def test_influence_rows_f64_equals_rows_plus_export(gpu, graph_kind, mode):
    """lt_influence_rows_f64 (the probes' blocks of the fused `delta` route write their own float64 rows into pinned host memory;
    every other route ends with the export launch) gives the bits of lt_influence_rows + lt_export_rows_f64 -- rows of an odd
    and an even width (16-byte and 8-byte host stores), several probe chunks, device and pinned destinations, and through
    Attacker.influence_matrix()."""
    from linkteller_amd import _lib, engine, graph, synth
    n, f, h = 700, 96, 64
    adj = synth.erdos_renyi_graph(n, 3000, seed=1) if graph_kind == "er" else synth.powerlaw_graph(n, 3000, seed=1)
    hg = graph.HipGraph(graph.first_order_gcn(adj))
    x = torch.from_numpy(synth.twitch_like_features(n, f, seed=2, density=0.05)).to(gpu)
    w = synth.gcn_weights(f, h, 2, seed=3)
    base = engine.Baseline(hg, x, *_params(w, gpu))
    rng = np.random.RandomState(4)
    for n_probe, n_obs in ((40, 57), (64, 64), (1, 3)):
        probes = rng.choice(n, n_probe, replace=False)
        obs = rng.choice(n, n_obs, replace=False)
        want = engine.export_rows_f64(base.influence_rows(probes, obs, 1e-4, mode))
        got = base.influence_matrix_host(probes, obs, 1e-4, mode)
        assert got.dtype == np.float64 and got.shape == (n_probe, n_obs)
        assert np.array_equal(got, want)
        dev64 = torch.full((n_probe, n_obs), -1.0, dtype=torch.float64, device=gpu)
        base.influence_rows(probes, obs, 1e-4, mode, host=dev64)
        assert np.array_equal(dev64.cpu().numpy(), want)
    # several probe chunks (a small scratch budget): the chunks' blocks write disjoint row ranges
    probes = rng.choice(n, 300, replace=False)
    obs = rng.choice(n, 301, replace=False)
    want = engine.export_rows_f64(base.influence_rows(probes, obs, 1e-4, mode))
    _lib.set_tuning("chunk_budget_bytes", 1 << 20)
    try:
        got = base.influence_matrix_host(probes, obs, 1e-4, mode)
    finally:
        _lib.set_tuning("chunk_budget_bytes", None)
    assert np.array_equal(got, want)
    with pytest.raises(IndexError):
        base.influence_matrix_host(torch.tensor([0, n], dtype=torch.int32, device=gpu), obs, 1e-4, mode)
    engine.node_check()


def test_probe_sharding_auto_never_slower_than_one_gpu(gpu, tmp_path):
    """LT_SHARD_PROBES=auto (default) under LT_FORCE_COLLECTIVES=1: bench.py times 'shard + all-gather' against 'every rank builds
    all rows' through the group's collectives (RCCL at world size 1) and takes the faster; on the n_test = 500-shaped `delta`
    build -- whose step is its loop-invariant baseline -- that is the local build, the line says so, and the matrix equals the
    plain run's bit for bit."""
    from test_gpu_round4 import _bench
    common = ["--steps", "5", "--warmup", "1", "--blocks", "3", "--no-cpu-baseline", "--no-extras", "--no-pmc", "--no-scaling-workloads",
              "--no-api-wall", "--mode", "delta", "--n-test", "200"]
    plain = _bench(common, {"LT_BENCH_DUMP": str(tmp_path / "plain.npy")})
    auto = _bench(common, {"LT_FORCE_COLLECTIVES": "1", "LT_BENCH_DUMP": str(tmp_path / "auto.npy")})
    ps = auto["config"]["probe_sharding"]
    assert ps["policy"] == "auto" and ps["sharded"] is False, ps
    assert ps["ms_per_step_every_rank_all_rows"] < ps["ms_per_step_sharded_all_gather"], ps
    assert auto["config"]["collective_bytes_per_step"] == 0
    assert np.array_equal(np.load(tmp_path / "plain.npy"), np.load(tmp_path / "auto.npy"))
    print(f"plain {plain['ms_per_step']} ms, forced collectives + auto {auto['ms_per_step']} ms per step")
    if os.environ.get("LT_ASSERT_TIMINGS"):
        assert auto["ms_per_step"] <= 1.10 * plain["ms_per_step"]


def test_on_demand_preactivation_by_row_list(gpu):
    """lt_graph_reached_rows + lt_baseline_form / gather / scatter_rows_fp64 (the row-list form of the aggregate-first route): the hub
    rows a probe list reaches, formed ahead of the probe call, packed, and adopted from a buffer -- the matrix keeps every bit
    of the plain call's; rows stay valid until the next refresh (the call skips them); other routes refuse."""
    import scipy.sparse as sp
    from linkteller_amd import _lib, engine, graph, synth
    n, f, h = 3000, 96, 64
    adj = synth.powerlaw_graph(n, 20000, seed=5)
    a_hat = graph.first_order_gcn(adj)
    hg = graph.HipGraph(a_hat)
    x = torch.from_numpy(synth.gaussian_features(n, f, seed=6)).to(gpu)
    w = synth.gcn_weights(f, h, 2, seed=7)
    rng = np.random.RandomState(8)
    obs = rng.choice(n, 400, replace=False)
    probes = obs[:100]
    _lib.set_tuning("aggregate_first", 1)
    try:
        base = engine.Baseline(hg, x, *_params(w, gpu)).enable_fp64()
        assert base.fp64_route() == 2
        plain = base.influence_rows(probes, obs, 1e-4, "delta").clone()
        # the reached rows against scipy
        thr = 40
        rows = base.reached_rows(obs, thr)
        csc = sp.csc_matrix(a_hat)
        reach = np.unique(np.concatenate([csc.indices[csc.indptr[v]:csc.indptr[v + 1]] for v in obs]))
        lens = np.diff(sp.csr_matrix(a_hat).indptr)
        want_rows = reach[lens[reach] >= thr]
        assert len(want_rows) > 10 and np.array_equal(rows.cpu().numpy(), want_rows)
        hp = (h + 3) // 4 * 4
        buf = torch.full((rows.numel(), hp), float("nan"), dtype=torch.float64, device=gpu)
        base.refresh("delta")
        base.form_rows_fp64(rows)
        base.gather_rows_fp64(rows, buf)
        assert bool(torch.isfinite(buf).all())
        a = base.influence_rows(probes, obs, 1e-4, "delta")          # (the hub rows are valid: only the others are formed)
        assert torch.equal(a, plain)
        # adopt the rows from the buffer instead of forming them (ids out of range in the list are skipped)
        base.refresh("delta")
        padded = torch.cat([rows, torch.tensor([-1, n, -5], dtype=torch.int32, device=gpu)])
        bufp = torch.cat([buf, torch.full((3, hp), float("nan"), dtype=torch.float64, device=gpu)])
        base.scatter_rows_fp64(padded, bufp)
        b = base.influence_rows(probes, obs, 1e-4, "delta")
        assert torch.equal(b, plain)
        # a share formed here, the rest adopted: what dist.SharedHubRows does on one of several ranks
        base.refresh("delta")
        k = rows.numel() // 3
        base.form_rows_fp64(rows[:k].contiguous())
        base.scatter_rows_fp64(rows[k:].contiguous(), buf[k:].contiguous())
        assert torch.equal(base.influence_rows(probes, obs, 1e-4, "delta"), plain)
        # poisoned rows DO reach the result (the call really takes them from the buffer)
        base.refresh("delta")
        base.scatter_rows_fp64(rows, -buf)               # (the kink test reads the sign: a scaled row would change nothing)
        assert not torch.equal(base.influence_rows(probes, obs, 1e-4, "delta"), plain)
        base.refresh("delta")
    finally:
        _lib.set_tuning("aggregate_first", None)
    _lib.set_tuning("aggregate_first", 0)
    try:
        base2 = engine.Baseline(hg, x, *_params(w, gpu)).enable_fp64()
        assert base2.fp64_route() != 2
        with pytest.raises(RuntimeError):
            base2.form_rows_fp64(rows)
    finally:
        _lib.set_tuning("aggregate_first", None)


@pytest.mark.parametrize("world", [2, 3])
def test_hub_rows_shared_between_ranks_keep_every_bit(gpu, tmp_path, world):
    """dist.SharedHubRows through the product path (Attacker.influence_matrix on the on-demand route, several ranks as processes on
    one device over gloo): the ranks split the hub rows all probes reach, one all-gather moves them, and the matrix equals the
    single-rank one bit for bit."""
    import textwrap
    from test_gpu_round2 import _run_ranks
    from linkteller_amd import _lib, engine, graph, synth
    code = textwrap.dedent('''
        import argparse, os, types, numpy as np, torch
        from linkteller_amd import graph, synth, main as lt_main, dist as lt_dist
        from linkteller_amd.attacker import Attacker
        from linkteller_amd.gcn import GCN
        assert lt_main.init_distributed()
        import torch.distributed as dist
        n, f, h = 3000, 96, 64
        adj = synth.powerlaw_graph(n, 20000, seed=5)
        x = torch.from_numpy(synth.gaussian_features(n, f, seed=6)).cuda()
        w = synth.gcn_weights(f, h, 2, seed=7)
        model = GCN(f, h, 2, 0.5)
        model.load_state_dict({"gc1.weight": torch.from_numpy(w["W1"]), "gc1.bias": torch.from_numpy(w["b1"]),
                               "gc2.weight": torch.from_numpy(w["W2"]), "gc2.bias": torch.from_numpy(w["b2"])})
        model.cuda().eval()
        adj_t = graph.sparse_mx_to_torch_sparse_tensor(graph.first_order_gcn(adj)).cuda()
        wk = types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=adj.tocsr(), n_nodes=n)
        ns = argparse.Namespace(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=301, sample_seed=42, influence=1e-4,
                                mode="vanilla-clean", attack_mode="efficient", influence_mode="delta")
        atk = Attacker(ns, model, wk)
        atk.test_nodes = np.random.RandomState(8).choice(n, 301, replace=False)
        m1 = atk.influence_matrix()
        m2 = atk.influence_matrix()                      # (a second attack: the row list is cached, the rows are exchanged again)
        hub = atk._hub_rows[1]
        assert hub.n_rows > 10 and hub.per * dist.get_world_size() >= hub.n_rows
        assert np.array_equal(m1, m2)
        if dist.get_rank() == 0:
            np.save(os.environ["LT_TEST_OUT"], m1)
        dist.barrier(); dist.destroy_process_group()
    ''')
    n, f, h = 3000, 96, 64
    adj = synth.powerlaw_graph(n, 20000, seed=5)
    hg = graph.HipGraph(graph.first_order_gcn(adj))
    x = torch.from_numpy(synth.gaussian_features(n, f, seed=6)).to(gpu)
    w = synth.gcn_weights(f, h, 2, seed=7)
    nodes = np.random.RandomState(8).choice(n, 301, replace=False)
    _lib.set_tuning("aggregate_first", 1)
    try:
        base = engine.Baseline(hg, x, *_params(w, gpu)).enable_fp64()
        assert base.fp64_route() == 2
        single = base.influence_rows(nodes, nodes, 1e-4, "delta").cpu().numpy().astype(np.float64)
    finally:
        _lib.set_tuning("aggregate_first", None)
    out = tmp_path / f"hub{world}.npy"
    _run_ranks(code, world, {"LT_TEST_OUT": str(out), "LT_AGGREGATE_FIRST": "1", "LT_HUB_ROW_MIN_ENTRIES": "40", "LT_SHARD_PROBES": "1"})
    assert np.array_equal(np.load(out), single)


def test_rmat_scale21_config5_per_rank_shape(gpu):
    """BASELINE configs[4] at FULL size in the default set (the 150-s `slow` test was the only one at 2 M nodes):
    R-MAT scale 21 (2 097 152 nodes, nnz(A_hat) ~ 77 M, a hub row of > 10^5 entries), F = H = 256, the per-rank shape of the
    config (512 probes x 4096 observed, the three biggest hubs on both sides) in `delta` and `sparse`: 4 probe rows (the biggest
    hub, a mid-degree probe, two random ones) against oracle.RestrictedOracle (pinned to the verbatim op sequence by
    tests/test_oracle_golden.py), exact zeros off the 2-hop set, the two modes within the fp32 difference's noise of each other,
    and the hub rows shared between ranks (dist.SharedHubRows' three calls) keeping every bit.  The verbatim-oracle rows stay
    in the `slow` variant (tests/test_gpu_round2.py)."""
    import time
    import scipy.sparse as sp
    from linkteller_amd import dist as lt_dist, engine, graph, synth
    from oracle import linkteller_oracle as O
    t0 = time.time()
    adj = synth.rmat_graph(21, synth.rmat_draws(21), seed=42)
    a_hat = graph.first_order_gcn(adj)
    n = adj.shape[0]
    deg = np.diff(a_hat.indptr)
    assert n == 1 << 21 and a_hat.nnz > 70_000_000 and deg.max() > 50_000
    x_np = synth.gaussian_features(n, 256, seed=1)
    w = synth.gcn_weights(256, 256, 2, seed=42)
    x = torch.from_numpy(x_np).to(gpu)
    base = engine.Baseline(graph.HipGraph(a_hat), x, *_params(w, gpu))
    t_build = time.time() - t0
    rng = np.random.RandomState(3)
    hubs = np.argsort(-deg)[:3].astype(np.int64)
    rest = rng.choice(np.setdiff1d(np.arange(n), hubs), 4096 - len(hubs), replace=False)
    observe = np.concatenate([hubs, rest])
    probes = observe[:512]
    t0 = time.time()
    delta = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy()
    assert base.fp64_route() == 2                                   # the on-demand route: what this size takes
    sparse = base.influence_rows(probes, observe, 1e-4, "sparse").cpu().numpy()
    t_gpu = time.time() - t0
    assert np.isfinite(delta).all() and np.isfinite(sparse).all()
    # exact zeros off the 2-hop set, both modes (the mask: pattern product restricted to the probes' columns and the observed rows)
    pat = sp.csr_matrix((np.ones(a_hat.nnz, np.float32), a_hat.indices, a_hat.indptr), shape=a_hat.shape)
    r1 = pat.T.tocsr()[probes]
    mask = np.asarray((r1 @ pat[observe].T.tocsc()).todense()) > 0
    assert np.all(delta[~mask] == 0) and np.all(sparse[~mask] == 0)
    assert (delta[mask] > 0).mean() > 0.9
    assert np.abs(sparse - delta).max() <= 0.05 * delta.max()
    # the fp64 oracle on 4 rows
    pdeg = deg[probes]
    rows = np.unique(np.concatenate([[0, int(np.argsort(pdeg)[len(pdeg) // 2])], rng.choice(512, 2, replace=False)]))
    t0 = time.time()
    ro = O.RestrictedOracle(x_np, a_hat, w)
    ref_rows = ro.rows(probes[rows], observe, 1e-4)
    t_oracle = time.time() - t0
    for k, i in enumerate(rows):
        ref64 = ref_rows[k]
        assert np.abs(delta[i] - ref64).max() <= 1e-5 * max(ref64.max(), 1e-3), (int(i), int(probes[i]))
        assert np.all(delta[i][ref64 == 0] == 0) and np.all(sparse[i][ref64 == 0] == 0) and np.all(mask[i][ref64 > 0])
    del ro
    # the hub rows every rank's probes reach, split over 8 ranks: this GPU forms one share, adopts the rest -- same bits
    t0 = time.time()
    rows_all = base.reached_rows(observe, lt_dist.HUB_ROW_MIN_ENTRIES)
    nh = rows_all.numel()
    assert nh > 1000 and bool((torch.from_numpy(deg).to(gpu)[rows_all.long()] >= lt_dist.HUB_ROW_MIN_ENTRIES).all())
    buf = torch.empty((nh, 256), dtype=torch.float64, device=gpu)
    base.refresh("delta"); base.form_rows_fp64(rows_all); base.gather_rows_fp64(rows_all, buf)
    per8 = (nh + 7) // 8
    base.refresh("delta")
    base.form_rows_fp64(rows_all[3 * per8:4 * per8].contiguous())      # (rank 3's share)
    base.scatter_rows_fp64(rows_all, buf)
    shared = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy()
    assert np.array_equal(shared, delta)
    print(f"scale 21: build {t_build:.1f} s, delta + sparse {t_gpu:.2f} s, restricted oracle on rows {rows.tolist()} {t_oracle:.1f} s, "
          f"{nh} shared hub rows {time.time() - t0:.2f} s")


@pytest.mark.parametrize("share,share2", [(35, 15), (100, 0), (40, 60), (1, 1), (0, 50)])
def test_host_matrix_with_zero_filled_head_equals_the_dense_export(gpu, share, share2):
    """lt_influence_rows_f64 behind a refresh, fused `delta` route: the head of the float64 matrix is zero-filled by blocks riding in
    the product rows' launch and in the pre-activation's, and its probes' blocks write the touched positions only
    ("export_sparse"); same matrix as rows +
    lt_export_rows_f64 -- row widths around the pair stores, a dirty pinned buffer used twice without a host-side wait, device
    destinations, several probe chunks, the knob off."""
    from linkteller_amd import _lib, engine, graph, synth
    n, f, h = 1200, 96, 64
    adj = synth.erdos_renyi_graph(n, 5000, seed=1)
    hg = graph.HipGraph(graph.first_order_gcn(adj))
    x = torch.from_numpy(synth.twitch_like_features(n, f, seed=2, density=0.05)).to(gpu)
    w = synth.gcn_weights(f, h, 2, seed=3)
    base = engine.Baseline(hg, x, *_params(w, gpu))
    rng = np.random.RandomState(4)
    _lib.set_tuning("export_zero_share", share)
    _lib.set_tuning("export_zero_share2", share2)
    try:
        for nob in (1, 7, 64, 257, 1000):
            obs = torch.from_numpy(rng.choice(n, nob, replace=False).astype(np.int32)).to(gpu)
            probes = obs[: max(1, nob // 2)].contiguous()
            npb = probes.numel()
            base.refresh("delta")
            want = engine.export_rows_f64(base.influence_rows(probes, obs, 1e-4, "delta"))
            got = base.influence_matrix_host(probes, obs, 1e-4, "delta", refresh=True)
            assert got.dtype == np.float64 and np.array_equal(got, want), nob
            host = torch.full((npb, nob), 7.0, dtype=torch.float64).pin_memory()
            out = torch.empty((npb, nob), dtype=torch.float32, device=gpu)
            for _ in range(2):
                base.refresh("delta")
                base.influence_rows(probes, obs, 1e-4, "delta", out=out, host=host)
            torch.cuda.synchronize()
            assert np.array_equal(host.numpy(), want), (nob, "reused buffer")
            dev64 = torch.full((npb, nob), -1.0, dtype=torch.float64, device=gpu)
            base.refresh("delta")
            base.influence_rows(probes, obs, 1e-4, "delta", out=out, host=dev64)
            assert np.array_equal(dev64.cpu().numpy(), want), (nob, "device matrix")
            # an odd leading dimension (8-byte stores)
            wide = torch.full((npb, nob + 1), -1.0, dtype=torch.float64, device=gpu)
            base.refresh("delta")
            ws = base._ws[next(iter(base._ws))]
            _lib.check(_lib.lib().lt_influence_rows_f64(base._h, probes.data_ptr(), npb, obs.data_ptr(), nob, 1e-4, _lib.MODE_DELTA,
                                                        out.data_ptr(), nob, wide.data_ptr(), nob + 1, ws.data_ptr(), ws.numel(),
                                                        engine._stream()), "lt_influence_rows_f64")
            wide = wide.cpu().numpy()
            assert np.array_equal(wide[:, :nob], want) and np.all(wide[:, nob] == -1.0), (nob, "ld = cols + 1")
        # several probe chunks
        obs = torch.from_numpy(rng.choice(n, 301, replace=False).astype(np.int32)).to(gpu)
        probes = obs[:300].contiguous()
        base.refresh("delta")
        want = engine.export_rows_f64(base.influence_rows(probes, obs, 1e-4, "delta"))
        _lib.set_tuning("chunk_budget_bytes", 1 << 20)
        try:
            got = base.influence_matrix_host(probes, obs, 1e-4, "delta", refresh=True)
        finally:
            _lib.set_tuning("chunk_budget_bytes", None)
        assert np.array_equal(got, want)
        _lib.set_tuning("export_sparse", 0)
        try:
            got = base.influence_matrix_host(probes, obs, 1e-4, "delta", refresh=True)
        finally:
            _lib.set_tuning("export_sparse", None)
        assert np.array_equal(got, want)
    finally:
        _lib.set_tuning("export_zero_share", None)
        _lib.set_tuning("export_zero_share2", None)


def test_zero_filled_head_never_overtakes_the_values(gpu):
    """The zeros of the host matrix's head are written by launches that END before the probes' launch begins, its values by that
    launch: 3 000 host-landed builds at twitch-RU size into buffers that held 7.0 everywhere, alternating between two probe sets
    (so a stale value of the previous build cannot pass for the present one) -- every matrix equals the widened device matrix."""
    from linkteller_amd import engine, graph, synth
    n, f, h = 4385, 3170, 256
    hg = graph.HipGraph(graph.first_order_gcn(synth.erdos_renyi_graph(n, 37304, seed=42)))
    x = torch.from_numpy(synth.twitch_like_features(n, f, seed=1)).to(gpu)
    w = synth.gcn_weights(f, h, 2, seed=42)
    base = engine.Baseline(hg, x, *_params(w, gpu))
    rng = np.random.RandomState(7)
    sets = []
    for _ in range(2):
        nodes = torch.from_numpy(rng.choice(n, 500, replace=False).astype(np.int32)).to(gpu)
        base.refresh("delta")
        sets.append((nodes, engine.export_rows_f64(base.influence_rows(nodes, nodes, 1e-4, "delta"))))
    bufs = [torch.empty((500, 500), dtype=torch.float64).pin_memory() for _ in range(3)]
    out = torch.empty((500, 500), dtype=torch.float32, device=gpu)
    cur = torch.cuda.current_stream(gpu)
    bad = 0
    for it in range(3000):
        nodes, want = sets[it & 1]
        host = bufs[it % 3]
        host.fill_(7.0)
        base.refresh("delta")
        base.influence_rows(nodes, nodes, 1e-4, "delta", out=out, host=host)
        cur.synchronize()
        bad += int(not np.array_equal(host.numpy(), want))
    assert bad == 0, f"{bad} of 3000 host matrices differ"


@pytest.mark.parametrize("mode", ["sparse", "delta"])
def test_marked_pairs_as_a_list_give_the_per_pair_bits(gpu, mode):
    """"pair_list": calls that find their affected pairs by the join over the middle nodes compact the marks into a list, zero-fill the
    rows and walk the list 8 pairs to a wave -- the matrix of the form in which every pair's lane group reads its own mark, and of
    the bitmap route; hub rows on both sides, several chunks, a leading dimension wider than the row."""
    from linkteller_amd import _lib, engine, graph, synth
    n, f, h = 3000, 128, 64
    adj = synth.powerlaw_graph(n, 30000, seed=5)
    hg = graph.HipGraph(graph.first_order_gcn(adj))
    deg = np.diff(graph.first_order_gcn(adj).tocsr().indptr)
    x = torch.from_numpy(synth.twitch_like_features(n, f, seed=2, density=0.05)).to(gpu)
    w = synth.gcn_weights(f, h, 2, seed=3)
    base = engine.Baseline(hg, x, *_params(w, gpu))
    rng = np.random.RandomState(4)
    hubs = np.argsort(-deg)[:5]
    obs = np.unique(np.concatenate([hubs, rng.choice(n, 700, replace=False)])).astype(np.int32)
    probes = np.concatenate([hubs[:2], rng.choice(n, 150, replace=False)]).astype(np.int32)

    def run(**knobs):
        for k, v in knobs.items():
            _lib.set_tuning(k, v)
        try:
            base.refresh(mode)
            return base.influence_rows(probes, obs, 1e-4, mode).cpu().numpy()
        finally:
            for k in knobs:
                _lib.set_tuning(k, None)

    want = run(pair_marks=-1)                                   # no marks: the bitmap route
    for knobs in ({"pair_marks": 0, "pair_list": 0}, {"pair_marks": 0, "pair_list": 1}, {"pair_marks": 0, "pair_list": 1, "item_bits": 0},
                  {"pair_marks": 0, "pair_list": 1, "chunk_budget_bytes": 1 << 16}):
        assert np.array_equal(run(**knobs), want), knobs
    assert (want > 0).any() and (want == 0).any()
    # a result matrix wider than the row: the columns past n_obs are left alone
    wide = torch.full((len(probes), len(obs) + 3), -1.0, dtype=torch.float32, device=gpu)
    _lib.set_tuning("pair_marks", 0)
    try:
        base.refresh(mode)
        pt = torch.from_numpy(probes).to(gpu)
        ot = torch.from_numpy(obs).to(gpu)
        ws = engine._workspace(_lib.lib().lt_influence_workspace_bytes(base._h, len(probes), len(obs), _lib.MODES[mode]), gpu)
        _lib.check(_lib.lib().lt_influence_rows(base._h, pt.data_ptr(), len(probes), ot.data_ptr(), len(obs), 1e-4, _lib.MODES[mode],
                                                wide.data_ptr(), len(obs) + 3, ws.data_ptr(), ws.numel(), engine._stream()), "lt_influence_rows")
    finally:
        _lib.set_tuning("pair_marks", None)
    wide = wide.cpu().numpy()
    assert np.array_equal(wide[:, :len(obs)], want) and np.all(wide[:, len(obs):] == -1.0)


@pytest.mark.parametrize("n,f,h,kind", [(700, 300, 64, "gaussian"), (1500, 1000, 256, "gaussian"), (1500, 1000, 128, "outliers")])
def test_dense_product_on_the_int8_cores_against_the_f64_cores(gpu, n, f, h, kind):
    """"i8_split": the fp64 product of dense features as an error-free integer split on the int8 matrix cores -- `delta` within 1e-6 of the
    largest score of the f64 cores' matrix and within 1e-5 of the fp64 oracle (features with 60-sigma outliers included: they set
    a row's scale), the same bits run to run and whichever row range of the product a launch forms (the sharded refresh)."""
    from linkteller_amd import _lib, engine, graph, synth
    from test_gpu_parity import _oracle_matrix
    adj = synth.erdos_renyi_graph(n, 6 * n, seed=1)
    a_hat = graph.first_order_gcn(adj)
    hg = graph.HipGraph(a_hat)
    x = synth.gaussian_features(n, f, seed=2)
    if kind == "outliers":
        rs = np.random.RandomState(3)
        x[rs.randint(0, n, 200), rs.randint(0, f, 200)] = 60.0
    w = synth.gcn_weights(f, h, 2, seed=3)
    rng = np.random.RandomState(4)
    obs = rng.choice(n, 120, replace=False)
    probes = obs[:40]

    def run(i8):
        _lib.set_tuning("i8_split", i8)
        _lib.set_tuning("aggregate_first", 0)
        try:
            base = engine.Baseline(hg, torch.from_numpy(x).to(gpu), *_params(w, gpu))
            outs = [base.influence_rows(probes, obs, 1e-4, "delta").cpu().numpy().astype(np.float64)]
            base.refresh("delta")
            outs.append(base.influence_rows(probes, obs, 1e-4, "delta").cpu().numpy().astype(np.float64))
            assert base.fp64_route() == 0 and np.array_equal(outs[0], outs[1])
            return base, outs[0]
        finally:
            _lib.set_tuning("i8_split", None)
            _lib.set_tuning("aggregate_first", None)

    _, f64 = run(0)
    base, i8 = run(1)
    ref64 = _oracle_matrix(a_hat, x, w, probes, obs, 1e-4, torch.float64)
    scale = ref64.max()
    assert np.abs(i8 - f64).max() <= 1e-6 * scale, np.abs(i8 - f64).max() / scale
    assert np.abs(i8 - ref64).max() <= 1e-5 * scale and np.abs(f64 - ref64).max() <= 1e-5 * scale
    assert np.all(i8[ref64 == 0] == 0)
    # the product formed in row ranges (what a rank of a sharded refresh does): the rows have the bits of the whole product's
    _lib.set_tuning("i8_split", 1)
    _lib.set_tuning("aggregate_first", 0)
    try:
        hp = (h + 3) // 4 * 4
        whole = torch.empty((n, hp), dtype=torch.float64, device=gpu)
        _lib.check(_lib.lib().lt_baseline_refresh_rows_fp64(base._h, 0, n, whole.data_ptr(), engine._stream()), "rows_fp64")
        parts = torch.empty((n, hp), dtype=torch.float64, device=gpu)
        cuts = [0, 77, 640, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            _lib.check(_lib.lib().lt_baseline_refresh_rows_fp64(base._h, a, b, parts[a:].data_ptr(), engine._stream()), "rows_fp64")
        assert torch.equal(whole, parts)
        host = x.astype(np.float64) @ w["W1"].astype(np.float64)
        got = whole.cpu().numpy()[:, :h]
        rowmax = np.abs(host).max(axis=1, keepdims=True)
        assert (np.abs(got - host) / rowmax).max() <= 2e-9
    finally:
        _lib.set_tuning("i8_split", None)
        _lib.set_tuning("aggregate_first", None)
        base.refresh()


def test_attacker_sees_every_kind_of_parameter_change(gpu):
    """Attacker.influence_matrix() between attacks: an in-place update (the baseline's refresh reads the borrowed tensors), a
    parameter's storage replaced (`p.data = ...`), a parameter object replaced, a submodule replaced, a load_state_dict -- every
    one gives the matrix of a fresh Attacker on the same weights (the cached state_dict walk and the baseline key's fast path
    must not outlive what they cached)."""
    import argparse
    import contextlib
    import io
    import types
    from linkteller_amd import graph, synth
    from linkteller_amd.attacker import Attacker
    from linkteller_amd.gcn import GCN, GraphConvolution
    n, f, h, c = 600, 80, 32, 3
    adj = synth.erdos_renyi_graph(n, 2400, seed=1)
    a_hat = graph.first_order_gcn(adj)
    x = torch.from_numpy(synth.twitch_like_features(n, f, seed=2, density=0.05)).to(gpu)
    adj_t = graph.sparse_mx_to_torch_sparse_tensor(a_hat).to(gpu)
    wk = types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=adj.tocsr(), n_nodes=n)
    args = argparse.Namespace(dataset="twitch/RU", sample_type="unbalanced", n_test=40, sample_seed=42, influence=1e-4,
                              mode="vanilla-clean", attack_mode="efficient", influence_mode="delta")

    def attacker(model):
        atk = Attacker(args, model, wk)
        with contextlib.redirect_stdout(io.StringIO()):
            atk.prepare_test_data()
        return atk

    torch.manual_seed(0)
    model = GCN(f, h, c, 0.5).to(gpu).eval()
    atk = attacker(model)

    def fresh():
        twin = GCN(f, h, c, 0.5).to(gpu).eval()
        twin.load_state_dict(model.state_dict())
        return attacker(twin).influence_matrix()

    m0 = atk.influence_matrix()
    assert np.array_equal(m0, fresh()) and np.array_equal(m0, atk.influence_matrix())
    steps = []
    with torch.no_grad():
        model.gc2.weight.mul_(0.5)                                          # in place
    steps.append("in place")
    m1 = atk.influence_matrix()
    assert np.array_equal(m1, fresh()) and not np.array_equal(m1, m0)
    model.gc1.weight.data = (model.gc1.weight.data * 1.25).clone()          # the parameter's storage replaced
    m2 = atk.influence_matrix()
    assert np.array_equal(m2, fresh()) and not np.array_equal(m2, m1)
    model.gc2.weight = torch.nn.Parameter(torch.randn_like(model.gc2.weight) * 0.1)     # the parameter object replaced
    m3 = atk.influence_matrix()
    assert np.array_equal(m3, fresh()) and not np.array_equal(m3, m2)
    new_gc1 = GraphConvolution(f, h).to(gpu)                                # a submodule replaced
    model.gc1 = new_gc1
    m4 = atk.influence_matrix()
    assert np.array_equal(m4, fresh()) and not np.array_equal(m4, m3)
    sd = {k: torch.randn_like(v) * 0.1 for k, v in model.state_dict().items()}
    model.load_state_dict(sd)                                               # load_state_dict (copies in place)
    m5 = atk.influence_matrix()
    assert np.array_equal(m5, fresh()) and not np.array_equal(m5, m4)
    assert np.array_equal(m5, atk.influence_matrix())
