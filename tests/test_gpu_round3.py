"""GPU tests added in round 3: the unbalanced LSA2 attacks on the device forward, bench.py at N = 2 through its own
launcher, layers wider than one pass of the row kernels, the sharded / on-demand fp64 pre-activation, the device top-k
of LapGraph."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, csr_from, load_golden, noise_gate

pytestmark = pytest.mark.gpu


def _params(w, dev):
    return [torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")]


def test_unbalanced_lsa2_attacks_on_device(gpu, tmp_path, monkeypatch):
    """attacker.py:287-334 (`baseline` = LSA2-post, `baseline-feat` = LSA2-attr) with sample_type `unbalanced`: the
    posteriors come from the HIP forward (GCN.forward -> lt_gemm_f32 / lt_spmm_csr_f32), the correlation math runs as in
    the reference; scores against the reference's own (tests/golden/next_rows.npz), file name and schema included."""
    import argparse
    import types
    from linkteller_amd import graph
    from linkteller_amd.attacker import Attacker
    from linkteller_amd.gcn import GCN
    g = load_golden("next_rows.npz")
    a = csr_from(g, "adj")
    x = torch.from_numpy(g["x"]).to(gpu)
    adj_t = graph.sparse_mx_to_torch_sparse_tensor(graph.first_order_gcn(a)).to(gpu)
    w = types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=a, n_nodes=a.shape[0])
    sd = {k: torch.from_numpy(g[f"sd.{k}"]) for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias")}
    model = GCN(x.shape[1], sd["gc1.weight"].shape[1], sd["gc2.weight"].shape[1], 0.5)
    model.load_state_dict(sd)
    model.to(gpu).eval()
    monkeypatch.chdir(tmp_path)
    for mode in ("baseline", "baseline-feat"):
        args = argparse.Namespace(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=40, sample_seed=42,
                                  influence=1e-4, mode="vanilla-clean", attack_mode=mode)
        atk = Attacker(args, model, w)
        atk.prepare_test_data()
        assert np.array_equal(np.asarray(atk.test_nodes), g[f"{mode}.test_nodes"])
        atk.baseline_attack()
        saved = torch.load(str(g[f"{mode}.filename"]), weights_only=False)
        assert set(saved) == {"auc", "pr", "result"}
        ref = np.concatenate([g[f"{mode}.norm_exist"], g[f"{mode}.norm_nonexist"]])
        got = np.asarray(saved["result"]["pred"])
        assert got.shape == ref.shape
        # correlations of fp32 posteriors: our logits differ from torch's by fp32 rounding (2e-5 relative, tested in
        # test_forward_logits); with C = 2 the centred softmax posteriors are collinear, so the correlations are +-1 and
        # only a sign flip of a near-zero centred vector could move one -- none does on the fixture
        assert np.abs(got - ref).max() <= (2e-5 if mode == "baseline" else 2e-6), mode
        if mode == "baseline-feat":
            assert abs(atk.auc - float(g[f"{mode}.auc"])) <= 1e-4 and abs(atk.ap - float(g[f"{mode}.ap"])) <= 1e-4


def _bench(args, env=None, timeout=900):
    e = dict(os.environ, PYTHONPATH=REPO, **(env or {}))
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("mode,extra", [
    ("delta", {}),
    # (90 / 110 s each: the sharded fp64 and fp32 products through the launcher -- LT_RUN_SLOW=1 / tools/round_artifacts.sh; the
    # default set keeps the plain `delta` launch above and the RCCL-at-world-size-1 runs of tests/test_gpu_round4.py)
    pytest.param("delta", {"LT_FEATURE_DELTA": "0", "LT_AGGREGATE_FIRST": "0", "LT_SHARD_BASELINE": "1"}, marks=pytest.mark.slow),
    pytest.param("full", {"LT_SHARD_BASELINE": "1"}, marks=pytest.mark.slow)])
def test_bench_two_ranks_through_its_own_launcher(gpu, tmp_path, mode, extra):
    """`python bench.py --gpus 2` starts its two ranks itself (fresh child processes, before anything touches the GPU).
    On a 1-GPU box the ranks share device 0 over gloo (LT_BENCH_BACKEND / LT_BENCH_DEVICE): one parsed JSON line with
    n_gpus == 2, a non-zero collective, and the matrix of the last step equal to the one-rank run's bit for bit -- with the
    loop-invariant product replicated, sharded (fp32 X W1 for `full`; the fp64 product for `delta` on dense-feature
    routing), or served by the feature-difference route."""
    common = ["--steps", "3", "--warmup", "1", "--blocks", "2", "--no-cpu-baseline", "--no-extras", "--no-pmc", "--mode", mode,
              "--n-test", "120"]
    one = _bench(common, {"LT_BENCH_DUMP": str(tmp_path / "one.npy"), **extra})
    two = _bench(common + ["--gpus", "2"], {"LT_BENCH_BACKEND": "gloo", "LT_BENCH_DEVICE": "0", "LT_SHARD_PROBES": "1",
                                            "LT_BENCH_DUMP": str(tmp_path / "two.npy"), **extra})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["probes_per_rank"] == 60 and two["config"]["collective_bytes_per_step"] >= 120 * 120 * 4
    assert two["scaling"] == "strong" and two["value"] > 0 and len(two["timing"]["block_ms"]) == 2
    if extra.get("LT_SHARD_BASELINE") == "1":
        assert "sharded" in two["config"]["baseline_XW1"], two["config"]
        assert two["config"]["collective_bytes_per_step"] > 120 * 120 * 4
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert a.shape == (120, 120) and np.array_equal(a, b)


@pytest.mark.parametrize("ncols", [260, 300, 510, 512, 1000, 12, 9])
def test_spmm_any_width(gpu, ncols):
    """lt_spmm_csr_f32 has no width limit (gcn/layers.py:30-36): 256-column slices on the vector path, tails and
    unaligned widths through the 8-lane kernel; against an fp64 host product, on a graph with hub rows."""
    from test_gpu_parity import _hub_graph
    from linkteller_amd import engine, graph
    a_hat = graph.first_order_gcn(_hub_graph(1200, 6000, 700, seed=5))
    rng = np.random.RandomState(ncols)
    s = rng.standard_normal((a_hat.shape[0], ncols)).astype(np.float32)
    b = rng.standard_normal(ncols).astype(np.float32)
    got = engine.spmm(graph.HipGraph(a_hat), torch.from_numpy(s).to(gpu), torch.from_numpy(b).to(gpu), relu=True).cpu().numpy()
    want = np.maximum(a_hat.astype(np.float64) @ s.astype(np.float64) + b, 0)
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("h,c", [(512, 16), (300, 2), (64, 121)])
def test_layers_wider_than_the_fused_kernels(gpu, h, c):
    """--hidden 512 (main.py:30 has no limit) and class counts beyond 8 (ppi: 121): GCN.forward runs on the unfused HIP layers,
    the probe primitive slice by slice (engine.WideBaseline): logits and influence rows, all three modes, against the oracle
    (fp64 and fp32)."""
    from test_gpu_parity import _oracle_matrix
    from linkteller_amd import engine, graph, synth
    from linkteller_amd.gcn import GCN
    from oracle import linkteller_oracle as O
    n, f = 220, 40
    a_hat = graph.first_order_gcn(synth.powerlaw_graph(n, 700, seed=h))
    x = synth.gaussian_features(n, f, seed=3)
    w = synth.gcn_weights(f, h, c, seed=c)
    hg = graph.HipGraph(a_hat)
    base = engine.baseline_for(hg, torch.from_numpy(x).to(gpu), *_params(w, gpu))
    assert isinstance(base, engine.WideBaseline)
    rng = np.random.RandomState(1)
    probes = np.concatenate([rng.choice(n, 9, replace=False), [3, 3]])
    observe = rng.choice(n, 40, replace=False)
    ref64 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float64)
    ref32 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float32)
    # one batched call per matrix in every mode (round 4: the model is served per slice of <= 256 hidden units / <= 8 classes
    # through lt_influence_rows_vec + lt_wide_combine; round 3 looped ~5 launches per probe and had no `delta`)
    res = {m: base.influence_rows(probes, observe, 1e-4, m).cpu().numpy().astype(np.float64) for m in ("sparse", "full", "delta")}
    got = res["sparse"]
    assert np.array_equal(res["full"], got)
    e32 = np.abs(ref32 - ref64).max()
    noise_gate(f"wide.h{h}c{c}.sparse", np.abs(got - ref64).max() / max(e32, 1e-4 * ref64.max()))
    assert np.abs(res["delta"] - ref64).max() <= 1e-5 * ref64.max(), np.abs(res["delta"] - ref64).max() / ref64.max()
    for r in res.values():
        assert np.all(r[ref64 == 0] == 0)
        assert np.array_equal(r[-1], r[-2])
    # the borrowed weights change in place: refresh() re-cuts the slices
    w2_dev = base.w2
    keep = w2_dev.clone()
    w2_dev.mul_(0.5)
    base.refresh()
    half = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    assert np.abs(half - 0.5 * res["delta"]).max() <= 2e-5 * ref64.max()
    w2_dev.copy_(keep)
    base.refresh()
    assert np.array_equal(base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64), res["delta"])
    # (round 6) a refresh FOR `delta` forms no fp32 product; the fp32 modes behind it recompute what they read, W1 changed in place too
    w1_dev = base.w1
    keep1 = w1_dev.clone()
    w1_dev.mul_(0.75)
    base.refresh("delta")
    d75 = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    s75 = base.influence_rows(probes, observe, 1e-4, "sparse").cpu().numpy().astype(np.float64)
    fresh = engine.baseline_for(hg, base.x, w1_dev, base.b1, base.w2, base.b2)
    assert np.array_equal(s75, fresh.influence_rows(probes, observe, 1e-4, "sparse").cpu().numpy().astype(np.float64))
    assert np.array_equal(d75, fresh.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64))
    w1_dev.copy_(keep1)
    base.refresh("delta")
    assert np.array_equal(base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64), res["delta"])
    base.refresh()
    assert np.array_equal(base.influence_rows(probes, observe, 1e-4, "sparse").cpu().numpy().astype(np.float64), res["sparse"])
    P64 = {k: torch.from_numpy(w[k]).double() for k in w}
    ref_logits = O.gcn_forward(torch.from_numpy(x).double(), O.to_torch_sparse(a_hat).double(), P64).numpy()
    tol = 2e-5 * max(1.0, np.abs(ref_logits).max())
    assert np.abs(base.logits().cpu().numpy() - ref_logits).max() <= tol
    model = GCN(f, h, c, 0.5)
    model.load_state_dict({"gc1.weight": torch.from_numpy(w["W1"]), "gc1.bias": torch.from_numpy(w["b1"]),
                           "gc2.weight": torch.from_numpy(w["W2"]), "gc2.bias": torch.from_numpy(w["b2"])})
    model.to(gpu).eval()
    with torch.no_grad():
        out = model(torch.from_numpy(x).to(gpu), hg).cpu().numpy()
    assert np.abs(out - ref_logits).max() <= tol
    assert np.abs(engine.gcn2_forward(hg, torch.from_numpy(x).to(gpu), *_params(w, gpu)).cpu().numpy() - ref_logits).max() <= tol


def test_feature_difference_route_of_the_fp64_product(gpu):
    """The fp64 product X W1 of `delta` from the feature rows' differences to a reference row (lt_fp64.hip,
    k_s1d_feature_rows): taken for standardised indicator features (what the reference's twitch loader produces), not for
    Gaussian ones; `delta` within 1e-5 of the fp64 oracle on both routes, the two routes within fp32 rounding of each
    other; and a baseline whose features BECOME dense after the route was chosen falls back on the device (the gate) to
    the matrix-core product from the next refresh on, giving the bits of a baseline created on the dense features."""
    from test_gpu_parity import _oracle_matrix
    from linkteller_amd import _lib, engine, graph, synth
    n, f, h = 900, 700, 102            # H needs padding (Hp = 104): the pad columns of S1d must stay zero on every route
    a_hat = graph.first_order_gcn(synth.powerlaw_graph(n, 4000, seed=2))
    x = synth.twitch_like_features(n, f, seed=4, density=0.02)
    w = synth.gcn_weights(f, h, 3, seed=5)
    hg = graph.HipGraph(a_hat)
    xt = torch.from_numpy(x).to(gpu)
    base = engine.Baseline(hg, xt, *_params(w, gpu)).enable_fp64()
    assert base.fp64_route() == 1
    rng = np.random.RandomState(0)
    # (the last and the first node on both sides: the last row of X is read through a shifted window, lt_fp64.hip)
    probes = np.concatenate([[n - 1, 0], rng.choice(np.arange(1, n - 1), 38, replace=False)])
    observe = np.concatenate([[n - 1, 0], rng.choice(np.arange(1, n - 1), 198, replace=False)])
    ref64 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float64)
    got = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    assert np.abs(got - ref64).max() <= 1e-5 * ref64.max()
    _lib.set_tuning("feature_delta", 0)
    try:
        base.refresh()
        assert base.fp64_route() == 0
        dense = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    finally:
        _lib.set_tuning("feature_delta", None)
    assert np.abs(dense - ref64).max() <= 1e-5 * ref64.max()
    assert np.abs(dense - got).max() <= 1e-6 * ref64.max()
    # the reference vector's product formed first and added by the rows kernel, instead of riding in its launch
    _lib.set_tuning("defer_cref", 0)
    try:
        base.refresh()
        early = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    finally:
        _lib.set_tuning("defer_cref", None)
        base.refresh()
    assert np.abs(early - ref64).max() <= 1e-5 * ref64.max() and np.abs(early - got).max() <= 1e-6 * ref64.max()
    # dense features from the start: the probe at enable_fp64 picks the matrix cores
    xg = synth.gaussian_features(n, f, seed=9)
    bg = engine.Baseline(hg, torch.from_numpy(xg).to(gpu), *_params(w, gpu)).enable_fp64()
    assert bg.fp64_route() == 0
    want = bg.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy()
    # features that turn dense under a baseline that chose the sparse route: the refresh that meets them is still served by
    # the feature kernel (full lists are walked and emptied: correct, just slow) and its hint word moves the baseline to
    # the matrix cores from the next refresh on -- then the bits are those of a baseline created on the dense features
    xt.copy_(torch.from_numpy(xg).to(gpu))
    base.refresh()
    assert base.fp64_route() == 1
    slow = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    refg = _oracle_matrix(a_hat, xg, w, probes[:6], observe, 1e-4, torch.float64)
    assert np.abs(slow[:6] - refg).max() <= 1e-5 * refg.max()
    assert np.abs(slow - want).max() <= 1e-6 * refg.max()
    torch.cuda.synchronize()
    base.refresh()
    assert base.fp64_route() == 0
    assert np.array_equal(base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy(), want)
    # a single dense row among sparse ones is served too (and flips the route the same way)
    base2 = engine.Baseline(hg, xt, *_params(w, gpu))
    xm = x.copy()
    xm[17] = xg[17]
    xt.copy_(torch.from_numpy(xm).to(gpu))
    base2.enable_fp64()
    got_m = base2.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    refm = _oracle_matrix(a_hat, xm, w, probes[:6], observe, 1e-4, torch.float64)
    assert np.abs(got_m[:6] - refm).max() <= 1e-5 * refm.max()


@pytest.mark.parametrize("f,h,c,hub", [(256, 256, 2, True), (64, 100, 3, True), (33, 24, 2, False), (130, 66, 7, True)])
def test_aggregate_first_route_of_the_fp64_preactivation(gpu, f, h, c, hub):
    """`delta` on dense features no wider than ~2 H (BASELINE configs[4]: F = H = 256): the pre-activation is formed as
    (A_hat X)[r] W1 + b1 on the rows the call's probes reach, on demand (lt_fp64.hip "aggregate-first"; lt_baseline_fp64_route
    == 2).  Against the fp64 oracle (1e-5 of the largest score) and the S1d route (fp64 summation order only: fp32
    rounding of the result); rows stay valid across calls and chunks, a refresh invalidates them, hub rows go through the
    segment sums, widths that need padding and unaligned feature rows (F = 33) included."""
    from test_gpu_parity import _hub_graph, _oracle_matrix
    from linkteller_amd import _lib, engine, graph, synth
    n = 1300
    adj = _hub_graph(n, 6000, 700, seed=f) if hub else synth.erdos_renyi_graph(n, 4000, seed=f)
    a_hat = graph.first_order_gcn(adj)
    x = synth.gaussian_features(n, f, seed=1)
    w = synth.gcn_weights(f, h, c, seed=2)
    hg = graph.HipGraph(a_hat)
    xt = torch.from_numpy(x).to(gpu)
    base = engine.Baseline(hg, xt, *_params(w, gpu)).enable_fp64()
    assert base.fp64_route() == 2
    rng = np.random.RandomState(3)
    probes = np.concatenate([[0], rng.choice(np.arange(1, n), 47, replace=False)])      # node 0 is the hub
    observe = np.concatenate([[0], rng.choice(np.arange(1, n), 150, replace=False)])
    ref64 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float64)
    got = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    assert np.abs(got - ref64).max() <= 1e-5 * ref64.max()
    assert np.all(got[ref64 == 0] == 0)
    # a second call with other probes reuses the rows that are valid and adds the missing ones; then everything again in
    # chunks of a few probes: same bits as the one-chunk call
    probes2 = np.concatenate([probes[10:30], rng.choice(n, 30, replace=False)])
    got2 = base.influence_rows(probes2, observe, 1e-4, "delta").cpu().numpy()
    _lib.set_tuning("chunk_budget_bytes", 40 * 1024)
    try:
        base.refresh()
        chunked = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
        chunked2 = base.influence_rows(probes2, observe, 1e-4, "delta").cpu().numpy()
    finally:
        _lib.set_tuning("chunk_budget_bytes", None)
    assert np.array_equal(chunked, got) and np.array_equal(chunked2, got2)
    ref2 = _oracle_matrix(a_hat, x, w, probes2[:8], observe, 1e-4, torch.float64)
    assert np.abs(got2[:8] - ref2).max() <= 1e-5 * ref2.max()
    # the S1d route on the same inputs
    _lib.set_tuning("aggregate_first", 0)
    try:
        base.refresh()
        assert base.fp64_route() == 0
        s1d = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    finally:
        _lib.set_tuning("aggregate_first", None)
    assert np.abs(s1d - ref64).max() <= 1e-5 * ref64.max()
    assert np.abs(s1d - got).max() <= 1e-6 * ref64.max()
    # in-place weight update + refresh: the stale rows are recomputed
    base.w1.mul_(1.02)
    base.refresh()
    w2 = dict(w, W1=base.w1.cpu().numpy())
    ref3 = _oracle_matrix(a_hat, x, w2, probes[:8], observe, 1e-4, torch.float64)
    got3 = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    assert np.abs(got3[:8] - ref3).max() <= 1e-5 * ref3.max()
    # `full` / `sparse` are untouched by the route
    assert np.array_equal(base.influence_rows(probes, observe, 1e-4, "full").cpu().numpy(),
                          base.influence_rows(probes, observe, 1e-4, "sparse").cpu().numpy())


def test_lapgraph_topk_on_the_device(gpu, capsys):
    """LapGraph's N x N add + top-k select on the GPU (lt_lapgraph_select; the Laplace draws stay numpy's
    for stream compatibility).  Bit-exact against the reference's own perturbed graphs (tests/golden/dp_adjacency.npz,
    eps 5 and 1) and against the host route; at the BASELINE configs[3] size (twitch-RU shape, eps = 5) the two routes
    give the same CSR and the timings are printed."""
    import time
    import scipy.sparse as sp
    from linkteller_amd import dp, synth
    g = load_golden("dp_adjacency.npz")
    a = csr_from(g, "adj")
    for eps in (5.0, 1.0):
        res = sp.csr_matrix(dp.perturb_adj(sp.csr_matrix(a), "continuous", eps, 42, backend="hip"))
        res.sort_indices()
        tag = f"continuous.eps{eps:g}"
        assert np.array_equal(res.indptr, g[f"{tag}.indptr"]), tag
        assert np.array_equal(res.indices, g[f"{tag}.indices"]), tag
        assert np.array_equal(np.asarray(res.data, dtype=np.float64), g[f"{tag}.data"]), tag
    adj, _, _ = synth.twitch_like_problem("twitch-RU", hidden=256, n_classes=2, seed=0)
    out = {}
    for backend in ("hip", "host", "hip"):
        t0 = time.time()
        r = sp.csr_matrix(dp.perturb_adj(adj, "continuous", 5.0, 42, backend=backend))
        out[backend] = (r, time.time() - t0)
        r.sort_indices()
    assert (out["hip"][0] != out["host"][0]).nnz == 0 and np.array_equal(out["hip"][0].indices, out["host"][0].indices)
    with capsys.disabled():
        print(f"\nLapGraph N={adj.shape[0]} eps=5: device select {out['hip'][1]:.2f} s, host argpartition {out['host'][1]:.2f} s "
              f"(both include the 0.2-0.3 s numpy Laplace draw of N^2 cells)")


@pytest.mark.parametrize("family", ["powerlaw", "er", "directed"])
def test_stage_b_per_observed_row_keeps_every_bit(gpu, family):
    """SPARSE / DELTA stage B of calls with a bitmap row per probe runs one block per (observed node, probe slice) with
    the observed row staged in LDS (k_item_stageB_rows) instead of one 8-lane group per pair.  Same chains, same
    butterfly: pinned off ("stageb_rows" = 0) it must give the same bits -- hub rows on both sides, a non-symmetric
    pattern, several probe chunks, probe counts that leave partial groups and partial slices."""
    import scipy.sparse as sp
    from linkteller_amd import _lib, engine, graph, synth
    n = 1500
    if family == "powerlaw":
        a_hat = graph.first_order_gcn(synth.powerlaw_graph(n, 9000, seed=3))
        assert np.diff(a_hat.indptr).max() > 300
    elif family == "er":
        a_hat = graph.first_order_gcn(synth.erdos_renyi_graph(n, 12000, seed=4))
    else:
        rng = np.random.RandomState(5)
        m = sp.csr_matrix((rng.uniform(0.1, 1.0, 14000).astype(np.float32), (rng.randint(0, n, 14000), rng.randint(0, n, 14000))), shape=(n, n))
        m.sum_duplicates()
        m.sort_indices()
        a_hat = m
    x = synth.gaussian_features(n, 48, seed=1)
    w = synth.gcn_weights(48, 64, 3, seed=2)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu))
    rng = np.random.RandomState(6)
    deg = np.diff(sp.csr_matrix(a_hat).indptr)
    big = np.argsort(-deg)[:5]
    try:
        for n_probe, n_obs in ((1, 40), (33, 200), (300, 700), (517, 90)):
            probes = np.concatenate([big[:min(3, n_probe)], rng.choice(n, max(0, n_probe - 3), replace=False)])[:n_probe]
            obs = np.concatenate([big[:2], rng.choice(n, n_obs - 2, replace=False)])
            for budget in (None, 1 << 16):
                _lib.set_tuning("chunk_budget_bytes", budget)
                got = {}
                for rows in (1, 0):
                    _lib.set_tuning("stageb_rows", rows)
                    got[rows] = {m: base.influence_rows(probes, obs, 1e-4, m).cpu().numpy() for m in ("sparse", "delta")}
                for m in ("sparse", "delta"):
                    assert np.array_equal(got[0][m], got[1][m]), (family, n_probe, n_obs, budget, m)
                assert np.array_equal(got[1]["sparse"], base.influence_rows(probes, obs, 1e-4, "full").cpu().numpy())
    finally:
        _lib.set_tuning("stageb_rows", None)
        _lib.set_tuning("chunk_budget_bytes", None)


@pytest.mark.parametrize("features", ["indicator", "gaussian"])
def test_preactivation_rows_on_demand_keep_every_bit(gpu, features):
    """On the S1d routes a small `delta` call (one rank of many) forms the fp64 pre-activation only on the rows its items read
    ("z_on_demand"; rows stay valid until the next refresh), a large one on all rows.  Same chains: pinned on (1) and off (0) the
    matrices are bit-identical -- hub rows included, across several calls and chunks, and after a refresh."""
    from test_gpu_parity import _hub_graph
    from linkteller_amd import _lib, engine, graph, synth
    n, f = 1400, 600
    a_hat = graph.first_order_gcn(_hub_graph(n, 6000, 500, seed=8))
    x = synth.twitch_like_features(n, f, seed=4, density=0.02) if features == "indicator" else synth.gaussian_features(n, f, seed=4)
    w = synth.gcn_weights(f, 64, 2, seed=5)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu)).enable_fp64()
    assert base.fp64_route() == (1 if features == "indicator" else 0)
    rng = np.random.RandomState(2)
    obs = np.concatenate([[0], rng.choice(np.arange(1, n), 300, replace=False)])
    calls = [np.concatenate([[0], rng.choice(n, 9, replace=False)]), rng.choice(n, 40, replace=False), rng.choice(n, 400, replace=False)]
    got = {}
    try:
        for knob in (0, 1):
            _lib.set_tuning("z_on_demand", knob)
            base.refresh()
            got[knob] = [base.influence_rows(p, obs, 1e-4, "delta").cpu().numpy() for p in calls]
            _lib.set_tuning("chunk_budget_bytes", 1 << 15)
            base.w1.mul_(1.0)      # (no change: a refresh alone must invalidate and rebuild)
            base.refresh()
            got[knob] += [base.influence_rows(p, obs, 1e-4, "delta").cpu().numpy() for p in calls]
            _lib.set_tuning("chunk_budget_bytes", None)
    finally:
        _lib.set_tuning("z_on_demand", None)
        _lib.set_tuning("chunk_budget_bytes", None)
    for a, b in zip(got[0], got[1]):
        assert np.array_equal(a, b)
    for k in range(3):
        assert np.array_equal(got[0][k], got[0][k + 3]) and got[0][k].max() > 0


@pytest.mark.parametrize("h1,h2,c,hub,features", [(32, 16, 2, False, "gaussian"), (100, 36, 3, True, "gaussian"),
                                                  (256, 64, 2, True, "indicator"), (20, 256, 8, False, "indicator")])
def test_gcn3_delta_mode_against_the_fp64_oracle(gpu, h1, h2, c, hub, features):
    """`--n-layer 3` (gcn/models.py:28-46) with `--influence-mode delta`: the perturbation propagated exactly through the
    three layers (lt_influence3_rows_mode, two ReLU kink tests on fp64 pre-activations).  Within 1e-5 of the largest score of
    the reference evaluated in fp64 (oracle, verbatim op sequence), exact zeros off the 3-hop set, chunking transparent, a
    refresh after an in-place weight change picked up; the fp32 finite difference (`sparse`) of the same baseline sits in
    its recorded noise class next to it."""
    import scipy.sparse as sp
    from test_gpu_parity import _hub_graph
    from linkteller_amd import _lib, engine, graph, synth
    from oracle import linkteller_oracle as O
    n, f = (700, 400) if hub else (220, 300)
    if hub:
        a = _hub_graph(n, 2500, 400, seed=h1)
    else:
        a = synth.powerlaw_graph(n, 600, seed=h1).tolil()
        for k in (5, 17, 99):
            a[k, :] = 0
            a[:, k] = 0
        a = sp.csr_matrix(a)
        a.eliminate_zeros()
    a_hat = graph.first_order_gcn(a)
    x = synth.gaussian_features(n, f, seed=3) if features == "gaussian" else synth.twitch_like_features(n, f, seed=3, density=0.03)
    rng = np.random.RandomState(h2)

    def u(shape, fan):
        s = 1.0 / np.sqrt(fan)
        return rng.uniform(-s, s, size=shape).astype(np.float32)

    P = dict(W1=u((f, h1), h1), b1=u((h1,), h1), W2=u((h1, h2), h2), b2=u((h2,), h2), W3=u((h2, c), c), b3=u((c,), c))
    dev_p = [torch.from_numpy(P[k]).to(gpu) for k in ("W1", "b1", "W2", "b2", "W3", "b3")]
    base = engine.Baseline3(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *dev_p)
    probes = np.concatenate([rng.choice(n, 19, replace=False), [0, 5], [3, 3]])
    observe = np.concatenate([rng.choice(n, 40, replace=False), [0, 99]])
    adj_t = O.to_torch_sparse(a_hat)

    def oracle(params):
        Pd = {k: torch.from_numpy(v).double() for k, v in params.items()}
        xt = torch.from_numpy(x).double()
        m = np.zeros((len(probes), len(observe)))
        with torch.no_grad():
            for i, v in enumerate(probes):
                gm = O.get_gradient_eps_mat(xt, adj_t.double(), Pd, int(v), 1e-4, forward=O.gcn3_forward)
                m[i] = gm[torch.as_tensor(observe)].norm(dim=1).numpy()
        return m

    ref64 = oracle(P)
    scale = ref64.max()
    got = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    err = np.abs(got - ref64).max() / scale
    print(f"gcn3 delta h1={h1} h2={h2} c={c} hub={hub} {features}: max {scale:.3g}, |delta - ref64| / max = {err:.2e}")
    assert err <= 1e-5
    assert np.all(got[ref64 == 0] == 0)
    assert np.array_equal(got[-1], got[-2])                         # duplicate probe -> identical rows
    sparse = base.influence_rows(probes, observe, 1e-4, "sparse").cpu().numpy().astype(np.float64)
    assert np.abs(sparse - got).max() <= 0.05 * scale + 0.05
    # the probes' fp64 product rows read off the baseline's product (round 6, default) against X[probes] W1 formed again on the f64
    # cores ("gcn3_product_gather" = 0): fp64 summation order / the fixed-point storage of the rows only
    _lib.set_tuning("gcn3_product_gather", 0)
    try:
        base.refresh()
        regemm = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    finally:
        _lib.set_tuning("gcn3_product_gather", None)
        base.refresh()
    assert np.abs(regemm - ref64).max() / scale <= 1e-5 and np.abs(regemm - got).max() <= 1e-6 * scale
    assert np.all(regemm[ref64 == 0] == 0)
    _lib.set_tuning("chunk_budget_bytes", 1 << 19)
    try:
        chunked = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    finally:
        _lib.set_tuning("chunk_budget_bytes", None)
    assert np.array_equal(chunked, got)
    base.w2.mul_(1.03)
    base.b1.add_(0.002)
    base.refresh()
    P2 = dict(P, W2=base.w2.cpu().numpy(), b1=base.b1.cpu().numpy())
    ref2 = oracle(P2)
    got2 = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
    assert np.abs(got2 - ref2).max() <= 1e-5 * ref2.max()


@pytest.mark.gpu
@pytest.mark.parametrize("long_par,p", [(1, 32), (1, 16), (0, 16)])
def test_full_mode_on_a_hub_of_many_segments(gpu, long_par, p):
    """FULL mode on a row of 9 499 entries (75 segments of 128: more than one batch of hit masks in k_full_long_combine)
    whose first segment holds EVERY probe of the first probe groups: no chain of those waves is the unperturbed one (the
    baseline slot of the segment comes from the recomputation), every wave of the segment recomputes 16-32 substituted
    probes (several passes of four), and later segments meet a few more.  The bits must be those of `sparse` (independent
    kernels), and a few rows must agree with the fp64 oracle."""
    import scipy.sparse as sp
    from linkteller_amd import _lib, engine, graph, synth
    from oracle import linkteller_oracle as O
    n = 9500
    rng = np.random.RandomState(4)
    r = rng.randint(1, n, 30000)
    c = rng.randint(1, n, 30000)
    keep = r != c
    rows = np.concatenate([np.zeros(n - 1, int), r[keep]])
    cols = np.concatenate([np.arange(1, n), c[keep]])
    a = sp.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.float32).tocsr()
    a_hat = graph.first_order_gcn(a)
    assert np.diff(a_hat.indptr).max() == n
    x = synth.twitch_like_features(n, 200, seed=6, density=0.03)
    w = synth.gcn_weights(200, 256, 2, seed=8)
    probes = np.concatenate([np.arange(1, 71), rng.choice(np.arange(200, n), 30, replace=False)]).astype(np.int32)
    obs = np.concatenate([[0], np.arange(1, 40), rng.choice(np.arange(200, n), 60, replace=False)]).astype(np.int32)
    _lib.set_tuning("full_p", p)
    _lib.set_tuning("long_par", long_par)
    try:
        base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu))
        f = base.influence_rows(probes, obs, 1e-4, "full").cpu().numpy()
        s_ = base.influence_rows(probes, obs, 1e-4, "sparse").cpu().numpy()
    finally:
        _lib.set_tuning("full_p", None)
        _lib.set_tuning("long_par", None)
    assert np.array_equal(f, s_)
    assert np.isfinite(f).all() and f.max() > 0 and (f[:, 0] > 0).sum() > 50      # the hub sees every probe (some of the fp32 differences round to 0)
    # the REFERENCE's own fp32 scores and the fp64 evaluation on all 100 probe rows: tests/golden/hub_noise.npz (generated from the
    # imported reference by tests/golden/generate_hub_noise.py -- its get_gradient_eps_mat on torch's CPU kernels, one thread)
    from conftest import load_golden
    hg = load_golden("hub_noise.npz")
    assert np.array_equal(hg["probes"], probes) and np.array_equal(hg["obs"], obs)
    ref64, ref32 = hg["ref64"], hg["ref32"].astype(np.float64)
    ours = np.abs(f.astype(np.float64) - ref64).max(axis=1)
    theirs = np.abs(ref32 - ref64).max(axis=1)
    # A 9 500-term fp32 sum in two different orders (8 strided chains here, sequential in torch.spmm): the pair (probe, hub) carries
    # the hub's ulp-quantised logit difference / 1e-4 -- a few 1e-3 per ulp -- in both, and the reference's own figure moves with
    # the host it runs on (row maxima rms 0.0210 on the build container's Xeon, 0.0145 on the GPU box's EPYC for the oracle's port).
    # Measured (round 5, tools/hub_noise.py): ours 0.0273, worst row 0.067 against 0.055.  Gates (row by row again, so
    # that no row hides behind its neighbours): every row within 4x its own reference error or 5x the reference's rms, the rms of the
    # row maxima within 2x the reference's, the worst row within 2x the reference's worst.
    rms_o, rms_t = float(np.sqrt((ours ** 2).mean())), float(np.sqrt((theirs ** 2).mean()))
    print(f"hub rows: |full - ref64| row maxima rms {rms_o:.5f} max {ours.max():.5f}; reference fp32 rms {rms_t:.5f} max {theirs.max():.5f}")
    for i, (o, t) in enumerate(zip(ours, theirs)):
        assert o <= max(4.0 * t, 5.0 * rms_t), (i, o, t, rms_t)
    assert rms_o <= 2.0 * rms_t, (rms_o, rms_t)
    assert ours.max() <= 2.0 * theirs.max(), (ours.max(), theirs.max())
    #  why 2x and not BASELINE.md's 1x here: the golden ref32 is ONE draw of the reference's own noise (this case's
    # figure moves by 1.45x between two hosts for the reference itself), so 2x is the noise-class ceiling of conftest.noise_gate;
    # the MEASURED ratios are recorded per (p, long_par) and gated at +10 % like every other fp32 case -- a drift from 1.3x
    # towards 2x fails here first.  (Every summation order is fixed: the ratios reproduce to the last digit on every box.)
    from conftest import noise_gate
    noise_gate(f"hub_row.p{p}.lp{long_par}.rms", rms_o / rms_t)
    noise_gate(f"hub_row.p{p}.lp{long_par}.max", ours.max() / theirs.max())


@pytest.mark.gpu
def test_delta_with_a_whole_row_of_hidden_units_at_their_kinks(gpu):
    """The worst case for the storage precision of the fp64 product rows: the bias is chosen so that EVERY hidden unit
    of one row has a pre-activation of (almost) zero, i.e. every perturbation through that row crosses a ReLU kink and contributes
    `-z` / `z + dz` -- the place where a rounding of the TERMS z is summed from reaches the result as an absolute error.  Plain fp32
    rows (the default for most of round 3) gave ~7e-4 of the largest score here; the 32-bit fixed-point rows must stay inside the
    1e-5 every other `delta` test asserts, fp64 rows (`s1_f32 = 0`) inside 1e-6."""
    from linkteller_amd import _lib, engine, graph, synth
    from oracle import linkteller_oracle as O
    n, h, c, f = 600, 64, 2, 8
    a_hat = graph.first_order_gcn(synth.powerlaw_graph(n, 3000, seed=3))
    x = synth.gaussian_features(n, f, seed=5)            # F = 8: the feature-difference route serves it (route 1)
    w = dict(synth.gcn_weights(f, h, c, seed=7))
    deg = np.diff(a_hat.indptr)
    r0 = int(np.argsort(deg)[len(deg) // 2])
    z0 = (a_hat.astype(np.float64) @ (x.astype(np.float64) @ w["W1"].astype(np.float64)))[r0]
    w["b1"] = (-z0).astype(np.float32)
    nb = a_hat[r0].indices
    probes = np.unique(np.concatenate([nb, [r0]])).astype(np.int32)
    obs = np.unique(np.concatenate([nb, a_hat[nb[0]].indices])).astype(np.int32)
    P64 = {k: torch.from_numpy(w[k]).double() for k in ("W1", "b1", "W2", "b2")}
    adj_o, x64 = O.to_torch_sparse(a_hat).double(), torch.from_numpy(x).double()
    ref = np.zeros((len(probes), len(obs)))
    with torch.no_grad():
        for i, v in enumerate(probes):
            ref[i] = O.get_gradient_eps_mat(x64, adj_o, P64, int(v), 1e-4)[torch.as_tensor(obs.astype(np.int64))].norm(dim=1).numpy()
    scale = ref.max()
    err = {}
    try:
        for knob in (1, 0):
            _lib.set_tuning("s1_f32", knob)
            base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu))
            got = base.influence_rows(probes, obs, 1e-4, "delta").cpu().numpy().astype(np.float64)
            assert base.fp64_route() == 1
            err[knob] = np.abs(got - ref).max() / scale
    finally:
        _lib.set_tuning("s1_f32", None)
    assert err[1] <= 1e-5, err
    assert err[0] <= 1e-6, err
