"""GPU parity: the HIP path (through the C ABI) against the golden vectors of the reference.

Tolerances (see DESIGN.md "Parity"):
  * logits: fp32 kernels vs the reference's fp64 logits, |err| <= 2e-5 * max|logit| + 1e-6
    (the reference's own fp32 logits sit at the same distance from its fp64 ones).
  * influence, mode 'delta': relative to the matrix maximum, <= 1e-5 of the reference evaluated in
    fp64 (north_star tolerance is 1e-4).  The reference's own fp32 run is 2e-3..6e-3 away from that
    (fp32 cancellation amplified by 1/delta = 1e4), so it cannot itself be the
    1e-4 target; it is checked to be *further* from fp64 than we are.
  * influence, modes 'full'/'sparse' (the fp32 finite difference, same noise class as the
    reference): error vs fp64 <= 2x the reference-fp32's own error (measured 0.5 .. 1.5x; tiny cases where that
    error is itself a few ulps keep an absolute floor).  'sparse' == 'full' bit for bit.
  * AUC / AP from our scores (reference pair lookup + sklearn): delta within 1e-4 of the reference evaluated in
    fp64 (measured: equal to 6 digits).  full / sparse: within 1e-4 (+ the reference's own fp32 gap) on the edges
    the fp32 finite difference resolves; a low-score edge quantised to 0 moves the raw AUC by 1 / n_edges, in the
    reference's own fp32 run too, so the raw figure is printed, not pinned.
  * exact zeros: every pair the reference scores exactly 0 in fp64 is exactly 0 here (all modes).
"""
import os

import numpy as np
import pytest
import torch

from conftest import csr_from, golden_args, noise_gate

pytestmark = pytest.mark.gpu


def _setup(g, key, dev):
    from linkteller_amd import engine, graph
    args = golden_args(g, key)
    served = csr_from(g, f"{key}.served") if f"{key}.served.n" in g else csr_from(g, f"{key}.adj")
    a_hat = graph.fetch_normalization(args["norm"])(served)
    x = torch.from_numpy(g[f"{key}.x"]).to(dev)
    p = [torch.from_numpy(g[f"{key}.sd.{k}"]).to(dev) for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias")]
    base = engine.Baseline(graph.HipGraph(a_hat), x, *p)
    return args, base


@pytest.mark.parametrize("key", ["n64", "n600", "n200c7"])
def test_forward_logits(forward_golden, gpu, key):
    from linkteller_amd import engine, graph
    g = forward_golden
    a_hat = graph.first_order_gcn(csr_from(g, f"{key}.adj"))
    x = torch.from_numpy(g[f"{key}.x"]).to(gpu)
    p = [torch.from_numpy(g[f"{key}.sd.{k}"]).to(gpu) for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias")]
    hg = graph.HipGraph(a_hat)
    out = engine.gcn2_forward(hg, x, *p).cpu().numpy().astype(np.float64)
    ref64 = g[f"{key}.logits64"]
    tol = 2e-5 * np.abs(ref64).max() + 1e-6
    assert np.abs(out - ref64).max() <= tol
    # the baseline state computes the same logits bit for bit
    base = engine.Baseline(hg, x, *p)
    assert np.array_equal(base.logits().cpu().numpy().astype(np.float64), out)


@pytest.mark.parametrize("key", ["er300", "pl600", "pl600hi", "lap600", "rand400"])
def test_influence_matrix(influence_golden, gpu, key):
    g = influence_golden
    args, base = _setup(g, key, gpu)
    nodes = g[f"{key}.ref32.test_nodes"]
    ref64 = g[f"{key}.ref64.influence_val"]
    ref32 = g[f"{key}.ref32.influence_val"]
    scale = ref64.max()
    err32 = np.abs(ref32 - ref64).max()
    res = {m: base.influence_rows(nodes, nodes, args["influence"], m).cpu().numpy().astype(np.float64)
           for m in ("full", "sparse", "delta")}
    assert np.array_equal(res["full"], res["sparse"]), "sparse mode must be bit-identical to full"
    e_delta = np.abs(res["delta"] - ref64).max()
    e_full = np.abs(res["full"] - ref64).max()
    print(f"{key}: max score {scale:.3f}; |ref32-ref64|={err32:.2e}; |delta-ref64|={e_delta:.2e}; |full-ref64|={e_full:.2e}")
    assert e_delta <= 1e-5 * scale   # north_star asks 1e-4; measured 3e-7 (fp64 kink test, lt_fp64.hip)
    assert e_delta < err32
    # the fp32 finite difference: error in units of the reference's own fp32 error, against the recorded value (+10 %)
    noise_gate(f"influence.{key}.full_vs_ref32_error", e_full / err32)
    zero64 = ref64 == 0
    for m, r in res.items():
        assert np.all(r[zero64] == 0), f"{m}: non-zero where the reference is exactly zero"
    # AUC / AP (north_star: "influence scores and AUC within 1e-4"): the attack's own metrics from OUR scores,
    # through the reference's pair lookup and sklearn calls (attacker.py:233-247, 378-386).  delta sits within 1e-4 of
    # the reference evaluated in fp64; full sits inside the band the reference's own fp32 <-> fp64 runs span.
    from oracle import linkteller_oracle as O
    ex, nex = g[f"{key}.ref32.exist"].tolist(), g[f"{key}.ref32.nonexist"].tolist()
    auc64, ap64 = float(g[f"{key}.ref64.auc"]), float(g[f"{key}.ref64.ap"])
    auc32, ap32 = float(g[f"{key}.ref32.auc"]), float(g[f"{key}.ref32.ap"])
    got = {m: O.attack_metrics(*O.pair_scores(res[m], nodes, ex, nex)) for m in ("delta", "full")}
    print(f"{key}: auc ref64 {auc64:.6f} ref32 {auc32:.6f} delta {got['delta']['auc']:.6f} full {got['full']['auc']:.6f}; "
          f"ap ref64 {ap64:.6f} ref32 {ap32:.6f} delta {got['delta']['ap']:.6f} full {got['full']['ap']:.6f}")
    assert abs(got["delta"]["auc"] - auc64) <= 1e-4 and abs(got["delta"]["ap"] - ap64) <= 1e-4
    # RAW AUC / AP of the fp32 modes against the band the reference's own fp32 and fp64 runs span (+- 1e-4): the distance
    # outside the band is recorded per fixture (0 = inside).  It is NOT zero everywhere (pl600: one edge of true score
    # 0.0067 quantises to 0 here and not in the reference's fp32 run -> AUC -0.0099; the reference loses pl600hi's the same
    # way), which is why `delta` -- asserted to 1e-4 above -- is the product default and the mode bench.py's `value` is
    # measured in.
    def outside(v, a, b):
        lo, hi = min(a, b) - 1e-4, max(a, b) + 1e-4
        return max(0.0, lo - v, v - hi)
    # bound: ONE low-score edge quantising to 0 (here or in the reference's fp32 run) moves AUC by at most 1 / n_edges
    one_edge = 1.0 / max(len(ex), 1) + 1e-4
    noise_gate(f"influence.{key}.full_raw_auc_outside_band", outside(got["full"]["auc"], auc32, auc64), ceiling=one_edge)
    noise_gate(f"influence.{key}.full_raw_ap_outside_band", outside(got["full"]["ap"], ap32, ap64), ceiling=one_edge)
    # full / sparse are the fp32 finite difference (f(X + d) - f(X)) / 1e-4 itself: scores are quantised to
    # ulp(logit) / 1e-4 ~ 1e-2, so a pair whose true score is below that can come out exactly 0 -- in the reference's
    # fp32 run as well as here (counted below).  ONE low-score edge falling to zero moves AUC by up to 1 / n_edges
    # (pl600: the 0.0067 edge, 1 of 55 -> AUC -0.0099; the reference lost 0.0042 the same way on pl600hi), so the raw
    # AUC of these modes is only printed.  Asserted: on the edges the fp32 difference can resolve (fp64 score >= 2x
    # the reference's own fp32 error; all non-edges kept) AUC is within 1e-4 of the fp64 reference, on top of the
    # reference's own fp32 gap on that same set; and we lose no more nonzero scores to quantisation than it does.
    from sklearn.metrics import roc_auc_score
    se64 = np.asarray(O.pair_scores(ref64, nodes, ex, nex)[0])
    keep = se64 >= 2.0 * err32

    def resolvable_auc(mat):
        se, sn = O.pair_scores(mat, nodes, ex, nex)
        se = np.asarray(se)[keep]
        return roc_auc_score([1] * len(se) + [0] * len(sn), list(se) + list(sn))

    a64, a32, ours = resolvable_auc(ref64), resolvable_auc(ref32), resolvable_auc(res["full"])
    print(f"{key}: AUC on the {int(keep.sum())}/{len(keep)} resolvable edges: ref64 {a64:.6f} ref32 {a32:.6f} full {ours:.6f}")
    assert abs(ours - a64) <= 1e-4 + abs(a32 - a64)
    lost_ours = int(((ref64 > 0) & (res["full"] == 0)).sum())
    lost_ref = int(((ref64 > 0) & (ref32 == 0)).sum())
    print(f"{key}: nonzero fp64 scores that quantise to exactly 0: ours {lost_ours}, reference fp32 {lost_ref}")
    assert lost_ours <= 1.25 * lost_ref + 8


def _hub_graph(n, e, hub_deg, seed):
    """power-law graph plus one node wired to `hub_deg` others (exercises the long-row segments)."""
    import scipy.sparse as sp
    from linkteller_amd import synth
    a = synth.powerlaw_graph(n, e, seed=seed).tolil()
    rng = np.random.RandomState(seed)
    nb = rng.choice(np.arange(1, n), hub_deg, replace=False)
    a[0, nb] = 1.0
    a[nb, 0] = 1.0
    return sp.csr_matrix(a)


@pytest.mark.parametrize("ncols,bias,relu", [(256, True, True), (256, False, False), (64, True, False),
                                              (20, False, True), (2, True, False), (7, False, False), (8, True, True)])
def test_spmm_matches_scipy(gpu, ncols, bias, relu):
    from linkteller_amd import engine, graph
    a_hat = graph.first_order_gcn(_hub_graph(2500, 12000, 1400, seed=5))
    assert np.diff(a_hat.indptr).max() > 1024       # at least three 512-entry segments
    rng = np.random.RandomState(1)
    s = rng.standard_normal((a_hat.shape[0], ncols)).astype(np.float32)
    b = rng.standard_normal(ncols).astype(np.float32) if bias else None
    got = engine.spmm(graph.HipGraph(a_hat), torch.from_numpy(s).to(gpu),
                      None if b is None else torch.from_numpy(b).to(gpu), relu=relu).cpu().numpy()
    want = a_hat.astype(np.float64) @ s.astype(np.float64)
    if b is not None:
        want = want + b
    if relu:
        want = np.maximum(want, 0)
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_gemm_matches_numpy(gpu):
    from linkteller_amd import engine
    rng = np.random.RandomState(2)
    for m, k, n in ((1, 1, 1), (65, 33, 17), (500, 3170, 256), (130, 1500, 70)):
        a = rng.standard_normal((m, k)).astype(np.float32)
        b = rng.standard_normal((k, n)).astype(np.float32)
        got = engine.gemm(torch.from_numpy(a).to(gpu), torch.from_numpy(b).to(gpu)).cpu().numpy()
        want = a.astype(np.float64) @ b.astype(np.float64)
        assert np.abs(got - want).max() <= 2e-6 * np.sqrt(k) * max(1.0, np.abs(want).max())


def test_hub_rows_in_the_probe_kernels(gpu):
    """A hub node as probe AND as observed node: long rows through stage A / stage B / item kernels."""
    from linkteller_amd import engine, graph, synth
    adj = _hub_graph(1500, 6000, 700, seed=9)
    x = synth.twitch_like_features(1500, 64, seed=3, density=0.05)
    w = synth.gcn_weights(64, 256, 2, seed=42)
    base = engine.Baseline(graph.HipGraph(graph.first_order_gcn(adj)), torch.from_numpy(x).to(gpu),
                           *[torch.from_numpy(w[k]).to(gpu) for k in ("W1", "b1", "W2", "b2")])
    nodes = np.concatenate([[0], np.random.RandomState(0).choice(np.arange(1, 1500), 40, replace=False)])
    full = base.influence_rows(nodes, nodes, 1e-4, "full").cpu().numpy()
    sparse = base.influence_rows(nodes, nodes, 1e-4, "sparse").cpu().numpy()
    delta = base.influence_rows(nodes, nodes, 1e-4, "delta").cpu().numpy()
    assert np.array_equal(full, sparse)
    assert np.abs(full - delta).max() <= 0.02 * delta.max() + 0.05   # fp32 finite-difference noise only
    assert np.array_equal(full == 0, delta == 0) or np.all(full[delta == 0] == 0)


def test_cli_end_to_end_matches_oracle(gpu, tmp_path, monkeypatch, capsys):
    """README-style command line on a synthetic MUSAE-format dataset: Worker -> GCNTrainer -> Attacker
    -> HIP probe kernels -> result file, against the oracle run on the same files."""
    import argparse
    from test_cli_worker_dp import _write_musae
    from linkteller_amd import main as lt_main, synth
    from linkteller_amd.gcn import GCN
    from oracle import linkteller_oracle as O
    a1, a2 = synth.powerlaw_graph(260, 1200, seed=1), synth.powerlaw_graph(320, 1500, seed=2)
    _write_musae(str(tmp_path), "ES", a1, 400, 1)
    _write_musae(str(tmp_path), "RU", a2, 400, 2)
    torch.manual_seed(0)
    model = GCN(3170, 256, 2, 0.5)
    torch.save(model.state_dict(), tmp_path / "model.pt")
    monkeypatch.chdir(tmp_path)
    for mode in ("delta", "full"):
        lt_main.main(f"--mode vanilla-clean --dataset twitch/ES/RU --hidden 256 --norm FirstOrderGCN --test "
                     f"--model-path {tmp_path}/model.pt --attack --attack-mode efficient --sample-type unbalanced "
                     f"--n-test 60 --influence-mode {mode} --data-root {tmp_path}".split())
        out = capsys.readouterr().out
        assert "attack results saved to: eval_twitch/ES/RU/efficient_unbalanced_60_42.pt" in out
        saved = torch.load("eval_twitch/ES/RU/efficient_unbalanced_60_42.pt", weights_only=False)
        # oracle on the same inputs (fp64 evaluation of the reference algorithm)
        from linkteller_amd.worker import Worker
        args = argparse.Namespace(norm="FirstOrderGCN")
        monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
        w = Worker(args, "twitch/ES/RU", "vanilla-clean", data_root=str(tmp_path))
        monkeypatch.undo(); monkeypatch.chdir(tmp_path)
        np.random.seed(42)
        (ex, nex), nodes = O.sample_subgraph_pairs("twitch/ES/RU", "unbalanced", w.adj_ori, 60)
        P = {k: v.double() for k, v in zip(("W1", "b1", "W2", "b2"), [model.state_dict()[n] for n in
                                                                  ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias")])}
        infl = O.influence_matrix(w.features_2.double(), w.adj_2.double(), P, nodes, 1e-4)
        ne, nn = O.pair_scores(infl, nodes, ex, nex)
        m = O.attack_metrics(ne, nn)
        assert saved["result"]["y"] == m["y"]
        pred = np.asarray(saved["result"]["pred"])
        tol = 1e-5 if mode == "delta" else 2e-2
        assert np.abs(pred - np.asarray(m["pred"])).max() <= tol * max(1.0, np.max(m["pred"]))
        if mode == "delta":
            import sklearn.metrics as skm
            assert abs(skm.auc(saved["auc"]["fpr"], saved["auc"]["tpr"]) - m["auc"]) <= 1e-4


def _next_rows_setup(gpu, prefix_adj, xkey):
    import argparse
    import types
    from linkteller_amd import graph
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "next_rows.npz"), allow_pickle=False)
    a = csr_from(g, prefix_adj)
    x = torch.from_numpy(g[xkey]).to(gpu)
    adj_t = graph.sparse_mx_to_torch_sparse_tensor(graph.first_order_gcn(a)).to(gpu)
    w = types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=a, n_nodes=a.shape[0])
    return g, a, w, argparse


def test_balanced_full_and_baselines_on_device(gpu, tmp_path, monkeypatch):
    from linkteller_amd.attacker import Attacker
    from linkteller_amd.gcn import GCN
    g, ab, w, argparse = _next_rows_setup(gpu, "bf.adj", "bf.x")
    model = GCN(64, 32, 2, 0.5)
    model.load_state_dict({k: torch.from_numpy(g[f"sd.{k}"]) for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias")})
    model.to(gpu).eval()
    monkeypatch.chdir(tmp_path)
    for mode, tol in (("delta", 1e-5), ("full", 3e-2)):
        args = argparse.Namespace(dataset="twitch/ES/RU", sample_type="balanced-full", n_test=7, sample_seed=82,
                                  influence=1e-4, mode="vanilla-clean", attack_mode="efficient", influence_mode=mode)
        atk = Attacker(args, model, w)
        atk.prepare_test_data()
        assert np.array_equal(atk.exist_edges, g["bf.exist"])
        atk.link_prediction_attack_efficient_balanced(chunk=50)      # several probe chunks
        saved = torch.load(str(g["bf.ref32.filename"]), weights_only=False)
        ref = np.concatenate([g["bf.ref64.norm_exist"], g["bf.ref64.norm_nonexist"]])
        got = np.asarray(saved["result"]["pred"])
        assert got.shape == ref.shape and np.abs(got - ref).max() <= tol * ref.max()
    args = argparse.Namespace(dataset="twitch/ES/RU", sample_type="balanced-full", n_test=7, sample_seed=82,
                              influence=1e-4, mode="vanilla-clean", attack_mode="baseline")
    atk = Attacker(args, model, w)
    atk.prepare_test_data()
    atk.baseline_attack_balanced()
    got = np.asarray(torch.load(str(g["bf.baseline.filename"]), weights_only=False)["result"]["pred"])
    ref = np.concatenate([g["bf.baseline.norm_exist"], g["bf.baseline.norm_nonexist"]])
    assert np.abs(got - ref).max() <= 2e-5


def test_gcn3_efficient_attack_on_device(gpu, tmp_path, monkeypatch):
    from linkteller_amd.attacker import Attacker
    from linkteller_amd.gcn import GCN3
    g, a, w, argparse = _next_rows_setup(gpu, "adj", "x")
    model = GCN3(64, 32, 16, 2, 0.5)
    model.load_state_dict({k: torch.from_numpy(g[f"gcn3.sd.{k}"]) for k in
                           ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias", "gc3.weight", "gc3.bias")})
    model.to(gpu).eval()
    args = argparse.Namespace(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=32, sample_seed=42,
                              influence=1e-4, mode="vanilla-clean", attack_mode="efficient")
    atk = Attacker(args, model, w)
    atk.prepare_test_data()
    assert np.array_equal(atk.test_nodes, g["gcn3.ref32.test_nodes"])
    assert atk._is_three_layer()
    infl = atk.influence_matrix("sparse")                      # lt_influence3_rows: the 3-hop probe primitive (fp32 finite difference)
    exact = atk.influence_matrix("delta")                      # the perturbation propagated exactly through the three layers
    ref64_ = g["gcn3.ref64.influence_val"]
    print(f"gcn3 delta: |ours-ref64| / max = {np.abs(exact - ref64_).max() / ref64_.max():.2e}")
    assert np.abs(exact - ref64_).max() <= 1e-5 * ref64_.max() and np.all(exact[ref64_ == 0] == 0)
    assert np.array_equal(atk.influence_matrix(), exact)       # `delta` is the default mode, for three layers too
    ref64, ref32 = g["gcn3.ref64.influence_val"], g["gcn3.ref32.influence_val"]
    e32 = np.abs(ref32 - ref64).max()
    print(f"gcn3: |ref32-ref64|={e32:.3e} |ours-ref64|={np.abs(infl - ref64).max():.3e}")
    noise_gate("gcn3.fixture.primitive", np.abs(infl - ref64).max() / e32)   # fp32 finite difference: the reference's noise class
    assert np.all(infl[ref64 == 0] == 0)
    # the per-probe loop over the unfused layers (what round 1 shipped) is the same noise class, and slower
    import time
    nodes = np.asarray(atk.test_nodes, dtype=np.int64)
    loop = atk._rows_generic(nodes, nodes).cpu().numpy().astype(np.float64)
    noise_gate("gcn3.fixture.generic_loop", np.abs(loop - ref64).max() / e32)
    assert np.all(loop[ref64 == 0] == 0)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        atk._rows(nodes, nodes)
    torch.cuda.synchronize()
    t_prim = (time.time() - t0) / 5
    print(f"gcn3 n_test=32: probe primitive {t_prim * 1e3:.3f} ms per matrix (baseline refresh included)")
    if os.environ.get("LT_ASSERT_TIMINGS"):       # wall-clock bounds are opt-in: a shared or profiled GPU is not a bug
        assert t_prim < 5e-3
    # the baseline logits of the primitive equal the model's forward within fp32 rounding
    base_logits = atk.baseline3().logits().cpu().numpy().astype(np.float64)
    # logits of the 3-layer model through the unfused HIP layers
    with torch.no_grad():
        out = model(w.features_2, w.adj_2).cpu().numpy().astype(np.float64)
    from oracle import linkteller_oracle as O
    P3 = {k: torch.from_numpy(g[f"gcn3.sd.{n}"]).double() for k, n in (("W1", "gc1.weight"), ("b1", "gc1.bias"),
          ("W2", "gc2.weight"), ("b2", "gc2.bias"), ("W3", "gc3.weight"), ("b3", "gc3.bias"))}
    ref = O.gcn3_forward(torch.from_numpy(g["x"]).double(), O.to_torch_sparse(O.first_order_gcn(a)).double(), P3).numpy()
    assert np.abs(out - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6
    assert np.abs(base_logits - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6


def _oracle_matrix(adj_hat_csr, x, w, probes, observe, delta, dtype):
    """Reference algorithm (oracle) for arbitrary probe / observe node lists."""
    from oracle import linkteller_oracle as O
    adj_t = O.to_torch_sparse(adj_hat_csr).to(dtype)
    P = {k: torch.from_numpy(w[k]).to(dtype) for k in ("W1", "b1", "W2", "b2")}
    xt = torch.from_numpy(x).to(dtype)
    out = np.zeros((len(probes), len(observe)))
    with torch.no_grad():
        for i, v in enumerate(probes):
            g = O.get_gradient_eps_mat(xt, adj_t, P, int(v), delta)
            out[i] = g[torch.as_tensor(np.asarray(observe))].norm(dim=1).numpy()
    return out


@pytest.mark.parametrize("h,c,norm", [(10, 2, "FirstOrderGCN"), (100, 3, "FirstOrderGCN"), (16, 1, "AugNormAdj"),
                                      (256, 8, "FirstOrderGCN"), (64, 7, "NormAdj"), (132, 2, "FirstOrderGCN")])
def test_shapes_and_edge_cases(gpu, h, c, norm):
    """Hidden widths that need padding / several lane groupings, 1..8 classes, graphs with isolated nodes
    and (NormAdj) empty rows, duplicate probes, observe list != probe list, a single probe."""
    import scipy.sparse as sp
    from linkteller_amd import engine, graph, synth
    n, f = 180, 40
    a = synth.powerlaw_graph(n, 500, seed=h + c).tolil()
    for k in (5, 17, 99):                # isolated nodes
        a[k, :] = 0
        a[:, k] = 0
    a = sp.csr_matrix(a)
    a.eliminate_zeros()
    a_hat = graph.fetch_normalization(norm)(a)
    x = synth.gaussian_features(n, f, seed=3)
    w = synth.gcn_weights(f, h, c, seed=h)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu),
                           *[torch.from_numpy(w[k]).to(gpu) for k in ("W1", "b1", "W2", "b2")])
    rng = np.random.RandomState(1)
    probes = np.concatenate([rng.choice(n, 21, replace=False), [5, 17], [3, 3]])     # isolated + duplicate probes
    observe = np.concatenate([rng.choice(n, 30, replace=False), [99, 5]])
    ref64 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float64)
    ref32 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float32)
    scale = max(ref64.max(), 1e-6)
    e32 = np.abs(ref32 - ref64).max()
    res = {m: base.influence_rows(probes, observe, 1e-4, m).cpu().numpy().astype(np.float64) for m in ("full", "sparse", "delta")}
    assert np.array_equal(res["full"], res["sparse"])
    assert np.abs(res["delta"] - ref64).max() <= 1e-5 * scale
    # (tiny cases whose fp32 error is itself a few ulps keep an absolute floor of 1e-4 of the largest score)
    noise_gate(f"shapes.h{h}c{c}{norm}.full", np.abs(res["full"] - ref64).max() / max(e32, 1e-4 * scale))
    for r in res.values():
        assert np.all(r[ref64 == 0] == 0)
    assert np.array_equal(res["full"][-1], res["full"][-2])          # duplicate probe -> identical rows
    one = base.influence_rows(probes[:1], observe, 1e-4, "full").cpu().numpy()
    assert np.array_equal(one[0], res["full"][0].astype(np.float32))
    logits = base.logits().cpu().numpy().astype(np.float64)
    from oracle import linkteller_oracle as O
    ref_logits = O.gcn_forward(torch.from_numpy(x).double(), O.to_torch_sparse(a_hat).double(),
                               {k: torch.from_numpy(w[k]).double() for k in w}).numpy()
    assert np.abs(logits - ref_logits).max() <= 2e-5 * max(1.0, np.abs(ref_logits).max())


def test_empty_and_degenerate_calls(gpu):
    import scipy.sparse as sp
    from linkteller_amd import _lib, engine, graph, synth
    a_hat = graph.first_order_gcn(synth.erdos_renyi_graph(50, 100, seed=1))
    x = synth.gaussian_features(50, 12, seed=2)
    w = synth.gcn_weights(12, 32, 2, seed=1)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu),
                           *[torch.from_numpy(w[k]).to(gpu) for k in ("W1", "b1", "W2", "b2")])
    assert base.influence_rows([], [1, 2], 1e-4, "full").shape == (0, 2)
    assert base.influence_rows([1, 2], [], 1e-4, "delta").shape == (2, 0)
    with pytest.raises(IndexError):
        base.influence_rows([50], [0], 1e-4, "full")
    with pytest.raises(IndexError):
        base.influence_rows([0], [-1], 1e-4, "full")
    with pytest.raises(_lib.LinkTellerHipError):        # unsupported shapes fail loudly, no fallback
        engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), torch.zeros(12, 300, device=gpu),
                        torch.zeros(300, device=gpu), torch.zeros(300, 2, device=gpu), torch.zeros(2, device=gpu))
    with pytest.raises(_lib.LinkTellerHipError):
        engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), torch.zeros(12, 16, device=gpu),
                        torch.zeros(16, device=gpu), torch.zeros(16, 9, device=gpu), torch.zeros(9, device=gpu))
    # non-finite inputs are refused, not masked (ReLU is a v_max: NaN -> 0 would hide them; torch propagates NaN)
    x_bad = torch.from_numpy(x).to(gpu).clone()
    x_bad[3, 2] = float("nan")
    with pytest.raises(ValueError, match="non-finite"):
        engine.Baseline(graph.HipGraph(a_hat), x_bad, *[torch.from_numpy(w[k]).to(gpu) for k in ("W1", "b1", "W2", "b2")])
    # graph with no edges at all: A_hat = I, influence is purely the self term
    eye = graph.first_order_gcn(sp.csr_matrix((50, 50), dtype=np.float32))
    b2 = engine.Baseline(graph.HipGraph(eye), torch.from_numpy(x).to(gpu),
                         *[torch.from_numpy(w[k]).to(gpu) for k in ("W1", "b1", "W2", "b2")])
    r = b2.influence_rows([0, 1, 2], [0, 1, 2], 1e-4, "delta").cpu().numpy()
    assert np.all(r[~np.eye(3, dtype=bool)] == 0) and np.all(np.diag(r) > 0)


def test_probe_chunking_is_transparent(gpu, influence_golden, monkeypatch):
    """A workspace budget small enough to split the probe list into many chunks gives the same bits."""
    g = influence_golden
    args, base = _setup(g, "pl600", gpu)
    nodes = g["pl600.ref32.test_nodes"]
    ref = {m: base.influence_rows(nodes, nodes, args["influence"], m).cpu().numpy() for m in ("full", "sparse", "delta")}
    from linkteller_amd import _lib
    _lib.set_tuning("chunk_budget_bytes", 120 * 1024)     # ~10 probes per chunk at n=600, H=256
    try:
        args2, base2 = _setup(g, "pl600", gpu)
        for m in ("full", "sparse", "delta"):
            got = base2.influence_rows(nodes, nodes, args["influence"], m).cpu().numpy()
            assert np.array_equal(got, ref[m]), m
    finally:
        _lib.set_tuning("chunk_budget_bytes", None)


def test_rmat_shape_scaled_config5(gpu):
    """BASELINE configs[4] shape (R-MAT, F = H = 256, C = 2) at a scale the oracle finishes in seconds:
    heavy-tailed degrees (hub rows of thousands of entries) through every mode."""
    from linkteller_amd import engine, graph, synth
    adj = synth.rmat_graph(13, (1 << 13) * 12, seed=42)
    a_hat = graph.first_order_gcn(adj)
    assert np.diff(a_hat.indptr).max() > 1000
    n = adj.shape[0]
    x = synth.gaussian_features(n, 256, seed=1)
    w = synth.gcn_weights(256, 256, 2, seed=42)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu),
                           *[torch.from_numpy(w[k]).to(gpu) for k in ("W1", "b1", "W2", "b2")])
    rng = np.random.RandomState(3)
    hub = int(np.argmax(np.diff(a_hat.indptr)))
    probes = np.concatenate([[hub], rng.choice(n, 11, replace=False)])
    observe = np.concatenate([[hub], rng.choice(n, 200, replace=False)])
    ref64 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float64)
    ref32 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float32)
    res = {m: base.influence_rows(probes, observe, 1e-4, m).cpu().numpy().astype(np.float64) for m in ("full", "sparse", "delta")}
    assert np.array_equal(res["full"], res["sparse"])
    assert np.abs(res["delta"] - ref64).max() <= 1e-5 * ref64.max()
    noise_gate("rmat13.full", np.abs(res["full"] - ref64).max() / max(np.abs(ref32 - ref64).max(), 1e-4 * ref64.max()))
    assert np.all(res["full"][ref64 == 0] == 0) and np.all(res["delta"][ref64 == 0] == 0)


def _two_hop_mask(a_hat, probes, observe):
    """mask[i, j] = observe[j] is within two hops of probes[i] in the pattern of a_hat (self loops included)."""
    import scipy.sparse as sp
    pat = sp.csr_matrix((np.ones(a_hat.nnz, np.float32), a_hat.indices, a_hat.indptr), shape=a_hat.shape)
    sel = sp.csr_matrix((np.ones(len(probes), np.float32), (np.arange(len(probes)), probes)),
                        shape=(len(probes), a_hat.shape[0]))
    reach = (sel @ pat.T @ pat.T).toarray() > 0            # column v of pat = rows reading S1[v]
    return reach[:, observe]


@pytest.mark.parametrize("workload,n_test,served", [("twitch-ES", 64, "clean"), ("twitch-RU", 500, "clean"),
                                                    ("twitch-RU", 2000, "clean"), ("twitch-RU", 500, "lapgraph")])
def test_full_size_twitch(gpu, workload, n_test, served):
    """BASELINE configs[0] (twitch-ES shape N=4648, n_test=64), [1], [2] (on one GPU) and [3] at their full sizes:
    twitch-RU shape (N=4385, F=3170, H=256), n_test probes.  The oracle checks a sample of probe rows; the whole matrix is checked through
    size-independent properties: 'sparse' == 'full' bit for bit, exact zeros outside the 2-hop set, non-zero
    inside it (up to ReLU-dead paths), 'delta' close to 'full', symmetry of the support."""
    from linkteller_amd import dp, engine, graph, synth
    adj, x, w = synth.twitch_like_problem(workload, hidden=256, n_classes=2, seed=0)
    if served == "lapgraph":
        adj = dp.perturb_adj(adj, "continuous", 5.0, noise_seed=42)       # worker.py:206-335 (config 4)
    a_hat = graph.first_order_gcn(adj)
    n = adj.shape[0]
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu),
                           *[torch.from_numpy(w[k]).to(gpu) for k in ("W1", "b1", "W2", "b2")])
    nodes = np.sort(np.random.RandomState(7).choice(n, n_test, replace=False))
    res = {m: base.influence_rows(nodes, nodes, 1e-4, m).cpu().numpy() for m in ("full", "sparse", "delta")}
    assert np.array_equal(res["full"], res["sparse"])
    mask = _two_hop_mask(a_hat, nodes, nodes)
    for m, r in res.items():
        assert np.all(np.isfinite(r)), m
        assert np.all(r[~mask] == 0), m
    assert np.array_equal(mask, mask.T)
    assert (res["delta"][mask] > 0).mean() > 0.9
    scale = float(res["delta"].max())
    # the fp32 finite difference (full) sits within its own noise class of the exact perturbation (delta)
    gap = np.abs(res["full"].astype(np.float64) - res["delta"]).max()
    print(f"{workload} n_test={n_test} served={served}: max score {scale:.4g}, |full - delta| max {gap:.4g}, nnz(A_hat) {a_hat.nnz}")
    assert gap <= 0.05 * scale
    sample = np.random.RandomState(11).choice(n_test, 12, replace=False)
    ref64 = _oracle_matrix(a_hat, x, w, nodes[sample], nodes, 1e-4, torch.float64)
    ref32 = _oracle_matrix(a_hat, x, w, nodes[sample], nodes, 1e-4, torch.float32)
    e32 = np.abs(ref32 - ref64).max()
    assert np.abs(res["delta"][sample] - ref64).max() <= 1e-5 * ref64.max()
    # (max over a 12-row sample on both sides is an extreme-value ratio: only the noise-class bound here; BASELINE.md's
    # err(build) <= err(reference fp32) is asserted over the WHOLE matrix in tests/test_gpu_round4.py)
    noise_gate(f"fullsize.{workload}.{n_test}.{served}.full", np.abs(res["full"][sample] - ref64).max() / e32)
    assert np.all(res["full"][sample][ref64 == 0] == 0)


@pytest.mark.parametrize("p,long_par", [(8, 1), (16, 1), (32, 1), (8, 0), (16, 0), (32, 0)])
def test_every_probes_per_wave_variant(gpu, p, long_par):
    """The wide stage-A kernel exists for 8, 16 and 32 probes per wave (picked from the probe count), and hub
    rows take one of two routes (segments in separate waves, long_par = 1, or one wave per row walking its
    segments, long_par = 0: picked from the size of the graph); each combination, pinned through lt_set_tuning,
    must give the bits of `sparse` on a graph with hub rows, for probe counts that leave partial groups."""
    from linkteller_amd import _lib, engine, graph, synth
    adj = synth.powerlaw_graph(700, 4000, seed=5)
    a_hat = graph.first_order_gcn(adj)
    n = adj.shape[0]
    assert np.diff(a_hat.indptr).max() > 300          # rows of several 128-entry segments
    x = synth.gaussian_features(n, 96, seed=2)
    w = synth.gcn_weights(96, 256, 2, seed=3)
    _lib.set_tuning("full_p", p)
    _lib.set_tuning("long_par", long_par)
    _lib.set_tuning("item_bits", long_par)             # 0: no membership bitmap either
    try:
        base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu),
                               *[torch.from_numpy(w[k]).to(gpu) for k in ("W1", "b1", "W2", "b2")])
        rng = np.random.RandomState(1)
        for n_probe in (1, 7, 33, 77):
            probes = rng.choice(n, n_probe, replace=False)
            obs = rng.choice(n, 150, replace=False)
            f = base.influence_rows(probes, obs, 1e-4, "full").cpu().numpy()
            s_ = base.influence_rows(probes, obs, 1e-4, "sparse").cpu().numpy()
            assert np.array_equal(f, s_), n_probe
            assert np.isfinite(f).all() and f.max() > 0
    finally:
        for k in ("full_p", "long_par", "item_bits"):
            _lib.set_tuning(k, None)


def test_row_that_contains_every_probe(gpu):
    """A hub row adjacent to every probe of the first probe group: in FULL mode no chain of that wave is the
    unperturbed one, so the baseline value of the row takes the explicit recomputation path.  Star + ring graph,
    probes = leaves only, observed = everything (the hub included)."""
    import scipy.sparse as sp
    from linkteller_amd import engine, graph, synth
    n = 90
    rows = np.concatenate([np.zeros(n - 1, int), np.arange(1, n)])
    cols = np.concatenate([np.arange(1, n), np.roll(np.arange(1, n), 1)])
    a = sp.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.float32).tocsr()
    a_hat = graph.first_order_gcn(a)
    x = synth.gaussian_features(n, 24, seed=4)
    w = synth.gcn_weights(24, 256, 2, seed=6)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu),
                           *[torch.from_numpy(w[k]).to(gpu) for k in ("W1", "b1", "W2", "b2")])
    observe = np.arange(n)
    for probes in (np.arange(1, 6), np.arange(1, 41), np.array([3]), np.arange(0, 33)):
        res = {m: base.influence_rows(probes, observe, 1e-4, m).cpu().numpy().astype(np.float64)
               for m in ("full", "sparse", "delta")}
        assert np.array_equal(res["full"], res["sparse"])
        ref64 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float64)
        ref32 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float32)
        assert np.abs(res["delta"] - ref64).max() <= 1e-5 * ref64.max()
        noise_gate(f"star.{len(probes)}.{int(probes[0])}.full",
                   np.abs(res["full"] - ref64).max() / max(np.abs(ref32 - ref64).max(), 1e-4 * ref64.max()))
    logits = base.logits().cpu().numpy()
    from oracle import linkteller_oracle as O
    ref_logits = O.gcn_forward(torch.from_numpy(x).double(), O.to_torch_sparse(a_hat).double(),
                               {k: torch.from_numpy(w[k]).double() for k in ("W1", "b1", "W2", "b2")}).numpy()
    assert np.abs(logits - ref_logits).max() <= 2e-5 * np.abs(ref_logits).max() + 1e-6


def test_refresh_picks_up_new_weights_in_every_mode(gpu, influence_golden):
    """lt_baseline_refresh recomputes S1 only and leaves the layer activations stale until something reads
    them; after an in-place change of the (borrowed) weights every consumer -- full rows (which never read
    them), sparse / delta rows, logits -- must equal a baseline created from scratch on the new weights, in any
    order of calls."""
    g = influence_golden
    args, base = _setup(g, "pl600", gpu)
    nodes = g["pl600.ref32.test_nodes"]
    base.influence_rows(nodes, nodes, args["influence"], "delta")        # enables the fp64 copy too
    for order in (("full", "logits", "sparse", "delta"), ("delta", "full", "sparse", "logits"), ("logits", "sparse", "full", "delta")):
        base.w1.mul_(1.01)
        base.b1.add_(0.003)
        base.refresh()
        got = {}
        for what in order:
            got[what] = (base.logits() if what == "logits"
                         else base.influence_rows(nodes, nodes, args["influence"], what)).cpu().numpy()
        from linkteller_amd import engine
        fresh = engine.Baseline(base.graph, base.x, base.w1, base.b1, base.w2, base.b2)
        for what in order:
            ref = (fresh.logits() if what == "logits"
                   else fresh.influence_rows(nodes, nodes, args["influence"], what)).cpu().numpy()
            assert np.array_equal(got[what], ref), (order, what)
        assert np.array_equal(got["full"], got["sparse"])
