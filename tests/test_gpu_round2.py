"""GPU tests added in round 2: the tiled SpMM / layer-1 route, BASELINE configs[4] at its full size, the
INTEGRATION.md ctypes stub executed as written, the multi-rank product path on one device, API-parity methods."""
import os
import re
import socket
import subprocess
import sys
import textwrap
import time

import numpy as np
import pytest
import torch

from conftest import REPO, csr_from, golden_args, load_golden, noise_gate

pytestmark = pytest.mark.gpu


def _params(w, dev):
    return [torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")]


@pytest.fixture
def tiled_everywhere():
    """Forces the large-graph (tiled) SpMM / layer-1 route on graphs of any size."""
    from linkteller_amd import _lib
    _lib.set_tuning("tiled_min_bytes", 0)
    yield
    _lib.set_tuning("tiled_min_bytes", None)


@pytest.mark.parametrize("ncols,bias,relu", [(256, True, True), (256, False, False), (64, True, False), (20, False, True),
                                              (132, True, False), (192, False, False)])
def test_tiled_spmm_equals_row_kernel_bit_for_bit(gpu, ncols, bias, relu):
    """Both routes of lt_spmm_csr_f32 sum a row in the same order (128-entry fmaf chains added in segment order), so
    the column-sliced work-item kernel must give the bits of the one-group-per-row kernel -- hub rows included."""
    from test_gpu_parity import _hub_graph
    from linkteller_amd import _lib, engine, graph
    a_hat = graph.first_order_gcn(_hub_graph(2500, 12000, 1400, seed=5))
    hg = graph.HipGraph(a_hat)
    rng = np.random.RandomState(1)
    s = torch.from_numpy(rng.standard_normal((a_hat.shape[0], ncols)).astype(np.float32)).to(gpu)
    b = torch.from_numpy(rng.standard_normal(ncols).astype(np.float32)).to(gpu) if bias else None
    small = engine.spmm(hg, s, b, relu=relu).cpu().numpy()
    _lib.set_tuning("tiled_min_bytes", 0)
    try:
        tiled = engine.spmm(hg, s, b, relu=relu).cpu().numpy()
    finally:
        _lib.set_tuning("tiled_min_bytes", None)
    assert np.array_equal(small, tiled)
    _lib.set_tuning("tiled_min_bytes", 0)
    _lib.set_tuning("tiled_big", 1)                       # the 64-bit offset instantiation (S of 4 GiB and more)
    try:
        assert np.array_equal(engine.spmm(hg, s, b, relu=relu).cpu().numpy(), small)
    finally:
        _lib.set_tuning("tiled_min_bytes", None)
        _lib.set_tuning("tiled_big", None)
    want = a_hat.astype(np.float64) @ s.cpu().numpy().astype(np.float64)
    if bias:
        want = want + b.cpu().numpy()
    if relu:
        want = np.maximum(want, 0)
    assert np.abs(tiled - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("key", ["pl600", "er300"])
def test_tiled_layer1_keeps_every_bit(gpu, influence_golden, key):
    """The baseline forward through the tiled layer-1 route (Z1 from the work-item kernel, then the finishing
    passes) equals the fused row kernel bit for bit: logits, and full == sparse still holds on top of it."""
    from test_gpu_parity import _setup
    from linkteller_amd import _lib
    g = influence_golden
    args, base = _setup(g, key, gpu)
    nodes = g[f"{key}.ref32.test_nodes"]
    ref_logits = base.logits().cpu().numpy()
    ref = {m: base.influence_rows(nodes, nodes, args["influence"], m).cpu().numpy() for m in ("full", "sparse", "delta")}
    _lib.set_tuning("tiled_min_bytes", 0)
    try:
        args2, base2 = _setup(g, key, gpu)
        assert np.array_equal(base2.logits().cpu().numpy(), ref_logits)
        for m in ("sparse", "delta", "full"):
            assert np.array_equal(base2.influence_rows(nodes, nodes, args["influence"], m).cpu().numpy(), ref[m]), m
    finally:
        _lib.set_tuning("tiled_min_bytes", None)


def test_tiled_route_on_hub_graph_and_gcn2_forward(gpu, tiled_everywhere):
    from test_gpu_parity import _hub_graph
    from linkteller_amd import _lib, engine, graph, synth
    adj = _hub_graph(1500, 6000, 700, seed=9)
    x = torch.from_numpy(synth.twitch_like_features(1500, 64, seed=3, density=0.05)).to(gpu)
    w = synth.gcn_weights(64, 200, 3, seed=42)                      # Hp = 200: four slices, the last one partial
    hg = graph.HipGraph(graph.first_order_gcn(adj))
    tiled = engine.gcn2_forward(hg, x, *_params(w, gpu)).cpu().numpy()
    base_t = engine.Baseline(hg, x, *_params(w, gpu))
    assert np.array_equal(base_t.logits().cpu().numpy(), tiled)
    _lib.set_tuning("tiled_min_bytes", None)
    fused = engine.gcn2_forward(hg, x, *_params(w, gpu)).cpu().numpy()
    _lib.set_tuning("tiled_min_bytes", 0)
    assert np.array_equal(tiled, fused)


def test_get_gradient_eps_mat_matches_oracle(gpu, influence_golden):
    """The API-parity method Attacker.get_gradient_eps_mat (attacker.py:100-108): [N, C] finite difference of one
    probe, against the oracle's verbatim restatement in fp64 (fp32 noise class) and its rows' norms against the
    batched primitive."""
    import argparse
    import types
    from linkteller_amd import graph
    from linkteller_amd.attacker import Attacker
    from linkteller_amd.gcn import GCN
    from oracle import linkteller_oracle as O
    g, key = influence_golden, "pl600"
    args = golden_args(g, key)
    a = csr_from(g, f"{key}.adj")
    x = torch.from_numpy(g[f"{key}.x"]).to(gpu)
    sd = {k: torch.from_numpy(g[f"{key}.sd.{k}"]) for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias")}
    model = GCN(x.shape[1], sd["gc1.weight"].shape[1], sd["gc2.weight"].shape[1], 0.5)
    model.load_state_dict(sd)
    model.to(gpu).eval()
    adj_t = graph.sparse_mx_to_torch_sparse_tensor(graph.fetch_normalization(args["norm"])(a)).to(gpu)
    worker = types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=a, n_nodes=a.shape[0])
    ns = argparse.Namespace(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=10, sample_seed=42,
                            influence=args["influence"], mode="vanilla-clean", attack_mode="efficient")
    atk = Attacker(ns, model, worker)
    P64 = {k: sd[n].double() for k, n in (("W1", "gc1.weight"), ("b1", "gc1.bias"), ("W2", "gc2.weight"), ("b2", "gc2.bias"))}
    P32 = {k: v.float() for k, v in P64.items()}
    adj_o = O.to_torch_sparse(graph.fetch_normalization(args["norm"])(a))
    x_cpu = x.cpu()
    for v in (int(g[f"{key}.ref32.test_nodes"][0]), 17):
        got = atk.get_gradient_eps_mat(v).cpu().numpy().astype(np.float64)
        with torch.no_grad():
            ref64 = O.get_gradient_eps_mat(x_cpu.double(), adj_o.double(), P64, v, args["influence"]).numpy()
            ref32 = O.get_gradient_eps_mat(x_cpu, adj_o, P32, v, args["influence"]).numpy().astype(np.float64)
        assert got.shape == ref64.shape
        e32 = np.abs(ref32 - ref64).max()
        noise_gate(f"eps_mat.pl600.v{v}.logit_diff", np.abs(got - ref64).max() / e32)
        assert np.all(got[np.all(ref64 == 0, axis=1)] == 0)          # untouched rows: exactly zero
        rows = atk.baseline().influence_rows([v], np.arange(a.shape[0]), args["influence"], "full").cpu().numpy()[0]
        # (two of OUR outputs: the row norms of the [N, C] difference against the batched primitive's row -- fp32 rounding of a norm)
        noise_gate(f"eps_mat.pl600.v{v}.norm_vs_rows", np.abs(np.linalg.norm(got, axis=1) - rows).max() / e32, ceiling=1e-4)


def test_integration_md_stub_runs_as_written(gpu, influence_golden, tmp_path):
    """INTEGRATION.md section B shows the ctypes stub a reference maintainer would add.  The code block is extracted
    and executed verbatim (fresh process, liblinkteller_hip.so found through LD_LIBRARY_PATH as a maintainer would
    install it) and its influence matrix is checked against the reference's golden fp64 matrix."""
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# reference tree: lt_hip\.py.*?)```", md, re.S).group(1)
    (tmp_path / "lt_hip.py").write_text(block)
    driver = textwrap.dedent('''
        import sys, numpy as np, torch, scipy.sparse as sp
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import lt_hip                                     # the stub, verbatim
        from conftest import csr_from, golden_args, load_golden
        from linkteller_amd import graph
        from linkteller_amd.gcn import GCN
        g = load_golden("influence.npz"); key = "pl600"
        args = golden_args(g, key)
        a_hat = graph.fetch_normalization(args["norm"])(csr_from(g, key + ".adj"))
        adj_coo = graph.sparse_mx_to_torch_sparse_tensor(a_hat)          # what the reference holds (utils/load.py:552-559)
        x = torch.from_numpy(g[key + ".x"]).cuda()
        sd = {k: torch.from_numpy(g[key + ".sd." + k]) for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias")}
        model = GCN(x.shape[1], sd["gc1.weight"].shape[1], sd["gc2.weight"].shape[1], 0.5)
        model.load_state_dict(sd); model.cuda().eval()
        nodes = g[key + ".ref32.test_nodes"]
        out = {m: lt_hip.influence_matrix(model, x, adj_coo, nodes, args["influence"], mode=m) for m in (0, 1, 2)}
        np.savez(%r, full=out[0], sparse=out[1], delta=out[2])
        print("stub ok")
    ''') % (str(tmp_path), os.path.join(REPO, "tests"), str(tmp_path / "out.npz"))
    env = dict(os.environ, PYTHONPATH=REPO, LD_LIBRARY_PATH=os.path.join(REPO, "linkteller_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([sys.executable, "-c", driver], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "stub ok" in r.stdout, r.stdout + r.stderr
    out = np.load(tmp_path / "out.npz")
    g = influence_golden
    ref64, ref32 = g["pl600.ref64.influence_val"], g["pl600.ref32.influence_val"]
    assert np.array_equal(out["full"], out["sparse"])
    assert np.abs(out["delta"] - ref64).max() <= 1e-4 * ref64.max()      # (no fp64 kink copy in the minimal stub)
    noise_gate("integration_stub.pl600.full", np.abs(out["full"] - ref64).max() / np.abs(ref32 - ref64).max())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(code, world, extra_env=None, timeout=900):
    """`code` in `world` fresh processes, all on device 0 over gloo (RCCL refuses two ranks on one GPU)."""
    port = _free_port()
    procs = []
    for r in range(world):
        # (LT_SHARD_PROBES=1 unless the caller says otherwise: these tests are about the sharded path; under the default `auto` ranks
        # that share ONE device would find the local build faster -- tests/test_gpu_round6.py covers the policy)
        env = dict(os.environ, PYTHONPATH=REPO + ":" + os.path.join(REPO, "tests"), RANK=str(r), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LT_DIST_BACKEND="gloo",
                   LT_DIST_DEVICE="0", **{"LT_SHARD_PROBES": "1", **(extra_env or {})})
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(outs)
    return outs


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_influence_matrix_equals_single_rank(gpu, influence_golden, tmp_path, world):
    """'sharded probe loop + all-gather equals the single-GPU result bit for bit' -- the PRODUCT
    path (Attacker.influence_matrix: shard_bounds + lt_influence_rows on the slice + one all-gather), with the ranks
    as separate processes on one device; also with the baseline's X*W1 sharded over the ranks."""
    code = textwrap.dedent('''
        import argparse, os, types, numpy as np, torch
        from conftest import csr_from, golden_args, load_golden
        from linkteller_amd import graph, main as lt_main
        from linkteller_amd.attacker import Attacker
        from linkteller_amd.gcn import GCN
        assert lt_main.init_distributed()
        import torch.distributed as dist
        g = load_golden("influence.npz"); key = "pl600"
        args = golden_args(g, key)
        a = csr_from(g, key + ".adj")
        x = torch.from_numpy(g[key + ".x"]).cuda()
        sd = {k: torch.from_numpy(g[key + ".sd." + k]) for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias")}
        model = GCN(x.shape[1], sd["gc1.weight"].shape[1], sd["gc2.weight"].shape[1], 0.5)
        model.load_state_dict(sd); model.cuda().eval()
        adj_t = graph.sparse_mx_to_torch_sparse_tensor(graph.fetch_normalization(args["norm"])(a)).cuda()
        w = types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=a, n_nodes=a.shape[0])
        ns = argparse.Namespace(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=len(g[key + ".ref32.test_nodes"]),
                                sample_seed=42, influence=args["influence"], mode="vanilla-clean", attack_mode="efficient")
        atk = Attacker(ns, model, w)
        atk.test_nodes = g[key + ".ref32.test_nodes"]
        res = {m: atk.influence_matrix(m) for m in ("full", "sparse", "delta")}
        if dist.get_rank() == 0:
            np.savez(os.environ["LT_TEST_OUT"], **res)
        dist.barrier(); dist.destroy_process_group()
    ''')
    from test_gpu_parity import _setup
    g = influence_golden
    args, base = _setup(g, "pl600", gpu)
    nodes = g["pl600.ref32.test_nodes"]
    single = {m: base.influence_rows(nodes, nodes, args["influence"], m).cpu().numpy().astype(np.float64)
              for m in ("full", "sparse", "delta")}
    # auto: both are timed, the faster kept; "1d": the fp64 product of `delta` forced onto the matrix cores
    # (LT_FEATURE_DELTA=0), so that it is the product that gets sharded (lt_baseline_refresh_rows_fp64 + all-gather of S1d)
    for shard_baseline in ("0", "1", "1d") + (("auto",) if world == 2 else ()):
        out = tmp_path / f"ranks{world}_{shard_baseline}.npz"
        env = {"LT_TEST_OUT": str(out), "LT_SHARD_BASELINE": shard_baseline[0] if shard_baseline != "auto" else "auto"}
        want = single
        if shard_baseline == "1d":
            env["LT_FEATURE_DELTA"] = env["LT_AGGREGATE_FIRST"] = "0"
            from linkteller_amd import _lib
            _lib.set_tuning("feature_delta", 0)
            _lib.set_tuning("aggregate_first", 0)
            try:
                base.refresh()
                assert base.fp64_route() == 0
                single_d = base.influence_rows(nodes, nodes, args["influence"], "delta").cpu().numpy().astype(np.float64)
            finally:
                _lib.set_tuning("feature_delta", None)
                _lib.set_tuning("aggregate_first", None)
                base.refresh()
            want = dict(single, delta=single_d)      # (the matrix-core route keeps its product rows in its own storage form)
        _run_ranks(code, world, env)
        got = np.load(out)
        for m in ("full", "sparse", "delta"):
            assert got[m].shape == want[m].shape and np.array_equal(got[m], want[m]), (world, shard_baseline, m)


def test_cli_two_ranks_write_one_result_file(gpu, tmp_path):
    """`torchrun --nproc-per-node 2 -m linkteller_amd.main ... --attack`: both ranks join
    the group, shard the probes, and only rank 0 writes the result file, whose content equals the one-process run."""
    from linkteller_amd import synth
    from linkteller_amd.gcn import GCN
    a1, a2 = synth.powerlaw_graph(260, 1200, seed=1), synth.powerlaw_graph(320, 1500, seed=2)
    synth.write_musae_dataset(str(tmp_path), "ES", a1, 400, 1)
    synth.write_musae_dataset(str(tmp_path), "RU", a2, 400, 2)
    torch.manual_seed(0)
    torch.save(GCN(3170, 256, 2, 0.5).state_dict(), tmp_path / "model.pt")
    argv = (f"--mode vanilla-clean --dataset twitch/ES/RU --hidden 256 --norm FirstOrderGCN --test "
            f"--model-path {tmp_path}/model.pt --attack --attack-mode efficient --sample-type unbalanced "
            f"--n-test 61 --data-root {tmp_path}").split()
    code = "import os, sys; os.chdir(os.environ['LT_TEST_CWD']); from linkteller_amd import main as m; m.main(%r)" % (argv,)
    files = {}
    for world in (1, 2):
        cwd = tmp_path / f"w{world}"
        cwd.mkdir()
        outs = _run_ranks(code, world, {"LT_TEST_CWD": str(cwd)})
        assert sum("attack results saved to:" in o for o in outs) == 1          # rank 0 only
        files[world] = torch.load(cwd / "eval_twitch/ES/RU/efficient_unbalanced_61_42.pt", weights_only=False)
    assert files[1]["result"]["y"] == files[2]["result"]["y"]
    assert np.array_equal(np.asarray(files[1]["result"]["pred"]), np.asarray(files[2]["result"]["pred"]))


@pytest.mark.slow      # 150 s (host graph build + fp64 oracle rows): LT_RUN_SLOW=1 / tools/round_artifacts.sh; default set: test_rmat_shape_scaled_config5
def test_rmat_scale21_config5_full_size(gpu):
    """BASELINE configs[4] at its full size on ONE GPU: R-MAT scale 21 (2 097 152 nodes, 40 M directed draws -> nnz(A_hat)
    ~ 78 M, hub rows of 10^5 entries), F = H = 256, C = 2.  Size-independent properties -- `sparse` == `full` bit for
    bit, the tiled SpMM == the row kernel bit for bit, the tiled layer 1 == the fused one through the logits, exact
    zeros off the 2-hop set -- at the per-rank shape of the config (512 probes x 4096 observed, hubs on both sides), and the
    fp64 oracle on 8 probe rows.  Timings go to gpurun_out/ (copied to profiles/)."""
    from linkteller_amd import _lib, engine, graph, synth
    from oracle import linkteller_oracle as O
    t0 = time.time()
    adj = synth.rmat_graph(21, synth.rmat_draws(21), seed=42)
    a_hat = graph.first_order_gcn(adj)
    n = adj.shape[0]
    deg = np.diff(a_hat.indptr)
    assert n == 1 << 21 and a_hat.nnz > 70_000_000 and deg.max() > 50_000
    x_np = synth.gaussian_features(n, 256, seed=1)
    w = synth.gcn_weights(256, 256, 2, seed=42)
    t_host = time.time() - t0
    x = torch.from_numpy(x_np).to(gpu)
    t0 = time.time()
    hg = graph.HipGraph(a_hat)
    base = engine.Baseline(hg, x, *_params(w, gpu))
    torch.cuda.synchronize()
    t_create = time.time() - t0
    log = [f"rmat scale 21: n {n} nnz {a_hat.nnz} max row {int(deg.max())}; host graph+features {t_host:.1f} s; "
           f"lt_graph_create + lt_baseline_create {t_create:.2f} s"]

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        t = time.time()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return r, (time.time() - t) / reps

    # standalone SpMM: tiled (default at this size) vs the row kernel
    s = torch.randn((n, 256), device=gpu)
    out_t, t_t = timed(lambda: engine.spmm(hg, s))
    _lib.set_tuning("tiled_min_bytes", 1 << 62)
    try:
        out_r, t_r = timed(lambda: engine.spmm(hg, s))
        logits_rows = base.logits().clone()                         # layer 1 through the fused row kernel
    finally:
        _lib.set_tuning("tiled_min_bytes", None)
    assert torch.equal(out_t, out_r)
    alg = a_hat.nnz * 8 + (n + 1) * 4 + 2 * n * 256 * 4
    log.append(f"lt_spmm_csr_f32 H=256: tiled {t_t * 1e3:.3f} ms = {alg / t_t / 1e9:.0f} GB/s algorithmic; "
               f"row kernel {t_r * 1e3:.3f} ms = {alg / t_r / 1e9:.0f} GB/s")
    del s, out_t, out_r
    base.refresh()
    logits_tiled = base.logits()
    assert torch.equal(logits_tiled, logits_rows)
    # The PER-RANK SHAPE of configs[4] (n_test = 4096 over 8 GPUs): 512 probes x 4096 observed nodes, the probes being the
    # first 512 of the observed list as in the sharded product path; the three biggest hubs sit on BOTH sides.
    import scipy.sparse as sp
    rng = np.random.RandomState(3)
    hubs = np.argsort(-deg)[:3].astype(np.int64)
    rest = rng.choice(np.setdiff1d(np.arange(n), hubs), 4096 - len(hubs), replace=False)
    observe = np.concatenate([hubs, rest])
    probes = observe[:512]
    res = {}
    for m in ("full", "sparse", "delta"):
        res[m], t = timed(lambda: base.influence_rows(probes, observe, 1e-4, m), reps=1)
        log.append(f"{m}: {len(probes)} probes x {len(observe)} observed: {t * 1e3:.2f} ms")
    full, sparse, delta = (res[m].cpu().numpy() for m in ("full", "sparse", "delta"))
    assert np.array_equal(full, sparse)
    assert np.isfinite(full).all() and np.isfinite(delta).all()
    # exact zeros off the 2-hop set (attacker.py:220-229: identical inputs -> identical outputs), every mode; the mask is
    # the pattern product restricted to the observed columns: reach = (E_probes P^T) (P^T)[:, observe]
    pat = sp.csr_matrix((np.ones(a_hat.nnz, np.float32), a_hat.indices, a_hat.indptr), shape=a_hat.shape)
    r1 = pat.T.tocsr()[probes]                                   # row i: the rows r with A_hat[r, probes[i]] != 0
    mask = np.asarray((r1 @ pat[observe].T.tocsc()).todense()) > 0
    for m, r in (("full", full), ("delta", delta)):
        assert np.all(r[~mask] == 0), m
    assert (delta[mask] > 0).mean() > 0.9
    log.append(f"2-hop pairs {int(mask.sum())} of {mask.size}; non-zero scores: full {int((full > 0).sum())}, delta {int((delta > 0).sum())}")
    assert np.abs(full - delta).max() <= 0.05 * delta.max()
    # fp64 oracle on 10 probe rows (the biggest hub, a mid-degree node, eight random ones): the restricted restatement
    # (oracle.RestrictedOracle: the baseline forward once, then only the rows a probe can change -- pinned against the
    # verbatim one in tests/test_oracle_golden.py), and the VERBATIM reference op sequence (attacker.py:100-108 + the norm
    # of :227-229: two full fp64 forwards of the 2 M-node graph per probe, ~50 s each here) on two of them, which also
    # checks the restricted oracle itself at this size
    pdeg = deg[probes]
    rows = np.unique(np.concatenate([[0, int(np.argsort(pdeg)[len(pdeg) // 2])], rng.choice(512, 8, replace=False)]))[:10]
    t0 = time.time()
    ro = O.RestrictedOracle(x_np, a_hat, w)
    ref_rows = ro.rows(probes[rows], observe, 1e-4)
    t_restricted = time.time() - t0
    worst = 0.0
    for k, i in enumerate(rows):
        ref64 = ref_rows[k]
        err = np.abs(delta[i] - ref64).max() / max(ref64.max(), 1e-3)
        worst = max(worst, err)
        assert err <= 1e-5, (int(i), int(probes[i]), err)
        assert np.all(full[i][ref64 == 0] == 0) and np.all(delta[i][ref64 == 0] == 0)
        assert np.all(mask[i][ref64 > 0])
    del ro
    t0 = time.time()
    P64 = {k: torch.from_numpy(w[k]).double() for k in ("W1", "b1", "W2", "b2")}
    adj_o = O.to_torch_sparse(a_hat).double()
    x64 = torch.from_numpy(x_np).double()
    for k in (0, len(rows) - 1):
        with torch.no_grad():
            gm = O.get_gradient_eps_mat(x64, adj_o, P64, int(probes[rows[k]]), 1e-4)
            verb = gm[torch.as_tensor(observe)].norm(dim=1).numpy()
        assert np.abs(verb - ref_rows[k]).max() <= 1e-8 * max(verb.max(), 1e-3), int(rows[k])
        assert np.abs(delta[rows[k]] - verb).max() <= 1e-5 * max(verb.max(), 1e-3)
    log.append(f"fp64 oracle on probe rows {rows.tolist()} (degrees {pdeg[rows].tolist()}): worst |delta - ref64| / max = {worst:.2e} "
               f"(restricted oracle {t_restricted:.0f} s; verbatim op sequence on 2 of them {time.time() - t0:.0f} s)")
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "rmat_scale21_test.txt"), "w") as fh:
        fh.write("\n".join(log) + "\n")
    print("\n".join(log))


@pytest.mark.parametrize("h1,h2,c,hub", [(32, 16, 2, False), (100, 36, 3, True), (256, 64, 2, True), (20, 256, 8, False)])
def test_gcn3_probe_primitive_against_oracle(gpu, h1, h2, c, hub):
    """lt_influence3_rows on graphs with isolated nodes / a hub row (several 128-entry segments in every layer),
    widths that need padding, 1..8 classes, duplicate probes, observe list != probe list: the fp32 finite difference
    of the reference's 3-layer forward (oracle, fp64 and fp32), exact zeros outside the 3-hop set."""
    import scipy.sparse as sp
    from test_gpu_parity import _hub_graph
    from linkteller_amd import engine, graph, synth
    from oracle import linkteller_oracle as O
    n, f = (700, 48) if hub else (220, 40)
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
    x = synth.gaussian_features(n, f, seed=3)
    rng = np.random.RandomState(h2)

    def u(shape, fan):
        s = 1.0 / np.sqrt(fan)
        return rng.uniform(-s, s, size=shape).astype(np.float32)

    P = dict(W1=u((f, h1), h1), b1=u((h1,), h1), W2=u((h1, h2), h2), b2=u((h2,), h2), W3=u((h2, c), c), b3=u((c,), c))
    dev_p = [torch.from_numpy(P[k]).to(gpu) for k in ("W1", "b1", "W2", "b2", "W3", "b3")]
    base = engine.Baseline3(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *dev_p)
    probes = np.concatenate([rng.choice(n, 19, replace=False), [0, 5], [3, 3]])
    observe = np.concatenate([rng.choice(n, 40, replace=False), [0, 99]])
    got = base.influence_rows(probes, observe, 1e-4).cpu().numpy().astype(np.float64)
    adj_t = O.to_torch_sparse(a_hat)
    ref = {}
    for dt in (torch.float64, torch.float32):
        Pd = {k: torch.from_numpy(v).to(dt) for k, v in P.items()}
        xt = torch.from_numpy(x).to(dt)
        m = np.zeros((len(probes), len(observe)))
        with torch.no_grad():
            for i, v in enumerate(probes):
                gm = O.get_gradient_eps_mat(xt, adj_t.to(dt), Pd, int(v), 1e-4, forward=O.gcn3_forward)
                m[i] = gm[torch.as_tensor(observe)].norm(dim=1).numpy()
        ref[dt] = m
    e32 = np.abs(ref[torch.float32] - ref[torch.float64]).max()
    scale = ref[torch.float64].max()
    print(f"gcn3 h1={h1} h2={h2} c={c} hub={hub}: max {scale:.3g} |ref32-ref64| {e32:.2e} |ours-ref64| {np.abs(got - ref[torch.float64]).max():.2e}")
    noise_gate(f"gcn3.h{h1}.{h2}.c{c}.hub{int(hub)}", np.abs(got - ref[torch.float64]).max() / max(e32, 1e-4 * scale))
    assert np.all(got[ref[torch.float64] == 0] == 0)
    assert np.array_equal(got[-1], got[-2])                         # duplicate probe -> identical rows
    logits = base.logits().cpu().numpy().astype(np.float64)
    ref_logits = O.gcn3_forward(torch.from_numpy(x).double(), adj_t.double(), {k: torch.from_numpy(v).double() for k, v in P.items()}).numpy()
    assert np.abs(logits - ref_logits).max() <= 2e-5 * max(1.0, np.abs(ref_logits).max())
    # a tiny chunk budget splits the probe list: same bits
    from linkteller_amd import _lib
    _lib.set_tuning("chunk_budget_bytes", 64 * 1024)
    try:
        base2 = engine.Baseline3(base.graph, base.x, *dev_p)
        assert np.array_equal(base2.influence_rows(probes, observe, 1e-4).cpu().numpy().astype(np.float64), got)
    finally:
        _lib.set_tuning("chunk_budget_bytes", None)


@pytest.mark.parametrize("seed,tiled", [(0, False), (1, True)])
def test_randomised_cross_check(gpu, seed, tiled, monkeypatch):
    """tools/fuzz_gpu.py inside the suite: 20 random (graph family, width, classes, normaliser, probe / observe list)
    combinations per seed -- full == sparse bit for bit, delta within 1e-5 of the fp64 oracle, exact zeros, logits;
    once on the default routes, once with the tiled SpMM / layer-1 route forced on every graph."""
    import importlib.util
    from linkteller_amd import _lib
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(REPO, "tools", "fuzz_gpu.py"))
    fuzz = importlib.util.module_from_spec(spec)
    monkeypatch.chdir(REPO)
    spec.loader.exec_module(fuzz)
    monkeypatch.setattr(sys, "argv", ["fuzz_gpu.py", "20", str(seed)])
    if tiled:
        _lib.set_tuning("tiled_min_bytes", 0)
    try:
        fuzz.main()
    finally:
        _lib.set_tuning("tiled_min_bytes", None)


def test_auc_ap_at_config1_size(gpu):
    """north_star: 'influence scores and AUC within 1e-4' -- at the BASELINE configs[1] graph size (twitch-RU shape,
    N = 4385, F = 3170, H = 256), the reference's own sampler (unbalanced, seed 42) and pair lookup, n_test = 160 so that
    the fp64 oracle (verbatim reference op sequence, 2 x 160 full forwards in fp64) stays within a minute: AUC and AP of
    the delta-mode scores equal the fp64 reference's within 1e-4; full / sparse within their quantisation class."""
    from linkteller_amd import engine, graph, synth
    from linkteller_amd.sampling import construct_edge_sets_from_random_subgraph
    from oracle import linkteller_oracle as O
    adj, x, w = synth.twitch_like_problem("twitch-RU", hidden=256, n_classes=2, seed=0)
    a_hat = graph.first_order_gcn(adj)
    np.random.seed(42)
    (ex, nex), nodes = construct_edge_sets_from_random_subgraph("twitch/ES/RU", "unbalanced", adj, 160)
    nodes = np.asarray(nodes, dtype=np.int64)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu))
    res = {m: base.influence_rows(nodes, nodes, 1e-4, m).cpu().numpy().astype(np.float64) for m in ("delta", "full", "sparse")}
    assert np.array_equal(res["full"], res["sparse"])
    t0 = time.time()
    P64 = {k: torch.from_numpy(w[k]).double() for k in ("W1", "b1", "W2", "b2")}
    ref64 = O.influence_matrix(torch.from_numpy(x).double(), O.to_torch_sparse(a_hat).double(), P64, nodes, 1e-4)
    t_oracle = time.time() - t0
    m64 = O.attack_metrics(*O.pair_scores(ref64, list(nodes), ex, nex))
    got = {m: O.attack_metrics(*O.pair_scores(res[m], list(nodes), ex, nex)) for m in ("delta", "full")}
    print(f"configs[1] size, n_test=160 ({len(ex)} edges / {len(nex)} non-edges; fp64 oracle {t_oracle:.0f} s): "
          f"AUC ref64 {m64['auc']:.6f} delta {got['delta']['auc']:.6f} full {got['full']['auc']:.6f}; "
          f"AP ref64 {m64['ap']:.6f} delta {got['delta']['ap']:.6f} full {got['full']['ap']:.6f}; "
          f"max|delta-ref64| {np.abs(res['delta'] - ref64).max():.2e} on max score {ref64.max():.3g}")
    assert np.abs(res["delta"] - ref64).max() <= 1e-5 * ref64.max()
    assert abs(got["delta"]["auc"] - m64["auc"]) <= 1e-4 and abs(got["delta"]["ap"] - m64["ap"]) <= 1e-4
    assert np.all(res["full"][ref64 == 0] == 0)
    assert abs(got["full"]["auc"] - m64["auc"]) <= 2.5 / max(len(ex), 1)      # a couple of low-score edges quantised to 0


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["powerlaw", "er", "directed"])
def test_pair_marks_route_keeps_every_bit(gpu, family):
    """Large SPARSE / DELTA calls find the affected (probe, observed) pairs through a join over the middle nodes
    (`pair_marks`, lt_items.hip.h) instead of a membership scan per pair.  Forced on (0) and off (-1) on small graphs
    -- hub rows on both sides, a non-symmetric pattern, several probe chunks, with and without the membership
    bitmap, duplicated observed nodes -- the matrices must agree bit for bit, and `sparse` must still equal `full`."""
    import scipy.sparse as sp
    from linkteller_amd import _lib, engine, graph, synth
    if family == "powerlaw":
        a_hat = graph.first_order_gcn(synth.powerlaw_graph(900, 6000, seed=7))
        assert np.diff(a_hat.indptr).max() > 300
    elif family == "er":
        a_hat = graph.first_order_gcn(synth.erdos_renyi_graph(1200, 9000, seed=3))
    else:   # a pattern that is NOT symmetric: paths u - r - v must follow rows on one side and columns on the other
        rng = np.random.RandomState(5)
        n = 800
        m = sp.random(n, n, density=0.01, random_state=rng, format="csr", dtype=np.float32)
        m.data[:] = rng.uniform(0.05, 0.5, m.nnz).astype(np.float32)
        a_hat = (m + sp.eye(n, dtype=np.float32, format="csr") * 0.3).tocsr()
        a_hat.sort_indices()
        assert (a_hat != a_hat.T).nnz > 0
    n = a_hat.shape[0]
    x = synth.gaussian_features(n, 64, seed=2)
    w = synth.gcn_weights(64, 256, 3, seed=3)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu))
    rng = np.random.RandomState(1)
    deg = np.diff(a_hat.indptr)
    hub = int(np.argmax(deg))
    try:
        for n_probe, n_obs, budget, bits in ((1, 1, None, 1), (37, 150, None, 1), (90, 333, 1 << 16, 1), (64, 200, None, 0)):
            probes = rng.choice(n, n_probe, replace=False)
            obs = rng.choice(n, n_obs, replace=False)
            if n_obs > 10:
                obs[0] = hub                     # an observed hub (served by the long blocks, not listed)
                obs[5] = obs[6]                  # a duplicate
                probes[0] = hub if n_probe > 1 else probes[0]
            _lib.set_tuning("chunk_budget_bytes", budget)
            _lib.set_tuning("item_bits", bits)
            res = {}
            for pm in (-1, 0):
                _lib.set_tuning("pair_marks", pm)
                # observed hubs: every entry tested against every probe (0) / members found from the shorter list (1)
                _lib.set_tuning("hub_short_side", 0 if pm < 0 else 1)
                for mode in ("sparse", "delta"):
                    res[pm, mode] = base.influence_rows(probes, obs, 1e-4, mode).cpu().numpy()
            for mode in ("sparse", "delta"):
                assert np.array_equal(res[-1, mode], res[0, mode]), (family, n_probe, n_obs, mode)
            full = base.influence_rows(probes, obs, 1e-4, "full").cpu().numpy()
            assert np.array_equal(full, res[0, "sparse"])
            assert np.isfinite(full).all() and (n_probe == 1 or full.max() > 0)
    finally:
        for k in ("chunk_budget_bytes", "item_bits", "pair_marks", "hub_short_side"):
            _lib.set_tuning(k, None)


@pytest.mark.gpu
def test_big_probe_bitmap_rows_keep_every_bit(gpu):
    """A call too large for a bitmap row per probe (forced here: bits_max_bytes = 1) gives rows to its big probes only
    (|R_v| > 512, at most 64 per chunk); all other probes search.  Hubs probed AND observed -- every combination of
    light / heavy probe with short / long observed row -- must give the bits of the all-bitmap route, for more big
    probes than there are slots too."""
    from linkteller_amd import _lib, engine, graph, synth
    a_hat = graph.first_order_gcn(synth.powerlaw_graph(8000, 300000, seed=13, exponent=1.8))
    deg = np.diff(a_hat.indptr)
    n = a_hat.shape[0]
    big = np.argsort(-deg)[:80]
    assert deg[big[70]] > 512, int(deg[big[70]])       # more big probes than the 64 slots
    x = synth.gaussian_features(n, 48, seed=2)
    w = synth.gcn_weights(48, 128, 2, seed=3)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu))
    rng = np.random.RandomState(4)
    probes = np.concatenate([big, rng.choice(np.setdiff1d(np.arange(n), big), 120, replace=False)])
    obs = np.concatenate([big[:20], rng.choice(n, 300, replace=False)])
    try:
        ref = {m: base.influence_rows(probes, obs, 1e-4, m).cpu().numpy() for m in ("sparse", "delta")}
        _lib.set_tuning("bits_max_bytes", 1)
        for pm in (0, -1):
            _lib.set_tuning("pair_marks", pm)
            for hs in (1, 0):
                _lib.set_tuning("hub_short_side", hs)
                for m in ("sparse", "delta"):
                    got = base.influence_rows(probes, obs, 1e-4, m).cpu().numpy()
                    assert np.array_equal(got, ref[m]), (pm, hs, m)
        assert ref["sparse"].max() > 0 and np.isfinite(ref["sparse"]).all()
    finally:
        for k in ("bits_max_bytes", "pair_marks", "hub_short_side"):
            _lib.set_tuning(k, None)


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,n", [(1500, 256, 256), (2048, 77, 128), (1031, 300, 256)])
def test_gemm_tile_routes_give_the_same_rows(gpu, m, k, n):
    """`lt_gemm_f32` serves a tall product (M >= 1024, N a multiple of 128: X*W1 of a 2 M-node graph with F = 256) with
    the 128 x 128 tiles and everything else with 64 x 64 ones.  Both sum a row's products in the same k order, so a row
    has the same bits whichever route the call's shape selects -- and both sit within fp32 rounding of the fp64 product."""
    from linkteller_amd import engine
    rng = np.random.RandomState(m + k)
    a = rng.standard_normal((m, k)).astype(np.float32)
    b = rng.standard_normal((k, n)).astype(np.float32)
    at, bt = torch.from_numpy(a).to(gpu), torch.from_numpy(b).to(gpu)
    tall = engine.gemm(at, bt).cpu().numpy()                     # 128-tile route
    for r0, r1 in ((0, 700), (700, m)):                          # < 1024 rows per call: 64-tile route
        part = engine.gemm(at[r0:r1].contiguous(), bt).cpu().numpy()
        assert np.array_equal(part, tall[r0:r1]), (r0, r1)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    assert np.abs(tall - ref).max() <= 2e-6 * np.abs(a).astype(np.float64).dot(np.abs(b).astype(np.float64)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("hidden", [256, 100, 24])
def test_tiled_fp64_preactivation_keeps_every_bit(gpu, hidden):
    """The fp64 pre-activation of `delta` takes the column-sliced work-item route on graphs beyond the caches
    (k_rows_tiled_f64).  Forced on a small hub-heavy graph (tiled_min_bytes = 0) it must give the bits of the row
    kernel: same entry-ordered fma chains, same ordered segment sums -- checked through the `delta` matrix."""
    from linkteller_amd import _lib, engine, graph, synth
    a_hat = graph.first_order_gcn(synth.powerlaw_graph(1100, 7000, seed=11))
    assert np.diff(a_hat.indptr).max() > 300
    n = a_hat.shape[0]
    x = synth.gaussian_features(n, 80, seed=2)
    w = synth.gcn_weights(80, hidden, 2, seed=3)
    _lib.set_tuning("aggregate_first", 0)      # (these shapes would take the aggregate-first route, which has no S1d SpMM)
    try:
        base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu)).enable_fp64()
        assert base.fp64_route() == 0
        rng = np.random.RandomState(2)
        probes, obs = rng.choice(n, 60, replace=False), rng.choice(n, 200, replace=False)
        rows = base.influence_rows(probes, obs, 1e-4, "delta").cpu().numpy()
        _lib.set_tuning("tiled_min_bytes", 0)
        base.refresh()
        tiled = base.influence_rows(probes, obs, 1e-4, "delta").cpu().numpy()
    finally:
        _lib.set_tuning("tiled_min_bytes", None)
        _lib.set_tuning("aggregate_first", None)
    assert rows.max() > 0 and np.array_equal(rows, tiled)


@pytest.mark.gpu
def test_delta_without_the_fp64_preactivation(gpu, influence_golden):
    """`lt_influence_rows(mode = LT_MODE_DELTA)` on a baseline that never had `lt_baseline_enable_fp64` called (a C-ABI
    caller may skip it; the Python engine never does): the kink test then reads the fp32 pre-activation.  Exact zeros
    off the 2-hop set, and the scores within the fp32 forward's own noise class of the fp64 reference (the propagation
    itself is exact; only hidden units within rounding of a ReLU kink can be classified differently)."""
    from test_gpu_parity import _setup
    g = influence_golden
    for key in ("pl600", "er300"):
        if key + ".ref64.influence_val" not in g:
            continue
        args, base = _setup(g, key, gpu)
        nodes = g[key + ".ref32.test_nodes"]
        ref64 = g[key + ".ref64.influence_val"]
        base._fp64 = True                      # keep influence_rows from enabling the fp64 buffers: Z1d stays NULL
        got = base.influence_rows(nodes, nodes, args["influence"], "delta").cpu().numpy().astype(np.float64)
        base._fp64 = False
        exact = base.influence_rows(nodes, nodes, args["influence"], "delta").cpu().numpy().astype(np.float64)
        scale = ref64.max()
        print(f"{key}: |delta(fp32 Z1) - ref64| max {np.abs(got - ref64).max():.3e}, |delta(fp64 Z1) - ref64| max "
              f"{np.abs(exact - ref64).max():.3e}, scale {scale:.3e}")
        assert np.all(got[ref64 == 0] == 0)
        assert np.abs(got - ref64).max() <= 1e-3 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["er", "powerlaw"])
def test_a_step_is_hipgraph_capturable(gpu, family):
    """include/linkteller_hip.h promises launch functions that only enqueue -- no host synchronisation, no allocation,
    no stream or event creation -- so a whole step (lt_baseline_refresh + lt_influence_rows, any mode; on a graph with
    hub rows FULL forks onto the baseline's side stream and joins it back) can be captured into a hipGraph.  Captured
    once, replayed after the output was cleared and after the weights changed in place: the bits of the eager calls."""
    from linkteller_amd import engine, graph, synth
    gen = synth.powerlaw_graph if family == "powerlaw" else synth.erdos_renyi_graph
    a_hat = graph.first_order_gcn(gen(700, 4000, seed=5))
    n = a_hat.shape[0]
    x = torch.from_numpy(synth.gaussian_features(n, 96, seed=2)).to(gpu)
    w = synth.gcn_weights(96, 256, 2, seed=3)
    p = _params(w, gpu)
    base = engine.Baseline(graph.HipGraph(a_hat), x, *p)
    rng = np.random.RandomState(1)
    nodes = torch.from_numpy(rng.choice(n, 120, replace=False).astype(np.int32)).to(gpu)
    for mode in ("full", "sparse", "delta"):
        out = torch.empty((120, 120), dtype=torch.float32, device=gpu)

        def step():
            base.refresh()
            base.influence_rows(nodes, nodes, 1e-4, mode, out=out)
        step()
        eager = out.clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            step()                      # workspaces and the fp64 buffers exist before the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager), (family, mode)
        # the graph reads the borrowed tensors: an in-place weight update shows in the next replay
        with torch.no_grad():
            p[0].mul_(1.5)
        g.replay()
        torch.cuda.synchronize()
        replayed = out.clone()
        step()
        torch.cuda.synchronize()
        assert torch.equal(replayed, out) and not torch.equal(replayed, eager), (family, mode)
        with torch.no_grad():
            p[0].div_(1.5)
        del g


@pytest.mark.gpu
def test_gcn3_step_is_hipgraph_capturable(gpu):
    """The 3-layer primitive too: since the level-2 GEMM reads its row count from the device, lt_influence3_rows has no
    host synchronisation left and a refresh + rows step replays from a captured hipGraph with the eager bits."""
    from linkteller_amd import engine, graph, synth
    a_hat = graph.first_order_gcn(synth.powerlaw_graph(500, 2500, seed=9))
    n, f, h1, h2, c = a_hat.shape[0], 40, 64, 32, 3
    rng = np.random.RandomState(3)

    def u(shape, fan):
        s = 1.0 / np.sqrt(fan)
        return torch.from_numpy(rng.uniform(-s, s, size=shape).astype(np.float32)).to(gpu)
    p = [u((f, h1), h1), u((h1,), h1), u((h1, h2), h2), u((h2,), h2), u((h2, c), c), u((c,), c)]
    base = engine.Baseline3(graph.HipGraph(a_hat), torch.from_numpy(synth.gaussian_features(n, f, seed=1)).to(gpu), *p)
    nodes = torch.from_numpy(rng.choice(n, 40, replace=False).astype(np.int32)).to(gpu)
    holder = {}

    def step():
        base.refresh()
        holder["out"] = base.influence_rows(nodes, nodes, 1e-4)
    step()
    eager = holder["out"].clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    captured = holder["out"]
    captured.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert eager.max() > 0 and torch.equal(captured, eager)
