"""GPU tests added in round 4: the N > 1 path executed by RCCL itself on one GPU (world size 1, every collective forced),
an observed hub repeated in observe_nodes, the last-arriver reduction of k_s1d_feature_rows under stress, a wide model on
two ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def _s1d_baseline(a_hat, x, w, gpu):
    """A `delta` baseline on one of the routes that form the product rows S1d (0 / 1): small shapes would otherwise go
    aggregate-first (route 2), whose pre-activation is formed on demand -- not the record route's case."""
    from linkteller_amd import _lib, engine, graph
    _lib.set_tuning("aggregate_first", 0)
    try:
        base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu)).enable_fp64()
    finally:
        _lib.set_tuning("aggregate_first", None)
    assert base.fp64_route() in (0, 1)
    return base


def _item_stage_a_launches(fn):
    """Launches of the item kernels' stage A (k_item_stageA*) while fn runs: 0 <=> a `delta` call took the record route."""
    import ctypes as C
    from linkteller_amd import _lib
    h = _lib.lib()
    h.lt_profile_reset()
    h.lt_profile_enable(1 << _lib.KERNEL_IDS["item_stageA"])
    try:
        fn()
        torch.cuda.synchronize()
        tot, cnt = C.c_double(), C.c_int64()
        _lib.check(h.lt_profile_summary(_lib.KERNEL_IDS["item_stageA"], C.byref(tot), C.byref(cnt)))
    finally:
        h.lt_profile_enable(0)
        h.lt_profile_reset()
    return cnt.value


def _params(w, dev):
    return [torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")]


def _bench(args, env=None, timeout=1200):
    e = dict(os.environ, PYTHONPATH=REPO, **(env or {}))
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("mode,extra", [("delta", {}), ("sparse", {"LT_SHARD_BASELINE": "1"}),
                                        ("delta", {"LT_FEATURE_DELTA": "0", "LT_AGGREGATE_FIRST": "0", "LT_SHARD_BASELINE": "1"})])
def test_bench_collectives_execute_under_rccl_at_world_size_1(gpu, tmp_path, mode, extra):
    """LT_FORCE_COLLECTIVES=1: bench.py creates the ``nccl`` process group (device_id=) at world size 1 and issues every
    collective of the N > 1 path -- the async all_gather_into_tensor of row slabs pipelined across steps, the sharded
    refresh's all-gather of X W1 (fp32 for `sparse`, fp64 for `delta` on dense routing), the policy's all-reduce, the
    timing all-reduce, barriers -- on device tensors, so RCCL itself runs them.  The process must have librccl mapped, the
    line must carry the three strong-scaling workloads with an event-timed collective, and the matrix of the last step
    must equal the plain one-process run bit for bit."""
    common = ["--steps", "3", "--warmup", "1", "--blocks", "2", "--no-cpu-baseline", "--no-extras", "--no-pmc", "--mode", mode,
              "--n-test", "120", "--spmm-scale", "14"]
    plain = _bench(common, {"LT_BENCH_DUMP": str(tmp_path / "plain.npy"), **extra})
    # (LT_SHARD_PROBES=1: the all-gather of row slabs on every step -- under the default `auto` a one-rank "group" would find
    # that building all rows locally is faster and issue none; tests/test_gpu_round6.py covers that policy)
    forced = _bench(common, {"LT_FORCE_COLLECTIVES": "1", "LT_SHARD_PROBES": "1", "LT_BENCH_DUMP": str(tmp_path / "forced.npy"), **extra})
    assert plain["collectives"]["backend"] is None and not plain["collectives"]["forced_at_world_size_1"]
    col = forced["collectives"]
    assert col["backend"] == "nccl" and col["forced_at_world_size_1"] and col["librccl_mapped"], col
    assert col["communicator_segment_mapped"] and not plain["collectives"]["communicator_segment_mapped"], (col, plain["collectives"])
    assert forced["n_gpus"] == 1 and forced["config"]["collective_bytes_per_step"] >= 120 * 120 * 4
    if extra.get("LT_SHARD_BASELINE") == "1":
        assert "sharded" in forced["config"]["baseline_XW1"], forced["config"]
    sw = forced["scaling_workloads"]
    for key, n_t in (("configs[1]", 120), ("configs[2]", 2000), ("configs[4]", 4096)):
        assert sw[key]["probes_per_rank"] == n_t and sw[key]["ms_per_step"] > 0 and sw[key]["collective_us"] > 0, sw[key]
    assert "scaling_workloads" not in plain           # (--no-extras at one rank: nothing to shard, the leg is skipped)
    a, b = np.load(tmp_path / "plain.npy"), np.load(tmp_path / "forced.npy")
    assert a.shape == (120, 120) and np.array_equal(a, b)


def test_attacker_through_rccl_at_world_size_1(gpu, tmp_path):
    """The product's own N > 1 path (main.init_distributed -> Attacker.influence_matrix -> dist.all_gather_rows, and the
    sharding policy's collectives) under RCCL at world size 1, in a child process: same matrix as without a group."""
    code = r'''
import os, sys, types, argparse
import numpy as np, torch
sys.path.insert(0, os.environ["LT_REPO"])
from linkteller_amd import graph, synth, main as lt_main, dist as lt_dist
from linkteller_amd.attacker import Attacker
from linkteller_amd.gcn import GCN
dev = torch.device("cuda:0")
n, f, h = 700, 300, 64
adj = synth.powerlaw_graph(n, 3500, seed=3)
x = torch.from_numpy(synth.gaussian_features(n, f, seed=4)).to(dev)
a_hat = graph.first_order_gcn(adj)
adj_t = graph.sparse_mx_to_torch_sparse_tensor(a_hat).to(dev)
w = synth.gcn_weights(f, h, 2, seed=5)
model = GCN(f, h, 2, 0.5)
model.load_state_dict({"gc1.weight": torch.from_numpy(w["W1"]), "gc1.bias": torch.from_numpy(w["b1"]),
                       "gc2.weight": torch.from_numpy(w["W2"]), "gc2.bias": torch.from_numpy(w["b2"])})
model.to(dev).eval()
wk = types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=adj.tocsr(), n_nodes=n)
def run():
    out = {}
    for mode in ("delta", "sparse"):
        args = argparse.Namespace(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=90, sample_seed=42, influence=1e-4,
                                  mode="vanilla-clean", attack_mode="efficient", influence_mode=mode)
        atk = Attacker(args, model, wk)
        atk.prepare_test_data()
        out[mode] = atk.influence_matrix()
    return out
plain = run()
os.environ["LT_FORCE_COLLECTIVES"] = "1"
os.environ["LT_SHARD_PROBES"] = "1"
os.environ["LT_SHARD_BASELINE"] = "1"
assert lt_main.init_distributed() is True
assert lt_dist.force_collectives() and lt_dist.collectives_on() and lt_dist.world() == (0, 1)
forced = run()
import torch.distributed as dist
assert dist.get_backend() == "nccl"
dist.barrier(); dist.destroy_process_group()
maps = open("/proc/self/maps").read()
assert "librccl" in maps, "RCCL not mapped"
for m in plain:
    assert plain[m].shape == (90, 90) and np.array_equal(plain[m], forced[m]), m
print("OK")
'''
    e = dict(os.environ, LT_REPO=REPO, PYTHONPATH=REPO)
    for k in ("LT_FORCE_COLLECTIVES", "RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    # (RCCL's version banner leaves C stdio at exit, behind the script's last print)
    assert r.returncode == 0 and "\nOK\n" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("n_leaves,repeat", [(400, 2), (400, 5), (150, 3)])
def test_observed_hub_repeated_in_observe_nodes(gpu, n_leaves, repeat):
    """observe_nodes may repeat a node (the reference indexes grad[test_nodes[j]], attacker.py:229).  A star graph has ONE hub
    row (> LT_ROW_SEG entries), so stage B launches hub blocks for min(n_obs, 1) observed hub -- and the hub observed
    `repeat` times fills `repeat` slots of hub_obs: every one of those columns must be written (they were left as
    torch.empty garbage before the blocks learnt to serve slot + k * hub_cap), equal to each other and to the oracle."""
    from test_gpu_parity import _oracle_matrix
    from linkteller_amd import engine, graph, synth
    n = n_leaves + 1
    rows = np.concatenate([np.zeros(n_leaves, dtype=np.int64), np.arange(1, n)])
    cols = np.concatenate([np.arange(1, n), np.zeros(n_leaves, dtype=np.int64)])
    adj = sp.csr_matrix((np.ones(2 * n_leaves, dtype=np.float32), (rows, cols)), shape=(n, n))
    a_hat = graph.first_order_gcn(adj)
    x = synth.twitch_like_features(n, 64, seed=3, density=0.05)
    w = synth.gcn_weights(64, 32, 2, seed=4)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu))
    rng = np.random.RandomState(1)
    leaves = rng.choice(np.arange(1, n), 20, replace=False)
    observe = np.concatenate([[0], leaves[:7], [0] * (repeat - 1), leaves[7:12]])
    probes = np.concatenate([[0], leaves[:10], leaves[15:]])
    ref64 = _oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float64)
    e32 = np.abs(_oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float32) - ref64).max()
    hub_cols = np.flatnonzero(observe == 0)
    assert len(hub_cols) == repeat
    for mode in ("delta", "sparse", "full"):
        # poison the output first: an unwritten column keeps the poison
        out = torch.full((len(probes), len(observe)), float("nan"), dtype=torch.float32, device=gpu)
        got = base.influence_rows(probes, observe, 1e-4, mode, out=out).cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), (mode, np.argwhere(~np.isfinite(got))[:5])
        for j in hub_cols[1:]:
            assert np.array_equal(got[:, j], got[:, hub_cols[0]]), (mode, j)
        tol = 1e-5 * ref64.max() if mode == "delta" else max(2.0 * e32, 2e-3 * ref64.max())
        assert np.abs(got - ref64).max() <= tol, (mode, np.abs(got - ref64).max())
    assert np.array_equal(base.influence_rows(probes, observe, 1e-4, "sparse").cpu().numpy(),
                          base.influence_rows(probes, observe, 1e-4, "full").cpu().numpy())


def test_last_arriver_reduction_under_stress(gpu):
    """k_s1d_feature_rows' first blocks form the K slices of cref = m W1 and the block that draws the last ticket adds them
    (lt_fp64.hip: sc1 stores -> per-wave s_waitcnt vmcnt(0) -> barrier -> agent-scope ticket -> sc1 loads).  A slice that
    the last block reads before it has landed would be the PREVIOUS refresh's -- so W1 alternates between two weight sets
    from refresh to refresh (a stale slice then belongs to the other set and moves every score), a side stream keeps the
    memory system busy with copies of uneven size, and 20 000 refreshes must each reproduce the bits of their weight set."""
    from linkteller_amd import _lib, engine, graph, synth
    n, f, h = 1200, 3170, 256            # twitch-shaped rows: F = 3170 -> 50 slab blocks in front of the row blocks
    a_hat = graph.first_order_gcn(synth.powerlaw_graph(n, 6000, seed=2))
    x = torch.from_numpy(synth.twitch_like_features(n, f, seed=4, density=0.006)).to(gpu)
    wa, wb = synth.gcn_weights(f, h, 2, seed=5), synth.gcn_weights(f, h, 2, seed=6)
    W = [torch.from_numpy(wa["W1"]).to(gpu), torch.from_numpy(wb["W1"]).to(gpu)]
    p = _params(wa, gpu)
    base = engine.Baseline(graph.HipGraph(a_hat), x, *p).enable_fp64()
    assert base.fp64_route() == 1
    rng = np.random.RandomState(0)
    probes = torch.from_numpy(rng.choice(n, 6, replace=False).astype(np.int32)).to(gpu)
    observe = torch.from_numpy(rng.choice(n, 64, replace=False).astype(np.int32)).to(gpu)

    def rows():
        base.refresh("delta")
        return base.influence_rows(probes, observe, 1e-4, "delta")
    want = []
    for k in (0, 1):                      # the expected bits of each weight set, formed on a quiet device ...
        p[0].copy_(W[k])
        torch.cuda.synchronize()
        r = rows().clone()
        torch.cuda.synchronize()
        # ... and cross-checked against the route without the in-launch reduction (cref by a launch of its own)
        _lib.set_tuning("defer_cref", 0)
        try:
            early = rows().clone()
        finally:
            _lib.set_tuning("defer_cref", None)
        assert float((early - r).abs().max()) <= 1e-6 * float(r.max())
        want.append(r)
    assert float((want[0] - want[1]).abs().max()) > 1e-3 * float(want[0].max())      # the two sets are told apart
    side = torch.cuda.Stream()
    junk, junk2 = (torch.empty(64 << 20, dtype=torch.uint8, device=gpu) for _ in range(2))
    bad = torch.zeros((), dtype=torch.int64, device=gpu)
    iters = 20000
    for it in range(iters):
        k = it & 1
        if it % 7 == 0:                   # uneven background traffic: copies of varying size on another stream
            with torch.cuda.stream(side):
                sz = (1 + (it * 2654435761) % 63) << 20
                junk2[:sz].copy_(junk[:sz])
        p[0].copy_(W[k])
        bad += (rows() != want[k]).any()
    torch.cuda.synchronize()
    assert int(bad.item()) == 0, f"{int(bad.item())} of {iters} refreshes read a stale slice of cref"


def _wide_rank(rank, ws, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    import argparse
    import types
    from linkteller_amd import graph, synth
    from linkteller_amd.attacker import Attacker
    from linkteller_amd.gcn import GCN
    dev = torch.device("cuda:0")
    n, f, h, c = 300, 80, 320, 12          # hidden > 256 and > 8 classes: outside one pass of the fused kernels
    adj = synth.powerlaw_graph(n, 1400, seed=3)
    x = torch.from_numpy(synth.gaussian_features(n, f, seed=4)).to(dev)
    adj_t = graph.sparse_mx_to_torch_sparse_tensor(graph.first_order_gcn(adj)).to(dev)
    w = synth.gcn_weights(f, h, c, seed=5)
    model = GCN(f, h, c, 0.5)
    model.load_state_dict({"gc1.weight": torch.from_numpy(w["W1"]), "gc1.bias": torch.from_numpy(w["b1"]),
                           "gc2.weight": torch.from_numpy(w["W2"]), "gc2.bias": torch.from_numpy(w["b2"])})
    model.to(dev).eval()
    wk = types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=adj.tocsr(), n_nodes=n)
    args = argparse.Namespace(dataset="twitch/ES/RU", sample_type="unbalanced", n_test=24, sample_seed=42, influence=1e-4,
                              mode="vanilla-clean", attack_mode="efficient")          # default mode (delta), default policy (auto)
    atk = Attacker(args, model, wk)
    atk.prepare_test_data()
    m = atk.influence_matrix()
    q.put((rank, m))
    dist.barrier()
    dist.destroy_process_group()


def test_wide_model_on_two_ranks_default_mode(gpu):
    """a model wider than the fused kernels under the defaults (mode `delta`, LT_SHARD_BASELINE=auto) on two
    ranks used to die in the sharding policy's timing loop.  Two ranks on one device (gloo hook): both finish, with the
    same matrix as one process."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_wide_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0].shape == (24, 24) and np.array_equal(res[0], res[1]) and np.isfinite(res[0]).all() and res[0].max() > 0


@pytest.mark.parametrize("key", ["twitch-ES.64.clean", "twitch-RU.500.clean", "twitch-RU.500.lapgraph", "twitch-RU.2000.clean"])
def test_fp32_modes_meet_the_reference_noise_over_the_whole_matrix(gpu, key):
    """``err(build) <= err(reference fp32)``, both against the fp64 evaluation -- on the WHOLE matrix of
    BASELINE configs[0], [1], [3] and on a 200-row sample of configs[2] (round 3 asserted it on 12-row samples, an
    extreme-value ratio that read 1.1 .. 2.06).  The right-hand side is the REFERENCE's own fp32 run (attacker.py:100-108 on
    torch's CPU kernels, one thread), generated in the build container by tests/golden/generate_fp32_noise.py and committed
    (fp32_whole_matrix.npz: the fp32 scores of every probed row); the fp64 side is oracle.RestrictedOracle, live.
    Gates: max |full - ref64| <= 1.10 x max |ref32 - ref64| and the same for the root mean square over the 2-hop support."""
    from conftest import load_golden, noise_gate
    from linkteller_amd import dp, engine, graph, synth
    from oracle import linkteller_oracle as O
    g = load_golden("fp32_whole_matrix.npz")
    workload, n_test, served = key.rsplit(".", 2)
    adj, x, w = synth.twitch_like_problem(workload, hidden=256, n_classes=2, seed=0)
    if served == "lapgraph":
        adj = dp.perturb_adj(adj, "continuous", 5.0, noise_seed=42)
    a_hat = graph.first_order_gcn(adj)
    nodes, rows, ref32 = g[f"{key}.nodes"], g[f"{key}.rows"], g[f"{key}.ref32"].astype(np.float64)
    assert len(nodes) == int(n_test) and np.array_equal(nodes, np.sort(np.random.RandomState(7).choice(adj.shape[0], int(n_test), replace=False)))
    ref64 = O.RestrictedOracle(x, a_hat, w).rows(nodes[rows], nodes, 1e-4)
    support = ref64 != 0
    e32_max, e32_rms = np.abs(ref32 - ref64).max(), np.sqrt(((ref32 - ref64)[support] ** 2).mean())
    # the committed statistics are those of the committed rows against THIS fp64 evaluation (pins the live oracle to the generator's)
    assert abs(e32_max - float(g[f"{key}.e32_max"])) <= 1e-9 and abs(e32_rms - float(g[f"{key}.e32_rms"])) <= 1e-9
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(gpu), *_params(w, gpu))
    res = {m: base.influence_rows(nodes[rows], nodes, 1e-4, m).cpu().numpy().astype(np.float64) for m in ("full", "sparse", "delta")}
    assert np.array_equal(res["full"], res["sparse"])
    assert np.all(res["full"][~support] == 0) and np.all(res["delta"][~support] == 0)
    assert np.abs(res["delta"] - ref64).max() <= 1e-5 * ref64.max()
    d = res["full"] - ref64
    r_max, r_rms = np.abs(d).max() / e32_max, np.sqrt((d[support] ** 2).mean()) / e32_rms
    print(f"{key}: rows {len(rows)} x {n_test}; reference fp32 error max {e32_max:.4e} rms {e32_rms:.4e}; "
          f"full: max ratio {r_max:.4f}, rms ratio {r_rms:.4f}; delta error {np.abs(res['delta'] - ref64).max():.2e}")
    noise_gate(f"whole.{key}.full_rms", r_rms, ceiling=1.10)
    noise_gate(f"whole.{key}.full_max", r_max, ceiling=1.10)


def test_gather_ceiling_entry_point(gpu):
    """lt_spmm_gather_ceiling (measurement support for bench.py's roofline_spmm.gather_ceiling): runs on the graph's own
    work items for both depths, with and without the result stores; argument errors come back as statuses."""
    import ctypes as C
    from linkteller_amd import _lib, engine, graph, synth
    a_hat = graph.first_order_gcn(synth.rmat_graph(13, synth.rmat_draws(13), seed=42))
    hg = graph.HipGraph(a_hat)
    n, h = a_hat.shape[0], 256
    s = torch.randn((n, h), device=gpu)
    L = _lib.lib()
    need = int(L.lt_spmm_gather_ceiling_bytes(hg.handle))
    assert need >= a_hat.shape[0] * 4
    sink = torch.zeros(need, dtype=torch.uint8, device=gpu)
    poison = 0x7FC12345                                              # (a NaN pattern: the stored words are XORs of float bits, any pattern)
    out = torch.full((n, h), poison, dtype=torch.int32, device=gpu).view(torch.float32)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for u in (8, 16):
        _lib.check(L.lt_spmm_gather_ceiling(hg.handle, s.data_ptr(), h, h, u, sink.data_ptr(), need, None, 0, st))
        _lib.check(L.lt_spmm_gather_ceiling(hg.handle, s.data_ptr(), h, h, u, sink.data_ptr(), need, out.data_ptr(), h, st))
    torch.cuda.synchronize()
    assert int(sink.view(torch.int32).ne(0).sum()) > 0            # the XOR words of the gathered rows landed
    short = np.diff(a_hat.indptr) <= 128
    stored = out.view(torch.int32)[torch.from_numpy(short).to(gpu)] != poison
    assert float(stored.float().mean()) > 0.999                     # every short row's slots were stored (an XOR word equal to the poison: ~2^-32)
    assert L.lt_spmm_gather_ceiling(hg.handle, s.data_ptr(), h, h, 12, sink.data_ptr(), need, None, 0, st) == -1
    assert L.lt_spmm_gather_ceiling(hg.handle, s.data_ptr(), h, h, 8, sink.data_ptr(), need - 64, None, 0, st) == -1
    # the real kernel still gives the row kernels' bits next to it (sanity that the shared work items were not disturbed)
    _lib.set_tuning("tiled_min_bytes", 0)
    try:
        tiled = engine.spmm(hg, s)
    finally:
        _lib.set_tuning("tiled_min_bytes", None)
    _lib.set_tuning("tiled_min_bytes", 1 << 60)
    try:
        rows = engine.spmm(hg, s)
    finally:
        _lib.set_tuning("tiled_min_bytes", None)
    assert torch.equal(tiled, rows)


@pytest.mark.parametrize("h,c,kind", [(256, 2, "er"), (100, 3, "er"), (16, 8, "er"), (132, 1, "iso"), (64, 7, "dup")])
def test_delta_fused_probe_blocks_keep_every_bit(gpu, h, c, kind):
    """k_delta_probe_finish (round 4, `delta_fused`): stage A and stage B of a probe in one block, from the probe node's
    incidence record (built with the graph) matched against the observed list by delta_record_block.  Graphs without hub rows; the matrix must equal the three-launch route's (`delta_fused`
    = 0) bit for bit -- widths that need padding and several lane groupings, 1 .. 8 classes, isolated nodes, duplicate probes
    and observed nodes, observe != probes, a single probe, a multi-chunk call, both storage forms of the product rows (twitch-like
    and Gaussian features) -- and stay within 1e-5 of the fp64 oracle."""
    from test_gpu_parity import _oracle_matrix
    from linkteller_amd import _lib, engine, graph, synth
    n, f = 700, 96
    a = synth.erdos_renyi_graph(n, 4200, seed=h + c).tolil()
    if kind == "iso":
        for k in (5, 77, 300):
            a[k, :] = 0
            a[:, k] = 0
    a = sp.csr_matrix(a)
    a.eliminate_zeros()
    a_hat = graph.first_order_gcn(a)
    assert np.diff(a_hat.indptr).max() <= 128                     # no hub rows: the fused route applies
    rng = np.random.RandomState(c)
    w = synth.gcn_weights(f, h, c, seed=3)
    for feats in ("twitch", "gauss"):
        x = synth.twitch_like_features(n, f, seed=2, density=0.05) if feats == "twitch" else synth.gaussian_features(n, f, seed=2)
        base = _s1d_baseline(a_hat, x, w, gpu)
        probes = rng.choice(n, 75, replace=False)
        observe = rng.choice(n, 90, replace=False)
        if kind == "dup":
            probes[3] = probes[9]
            observe[5] = observe[6]
        if kind == "iso":
            probes[0], observe[0] = 5, 77
        calls = [(probes, observe), (probes[:1], observe), (probes, probes)]
        # the record route is what the first call of each pair below takes (aggregate-first, which small wide-layer shapes
        # would otherwise choose, forms the pre-activation on demand: not this route's case)
        assert _item_stage_a_launches(lambda: base.influence_rows(probes, observe, 1e-4, "delta")) == 0
        for pr, ob in calls:
            fused = base.influence_rows(pr, ob, 1e-4, "delta").cpu().numpy()
            _lib.set_tuning("delta_fused", 0)
            try:
                plain = base.influence_rows(pr, ob, 1e-4, "delta").cpu().numpy()
            finally:
                _lib.set_tuning("delta_fused", None)
            assert np.array_equal(fused, plain), (feats, len(pr), np.abs(fused - plain).max())
        ref64 = _oracle_matrix(a_hat, x, w, probes[:12], observe, 1e-4, torch.float64)
        got = base.influence_rows(probes[:12], observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
        assert np.abs(got - ref64).max() <= 1e-5 * ref64.max()
        assert np.all(got[ref64 == 0] == 0)
        # several chunks per call
        _lib.set_tuning("chunk_budget_bytes", 1 << 15)
        try:
            chunked = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy()
        finally:
            _lib.set_tuning("chunk_budget_bytes", None)
        assert np.array_equal(chunked, base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy())


def test_delta_fused_on_a_directed_pattern_with_empty_columns_and_many_observed(gpu):
    """k_delta_probe_finish beyond the symmetric case: a DIRECTED adjacency (rows and columns differ: R_v is the column of v, the
    rows that hold a member are the member's column, `tpos` is the entry's place in its ROW), nodes nobody reads (empty
    columns: no items) and nobody is read by (empty rows), and more observed positions than a thread keeps canonical indices
    for (n_obs > 1024, with repeats).  Bits of the item kernels, and the fp64 oracle on a few rows."""
    from test_gpu_parity import _oracle_matrix
    from linkteller_amd import _lib, engine, graph, synth
    n, f, h, c = 2500, 48, 64, 3
    rng = np.random.RandomState(11)
    rows = rng.randint(0, n, 30000)
    cols = rng.randint(0, n, 30000)
    keep = (rows != cols) & (cols >= 40) & (rows >= 20)            # columns 0..39 stay empty, rows 0..19 too
    a = sp.csr_matrix((rng.uniform(0.05, 0.3, keep.sum()).astype(np.float32), (rows[keep], cols[keep])), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    assert np.diff(a.indptr).max() <= 128 and np.diff(a.tocsc().indptr).max() <= 60
    x = synth.gaussian_features(n, f, seed=2)
    w = synth.gcn_weights(f, h, c, seed=3)
    base = _s1d_baseline(a, x, w, gpu)
    probes = np.concatenate([[0, 5, 39, 40], rng.choice(n, 60, replace=False)])           # 0, 5, 39: empty columns (no items)
    observe = np.concatenate([rng.choice(n, 1400, replace=False), [3, 3, 41, 41, 41]])     # 3: an empty row; repeats
    assert _item_stage_a_launches(lambda: base.influence_rows(probes, observe, 1e-4, "delta")) == 0       # the record route
    fused = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy()
    _lib.set_tuning("delta_fused", 0)
    try:
        plain = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy()
    finally:
        _lib.set_tuning("delta_fused", None)
    assert np.array_equal(fused, plain), np.abs(fused - plain).max()
    assert np.all(fused[:3] == 0) and np.array_equal(fused[:, -1], fused[:, -3]) and fused.max() > 0
    ref64 = _oracle_matrix(a, x, w, probes[3:9], observe, 1e-4, torch.float64)
    assert np.abs(fused[3:9].astype(np.float64) - ref64).max() <= 1e-5 * max(ref64.max(), 1e-9)


def test_delta_fused_with_every_node_observed(gpu):
    """`balanced-full` observes every node (attacker.py:250-284): every touched node of a probe is then a touched position
    (n_obs = 4385 positions per table row, several search trips per thread in delta_record_block, touched positions beyond the
    two a thread of k_delta_probe_finish holds in registers).  Same bits as the item kernels; the call is chunked as the attack
    chunks it."""
    from linkteller_amd import _lib, engine, graph, synth
    adj, x, w = synth.twitch_like_problem("twitch-RU", hidden=256, n_classes=2, seed=0)
    a_hat = graph.first_order_gcn(adj)
    n = adj.shape[0]
    base = _s1d_baseline(a_hat, x, w, gpu)
    probes = np.random.RandomState(5).choice(n, 300, replace=False)
    everyone = np.arange(n)
    assert _item_stage_a_launches(lambda: base.influence_rows(probes, everyone, 1e-4, "delta")) == 0     # the record route
    fused = base.influence_rows(probes, everyone, 1e-4, "delta").cpu().numpy()
    _lib.set_tuning("delta_fused", 0)
    try:
        plain = base.influence_rows(probes, everyone, 1e-4, "delta").cpu().numpy()
    finally:
        _lib.set_tuning("delta_fused", None)
    assert np.array_equal(fused, plain)
    assert (fused > 0).sum() > 10 * len(probes) and np.isfinite(fused).all()


def test_delta_fused_long_lists_and_huge_observed_lists(gpu):
    """The record route beyond its common case: a 14-clique (every pair inside it shares 14 entries: 14 long positions per probe,
    more than the four waves of a block take in their first trip), a star of 110 leaves whose centre's own list has 111 entries
    (two 64-entry trips of the wave that owns it; the row stays under the 128 entries of a hub), every node observed three times
    over (more touched positions than a thread's registers hold, repeated long positions), and an observed list of more than
    65534 positions, which the record route leaves to the item kernels.  Same bits as `delta_fused` = 0, and the fp64 oracle."""
    from test_gpu_parity import _oracle_matrix
    from linkteller_amd import _lib, engine, graph, synth
    n, f, h, c = 900, 80, 96, 3
    a = synth.erdos_renyi_graph(n, 3600, seed=11).tolil()
    clique = np.arange(100, 114)
    for u in clique:
        for v_ in clique:
            if u != v_:
                a[u, v_] = 1
    centre, leaves = 300, np.arange(400, 510)
    for u in leaves:
        a[centre, u] = 1
        a[u, centre] = 1
    a = sp.csr_matrix(a)
    a_hat = graph.first_order_gcn(a)
    assert np.diff(a_hat.indptr).max() <= 128 and np.diff(a_hat.indptr)[centre] >= 111
    x = synth.twitch_like_features(n, f, seed=5, density=0.05)
    w = synth.gcn_weights(f, h, c, seed=4)
    base = _s1d_baseline(a_hat, x, w, gpu)
    probes = np.concatenate([clique[:6], [centre], leaves[:5], np.random.RandomState(2).choice(n, 40, replace=False)])
    everyone3 = np.tile(np.arange(n), 3)

    def both(pr, ob):
        fused = base.influence_rows(pr, ob, 1e-4, "delta").cpu().numpy()
        _lib.set_tuning("delta_fused", 0)
        try:
            plain = base.influence_rows(pr, ob, 1e-4, "delta").cpu().numpy()
        finally:
            _lib.set_tuning("delta_fused", None)
        assert np.array_equal(fused, plain), np.abs(fused - plain).max()
        return fused
    assert _item_stage_a_launches(lambda: base.influence_rows(probes, everyone3, 1e-4, "delta")) == 0    # the record route
    got = both(probes, everyone3)
    assert np.array_equal(got[:, :n], got[:, n:2 * n]) and np.array_equal(got[:, :n], got[:, 2 * n:])
    ref64 = _oracle_matrix(a_hat, x, w, probes[:9], np.arange(n), 1e-4, torch.float64)
    assert np.abs(got[:9, :n].astype(np.float64) - ref64).max() <= 1e-5 * ref64.max()
    assert np.all(got[:9, :n][ref64 == 0] == 0)
    # calls of more probes than the chip holds 4-wave blocks for take 2-wave (> 1280 probes) and 1-wave (> 2560) blocks
    many = np.resize(np.concatenate([probes, np.arange(n)]), 2700)
    obs64 = np.concatenate([clique, [centre], leaves[:9], np.arange(40)])
    for count in (1500, 2700):
        rows = both(many[:count], obs64)
        assert np.array_equal(rows[:len(probes)], got[:, obs64])
    # more observed positions than the table row's 16-bit counts take: the item kernels serve the call
    huge = np.tile(np.arange(n), 73)[:65600]
    assert _item_stage_a_launches(lambda: base.influence_rows(probes[:7], huge, 1e-4, "delta")) > 0       # the item kernels
    wide = both(probes[:7], huge)
    assert np.array_equal(wide[:, :n], got[:7, :n])


def test_delta_fused_on_dense_clusters(gpu):
    """Records of thousands of incidences per node (the cap is 4096): a 38-clique inside a sparse graph (each member: ~ 1 800
    incidences, every pair inside the clique a 38-entry list, 38 long positions per probe: one per wave first, then eight per
    wave side by side) and two 24-cliques sharing 8 nodes (lists of mixed lengths).  Same bits as the item kernels,
    within 1e-5 of the fp64 oracle.  (A 72-clique passes the cap: no records, the item kernels serve both calls.)"""
    from test_gpu_parity import _oracle_matrix
    from linkteller_amd import _lib, engine, graph, synth
    n, f, h, c = 2600, 64, 128, 2
    for size in (38, 72):
        a = synth.erdos_renyi_graph(n, 9000, seed=21).tolil()

        def clique(nodes):
            for u in nodes:
                for v_ in nodes:
                    if u != v_:
                        a[u, v_] = 1
        big = np.arange(500, 500 + size)
        clique(big)
        clique(np.arange(900, 924))
        clique(np.arange(916, 940))
        a_hat = graph.first_order_gcn(sp.csr_matrix(a))
        assert np.diff(a_hat.indptr).max() <= 128
        x = synth.twitch_like_features(n, f, seed=6, density=0.05)
        w = synth.gcn_weights(f, h, c, seed=8)
        base = _s1d_baseline(a_hat, x, w, gpu)
        rng = np.random.RandomState(3)
        probes = np.concatenate([big[:5], [905, 935, 960], rng.choice(n, 30, replace=False)])
        observe = np.concatenate([big, np.arange(900, 940), rng.choice(n, 200, replace=False)])
        launches = _item_stage_a_launches(lambda: base.influence_rows(probes, observe, 1e-4, "delta"))
        assert (launches == 0) == (size == 38), (size, launches)          # 38: the record route; 72: beyond the cap
        fused = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy()
        _lib.set_tuning("delta_fused", 0)
        try:
            plain = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy()
        finally:
            _lib.set_tuning("delta_fused", None)
        assert np.array_equal(fused, plain), (size, np.abs(fused - plain).max())
        ref64 = _oracle_matrix(a_hat, x, w, probes[:8], observe, 1e-4, torch.float64)
        assert np.abs(fused[:8].astype(np.float64) - ref64).max() <= 1e-5 * ref64.max()
        assert np.all(fused[:8][ref64 == 0] == 0)
