"""Randomised route cross-check on mid-size graphs (R-MAT scale 12-16, power-law up to 20 K nodes): every route of the item
modes -- pair marks, short-side hub search, bitmap per probe / for big probes only / none, chunking, tiled layers -- must give
the bits of the plain route, `sparse` the bits of `full` (a few probes), `delta` the fp64 oracle on one probe row.
python tools/fuzz_routes.py [cases] [seed]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linkteller_amd import _lib, engine, graph, synth
from oracle import linkteller_oracle as O

KNOBS = ("pair_marks", "hub_short_side", "bits_max_bytes", "item_bits", "chunk_budget_bytes", "tiled_min_bytes")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
for it in range(cases):
    if rng.rand() < 0.6:
        scale = int(rng.choice([12, 13, 14, 15, 16]))
        adj = synth.rmat_graph(scale, int((1 << scale) * rng.uniform(6, 20)), seed=int(rng.randint(1 << 30)))
        kind = f"rmat{scale}"
    else:
        n0 = int(rng.choice([3000, 8000, 20000]))
        adj = synth.powerlaw_graph(n0, int(n0 * rng.uniform(4, 30)), seed=int(rng.randint(1 << 30)), exponent=rng.uniform(1.6, 2.4))
        kind = f"pl{n0}"
    a_hat = graph.first_order_gcn(adj)
    n = a_hat.shape[0]
    deg = np.diff(a_hat.indptr)
    f, h, c = int(rng.choice([16, 64])), int(rng.choice([32, 100, 256])), int(rng.choice([1, 2, 5, 8]))
    x = synth.gaussian_features(n, f, seed=it)
    w = synth.gcn_weights(f, h, c, seed=it + 1)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(dev), *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
    hubs = np.argsort(-deg)[:40]
    n_probe, n_obs = int(rng.choice([33, 150, 400])), int(rng.choice([64, 500, 2000]))
    probes = np.concatenate([rng.choice(hubs, 6, replace=False), rng.choice(n, n_probe, replace=False)])
    obs = np.concatenate([rng.choice(hubs, 10, replace=False), rng.choice(n, min(n_obs, n), replace=False)])
    for k in KNOBS:
        _lib.set_tuning(k, None)
    _lib.set_tuning("pair_marks", -1); _lib.set_tuning("hub_short_side", 0)
    ref = {m: base.influence_rows(probes, obs, 1e-4, m) for m in ("sparse", "delta")}
    full = base.influence_rows(probes[:12], obs, 1e-4, "full")
    assert torch.equal(full, ref["sparse"][:12]), (it, kind, "full != sparse")
    for trial in range(4):
        knobs = {"pair_marks": int(rng.choice([-1, 0])), "hub_short_side": int(rng.choice([0, 1])),
                 "bits_max_bytes": int(rng.choice([1, 1 << 27])), "item_bits": int(rng.choice([0, 1, 1])),
                 "chunk_budget_bytes": int(rng.choice([1 << 22, 1 << 30])), "tiled_min_bytes": int(rng.choice([0, 1 << 25]))}
        for k, v in knobs.items():
            _lib.set_tuning(k, v)
        base.refresh()
        for m in ("sparse", "delta"):
            got = base.influence_rows(probes, obs, 1e-4, m)
            assert torch.equal(got, ref[m]), (it, kind, m, knobs)
    for k in KNOBS:
        _lib.set_tuning(k, None)
    # fp64 oracle on one random (non-hub) probe row
    v = int(probes[-1])
    with torch.no_grad():
        P64 = {k: torch.from_numpy(w[k]).double() for k in ("W1", "b1", "W2", "b2")}
        gm = O.get_gradient_eps_mat(torch.from_numpy(x).double(), O.to_torch_sparse(a_hat).double(), P64, v, 1e-4)
        r64 = gm[torch.as_tensor(obs)].norm(dim=1).numpy()
    err = np.abs(ref["delta"][-1].cpu().numpy() - r64).max() / max(r64.max(), 1e-6)
    assert err <= 1e-5, (it, kind, err)
    print(f"ok case {it}: {kind} n={n} nnz={a_hat.nnz} maxdeg={int(deg.max())} H={h} C={c} probes={len(probes)} obs={len(obs)} "
          f"nonzero {int((ref['delta'] > 0).sum())}/{ref['delta'].numel()} delta err {err:.1e}")
print("all", cases, "cases ok")
