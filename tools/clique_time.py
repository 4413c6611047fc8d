"""The record route against the item kernels where its long-position path dominates: a sparse graph with one k-clique, the clique's
members among the probes and all of it observed.  python tools/clique_time.py [clique size ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, scipy.sparse as sp, torch
from linkteller_amd import _lib, engine, graph, synth
dev = torch.device("cuda:0")
n, f, h, c = 2600, 64, 128, 2
for size in [int(v) for v in sys.argv[1:]] or [20, 40, 60]:
    a = synth.erdos_renyi_graph(n, 9000, seed=21).tolil()
    big = np.arange(500, 500 + size)
    for u in big:
        for v_ in big:
            if u != v_:
                a[u, v_] = 1
    a_hat = graph.first_order_gcn(sp.csr_matrix(a))
    x = synth.twitch_like_features(n, f, seed=6, density=0.05)
    w = synth.gcn_weights(f, h, c, seed=8)
    _lib.set_tuning("aggregate_first", 0)          # (small shapes would go aggregate-first: not the record route's case)
    base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(dev), *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")]).enable_fp64()
    _lib.set_tuning("aggregate_first", None)
    rng = np.random.RandomState(3)
    probes = torch.from_numpy(np.concatenate([big[:8], rng.choice(n, 250, replace=False)]).astype(np.int32)).to(dev)
    observe = torch.from_numpy(np.concatenate([big, rng.choice(n, 250, replace=False)]).astype(np.int32)).to(dev)
    out = torch.empty((len(probes), len(observe)), device=dev)
    res = {}
    for knob in (1, 0, 1, 0):
        _lib.set_tuning("delta_fused", knob)
        base.influence_rows(probes, observe, 1e-4, "delta", out=out); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            base.refresh("delta"); base.influence_rows(probes, observe, 1e-4, "delta", out=out)
        torch.cuda.synchronize()
        res[knob] = (time.perf_counter() - t0) / 50 * 1e6
    _lib.set_tuning("delta_fused", None)
    print(f"clique of {size}: record route {res[1]:.1f} us per step, item kernels {res[0]:.1f} us")
