"""`balanced-full` shape at twitch-RU size (1024 probes x every node observed): k_delta_probe_finish against the
item kernels.  python tools/balanced_full_time.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import time, numpy as np, torch
from linkteller_amd import _lib, engine, graph, synth
adj, x, w = synth.twitch_like_problem("twitch-RU", hidden=256, n_classes=2, seed=0)
a_hat = graph.first_order_gcn(adj); n = adj.shape[0]; dev = torch.device("cuda:0")
base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(dev), *[torch.from_numpy(w[k]).to(dev) for k in ("W1","b1","W2","b2")]).enable_fp64()
pr = torch.from_numpy(np.random.RandomState(5).choice(n, 1024, replace=False).astype(np.int32)).to(dev); ev = torch.arange(n, dtype=torch.int32, device=dev)
out = torch.empty((1024, n), device=dev)
for knob in (1, 0, 1, 0):
    _lib.set_tuning("delta_fused", knob)
    for _ in range(3): base.refresh("delta"); base.influence_rows(pr, ev, 1e-4, "delta", out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): base.refresh("delta"); base.influence_rows(pr, ev, 1e-4, "delta", out=out)
    torch.cuda.synchronize(); print("delta_fused", knob, "1024 probes x all", n, "observed:", round((time.perf_counter() - t) / 20 * 1e3, 4), "ms")
