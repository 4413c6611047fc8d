import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from linkteller_amd import engine, graph, synth
dev = torch.device("cuda:0")
n, f, h = 4385, 3170, 256
for kind in ("er", "pl"):
    adj = synth.erdos_renyi_graph(n, 37304, seed=42) if kind == "er" else synth.powerlaw_graph(n, 37304, seed=42)
    hg = graph.HipGraph(graph.first_order_gcn(adj))
    x = torch.from_numpy(synth.twitch_like_features(n, f, seed=1)).to(dev)
    w = synth.gcn_weights(f, h, 2, seed=42)
    base = engine.Baseline(hg, x, *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
    nodes = np.random.RandomState(42).choice(n, 500, replace=False)
    m = base.influence_rows(nodes, nodes, 1e-4, "delta").cpu().numpy()
    print(kind, "nonzero fraction", round(float((m > 0).mean()), 4), "rows' nonzeros: median", int(np.median((m > 0).sum(1))), "max", int((m > 0).sum(1).max()))
