"""WideBaseline (hidden width 512, 2 classes; and 256 wide with 16 classes) at twitch size, `delta` / `sparse`, for rocprofv3.
python tools/host_lab/wide_trace.py <h> <c> <mode>"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from linkteller_amd import engine, graph, synth
dev = torch.device("cuda:0")
n, f = 4385, 3170
h, c, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
hg = graph.HipGraph(graph.first_order_gcn(synth.erdos_renyi_graph(n, 37304, seed=42)))
x = torch.from_numpy(synth.twitch_like_features(n, f, seed=1)).to(dev)
w = synth.gcn_weights(f, h, c, seed=42)
base = engine.baseline_for(hg, x, *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
print(type(base).__name__)
nodes = torch.from_numpy(np.random.RandomState(42).choice(n, 500, replace=False).astype(np.int32)).to(dev)
for _ in range(3):
    base.refresh(mode); base.influence_rows(nodes, nodes, 1e-4, mode)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    base.refresh(mode); base.influence_rows(nodes, nodes, 1e-4, mode)
torch.cuda.synchronize()
print(h, c, mode, round((time.perf_counter() - t) / 20 * 1e3, 4), "ms per step")
