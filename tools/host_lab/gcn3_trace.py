"""GCN3 (3170-256-64-2) at n_test = 500 on the headline graph, `delta` / `sparse`, 30 steps each for rocprofv3 --kernel-trace --stats.
python tools/host_lab/gcn3_trace.py <mode>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from linkteller_amd import engine, graph, synth
dev = torch.device("cuda:0")
n, f, h, h2, c = 4385, 3170, 256, 64, 2
hg = graph.HipGraph(graph.first_order_gcn(synth.erdos_renyi_graph(n, 37304, seed=42)))
x = torch.from_numpy(synth.twitch_like_features(n, f, seed=1)).to(dev)
rs = np.random.RandomState(7)
u = lambda shape, fo: torch.from_numpy(rs.uniform(-1 / np.sqrt(fo), 1 / np.sqrt(fo), size=shape).astype(np.float32)).to(dev)
b3 = engine.Baseline3(hg, x, u((f, h), h), u((h,), h), u((h, h2), h2), u((h2,), h2), u((h2, c), c), u((c,), c))
nodes = torch.from_numpy(np.random.RandomState(42).choice(n, 500, replace=False).astype(np.int32)).to(dev)
out = torch.empty((500, 500), dtype=torch.float32, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "delta"
import time
for _ in range(5):
    b3.refresh(); b3.influence_rows(nodes, nodes, 1e-4, mode, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(30):
    b3.refresh(); b3.influence_rows(nodes, nodes, 1e-4, mode, out=out)
torch.cuda.synchronize()
print(mode, round((time.perf_counter() - t) / 30 * 1e3, 4), "ms per step")
