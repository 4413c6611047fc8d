import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from linkteller_amd import engine, graph, synth
adj, x_np, w = synth.twitch_like_problem("twitch-RU", hidden=256)
a_hat = graph.first_order_gcn(adj)
dev = torch.device('cuda:0')
base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x_np).to(dev), *[torch.from_numpy(w[k]).to(dev) for k in ("W1","b1","W2","b2")])
np.random.seed(42); nodes = np.random.choice(np.arange(adj.shape[0]), 500, replace=False).astype(np.int32)
probes = torch.from_numpy(nodes).to(dev); out = torch.empty((500, 500), dtype=torch.float32, device=dev)
import os
WHICH = os.environ.get("WHICH", "both")
MODE = os.environ.get("MODE", "full")
def step():
    if WHICH in ("both", "refresh"): base.refresh()
    if WHICH in ("both", "rows"): base.influence_rows(probes, probes, 1e-4, MODE, out=out)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50
ref = out.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    step()
out.zero_()
g.replay(); torch.cuda.synchronize()
print('graph result equal:', bool(torch.equal(out, ref)))
t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 50
print(f'eager {eager*1e3:.4f} ms/step   graph {gr*1e3:.4f} ms/step')
