"""One mode of the 2-layer build at twitch size for rocprofv3: python tools/host_lab/mode_trace.py <mode> <n_probe> <n_obs | all> [powerlaw]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from linkteller_amd import engine, graph, synth
dev = torch.device("cuda:0")
n, f, h = 4385, 3170, 256
mode, npb = sys.argv[1], int(sys.argv[2])
pl = len(sys.argv) > 4 and sys.argv[4] == "powerlaw"
adj = synth.powerlaw_graph(n, 37304, seed=42) if pl else synth.erdos_renyi_graph(n, 37304, seed=42)
hg = graph.HipGraph(graph.first_order_gcn(adj))
x = torch.from_numpy(synth.twitch_like_features(n, f, seed=1)).to(dev)
w = synth.gcn_weights(f, h, 2, seed=42)
base = engine.Baseline(hg, x, *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
rs = np.random.RandomState(42)
probes = torch.from_numpy(rs.choice(n, npb, replace=False).astype(np.int32)).to(dev)
obs = torch.arange(n, dtype=torch.int32, device=dev) if sys.argv[3] == "all" else probes[: int(sys.argv[3])].contiguous() if int(sys.argv[3]) <= npb else torch.from_numpy(rs.choice(n, int(sys.argv[3]), replace=False).astype(np.int32)).to(dev)
out = torch.empty((npb, obs.numel()), dtype=torch.float32, device=dev)
for _ in range(3):
    base.refresh(mode); base.influence_rows(probes, obs, 1e-4, mode, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    base.refresh(mode); base.influence_rows(probes, obs, 1e-4, mode, out=out)
torch.cuda.synchronize()
print(mode, npb, "x", obs.numel(), round((time.perf_counter() - t) / 20 * 1e3, 4), "ms per step")
