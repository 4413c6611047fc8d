"""What one rank of N does per step at twitch-RU size (n_test = 500 -> ceil(500 / N) probes x 500 observed, baseline
replicated): eager calls vs a captured hipGraph -- is the step GPU-bound or bound by the host's launch rate?
python tools/shard_step_time.py [mode] [n_test]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linkteller_amd import engine, graph, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
n_test = int(sys.argv[2]) if len(sys.argv) > 2 else 500
adj, x_np, w = synth.twitch_like_problem("twitch-RU", hidden=256)
a_hat = graph.first_order_gcn(adj)
dev = torch.device("cuda:0")
base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x_np).to(dev), *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
np.random.seed(42)
nodes = np.random.choice(np.arange(adj.shape[0]), n_test, replace=False).astype(np.int32)
obs = torch.from_numpy(nodes).to(dev)
for N in (1, 2, 4, 8):
    per = (n_test + N - 1) // N
    probes = torch.from_numpy(nodes[:per]).to(dev)
    out = torch.empty((per, n_test), dtype=torch.float32, device=dev)
    def step():
        base.refresh(mode)
        base.influence_rows(probes, obs, 1e-4, mode, out=out)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200): step()
    host_only = (time.perf_counter() - t0) / 200            # time for the host to ISSUE 200 steps (no sync)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s): step()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 200
    print(f"{mode} N={N}: {per} probes/rank  eager {eager*1e6:.1f} us/step  (host issues a step in {host_only*1e6:.1f} us)  hipGraph replay {gr*1e6:.1f} us/step  -> speed-up over N=1 eager/graph")
    del g
