"""How long after the last row has landed in pinned host memory does the stream wait return?  The matrix is pre-filled with NaN; the
host spins on cells of the rows the probes' blocks write last (a finished cell is a value or +0.0), then waits for the stream.
python tools/host_lab/host_poll.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from linkteller_amd import _lib, engine, graph, synth
dev = torch.device("cuda:0")
n, f, h = 4385, 3170, 256
hg = graph.HipGraph(graph.first_order_gcn(synth.erdos_renyi_graph(n, 37304, seed=42)))
x = torch.from_numpy(synth.twitch_like_features(n, f, seed=1)).to(dev)
w = synth.gcn_weights(f, h, 2, seed=42)
base = engine.Baseline(hg, x, *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
nodes = torch.from_numpy(np.random.RandomState(42).choice(n, 500, replace=False).astype(np.int32)).to(dev)
out = torch.empty((500, 500), dtype=torch.float32, device=dev)
pin = torch.empty((500, 500), dtype=torch.float64).pin_memory()
hv = pin.numpy()
cur = torch.cuda.current_stream(dev)
a, b_, c = [], [], []
for it in range(320):
    hv.fill(np.nan)
    t0 = time.perf_counter()
    base.refresh("delta"); base.influence_rows(nodes, nodes, 1e-4, "delta", out=out, host=pin)
    t1 = time.perf_counter()
    while np.isnan(hv[250:, 499]).any() or np.isnan(hv[499, :]).any():
        pass
    t2 = time.perf_counter()
    cur.synchronize()
    t3 = time.perf_counter()
    if it >= 20:
        a.append(t1 - t0); b_.append(t2 - t0); c.append(t3 - t0)
    assert not np.isnan(hv).any()
med = lambda v: round(float(np.median(v)) * 1e6, 1)
print(f"enqueued at {med(a)} us, last column of the dense rows + the last row on the host at {med(b_)} us, stream wait returned at {med(c)} us")
