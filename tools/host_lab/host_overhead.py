"""The host side of Baseline.influence_matrix_host at twitch-RU size: what each piece of Python / runtime costs per call.
python tools/host_lab/host_overhead.py"""
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


def med(fn, reps=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return round(float(np.median(ts)) * 1e6, 2)


cur = torch.cuda.current_stream(dev)
print("torch.empty pinned [500, 500] f64:", med(lambda: torch.empty((500, 500), dtype=torch.float64, pin_memory=True)), "us")
print("_as_nodes x2:", med(lambda: (engine._as_nodes(nodes, n, dev, "p"), engine._as_nodes(nodes, n, dev, "o"))), "us")
print("refresh():", med(lambda: base.refresh("delta")), "us")
print("node_check():", med(engine.node_check), "us")
print(".numpy():", med(lambda: pin.numpy()), "us")
print("idle stream synchronize:", med(cur.synchronize), "us; idle device synchronize:", med(torch.cuda.synchronize), "us")
print("engine._stream():", med(engine._stream), "us")


def launches():
    base.refresh("delta"); base.influence_rows(nodes, nodes, 1e-4, "delta", out=out, host=pin)


def t_enqueue():
    t = time.perf_counter(); launches(); e = time.perf_counter() - t
    cur.synchronize()
    return e


for _ in range(20):
    t_enqueue()
print("enqueue only (refresh + influence_rows(host=)):", round(float(np.median([t_enqueue() for _ in range(300)])) * 1e6, 2), "us")
print("whole influence_matrix_host(refresh=True):", med(lambda: base.influence_matrix_host(nodes, nodes, 1e-4, "delta", refresh=True)), "us")
print("enqueue + stream wait:", med(lambda: (launches(), cur.synchronize())), "us")
