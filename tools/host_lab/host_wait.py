"""Which wait returns first after the host-landed step's last kernel: the stream wait, an event polled from Python, the device wait.
python tools/host_lab/host_wait.py"""
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
cur = torch.cuda.current_stream(dev)
ev = torch.cuda.Event()
ev_b = torch.cuda.Event(blocking=True)


def launch():
    base.refresh("delta"); base.influence_rows(nodes, nodes, 1e-4, "delta", out=out, host=pin)


def w_stream():
    launch(); cur.synchronize()


def w_poll():
    launch(); ev.record(cur)
    while not ev.query():
        pass


def w_event():
    launch(); ev.record(cur); ev.synchronize()


def w_blocking():
    launch(); ev_b.record(cur); ev_b.synchronize()


def w_device():
    launch(); torch.cuda.synchronize()


def med(fn, reps=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        torch.cuda.synchronize()
    return round(float(np.median(ts)) * 1e6, 1)


for name, fn in (("stream wait", w_stream), ("event polled from Python", w_poll), ("event wait", w_event), ("blocking-sync event", w_blocking),
                 ("device wait", w_device)):
    print(f"{name}: {med(fn)} us")
