"""Does a copy-engine transfer of 2 MB of zeros into pinned host memory on a second stream run beside the step's kernels?
python tools/host_lab/host_sdma.py"""
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
zeros = torch.zeros((500, 500), dtype=torch.float64, device=dev)
side = torch.cuda.Stream(dev)
cur = torch.cuda.current_stream(dev)


def step():
    base.refresh("delta"); base.influence_rows(nodes, nodes, 1e-4, "delta", out=out)


def copy_side():
    with torch.cuda.stream(side):
        pin.copy_(zeros, non_blocking=True)


def memset_side():
    with torch.cuda.stream(side):
        pin.zero_()


def wall(fn, waiter, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); waiter(); ts.append(time.perf_counter() - t)
        torch.cuda.synchronize()
    return round(float(np.median(ts)) * 1e6, 1)


print("step alone:", wall(step, cur.synchronize), "us; copy alone:", wall(copy_side, side.synchronize), "us; fill kernel alone:", wall(memset_side, side.synchronize), "us")
print("copy beside the step: the step's stream done at", wall(lambda: (copy_side(), step()), cur.synchronize), "us, both at",
      wall(lambda: (copy_side(), step()), torch.cuda.synchronize), "us")
print("fill kernel beside the step: the step's stream done at", wall(lambda: (memset_side(), step()), cur.synchronize), "us, both at",
      wall(lambda: (memset_side(), step()), torch.cuda.synchronize), "us")
ev = torch.cuda.Event()


def joined():
    copy_side(); ev.record(side)
    base.refresh("delta")
    cur.wait_event(ev)
    base.influence_rows(nodes, nodes, 1e-4, "delta", out=out)


print("copy beside the step, the step's stream waits for it in front of its first launch:", wall(joined, cur.synchronize), "us")
