"""Where the host-landed step's time goes (twitch-RU size, `delta`): the device-resident step, + the export launch, the one-call
form with the probes' blocks widening their rows, and the announced form (zero-fill on the side stream + touched positions) --
into pinned host memory and into device memory (the same launches without the link).  Median wall of 200 steps each.
"share": the per cent of the matrix's rows zero-filled by blocks riding in the product rows' launch ("export_zero_share").
python tools/host_lab/host_step.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from linkteller_amd import _lib, engine, graph, synth
dev = torch.device("cuda:0")
n, f, h = 4385, 3170, 256
adj = synth.erdos_renyi_graph(n, 37304, seed=42)
hg = graph.HipGraph(graph.first_order_gcn(adj))
x = torch.from_numpy(synth.twitch_like_features(n, f, seed=1)).to(dev)
w = synth.gcn_weights(f, h, 2, seed=42)
base = engine.Baseline(hg, x, *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
nodes = torch.from_numpy(np.random.RandomState(42).choice(n, 500, replace=False).astype(np.int32)).to(dev)
out = torch.empty((500, 500), dtype=torch.float32, device=dev)
pin = torch.empty((500, 500), dtype=torch.float64).pin_memory()
d64 = torch.empty((500, 500), dtype=torch.float64, device=dev)
st = engine._stream


def v_device():
    base.refresh("delta"); base.influence_rows(nodes, nodes, 1e-4, "delta", out=out)


def v_export(t):
    base.refresh("delta"); base.influence_rows(nodes, nodes, 1e-4, "delta", out=out)
    _lib.check(_lib.lib().lt_export_rows_f64(out.data_ptr(), 500, 500, 500, t.data_ptr(), 500, st()), "export")


def v_onecall(t):
    base.refresh("delta"); base.influence_rows(nodes, nodes, 1e-4, "delta", out=out, host=t)


def wall(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return round(float(np.median(ts)) * 1e6, 1)


print("device-resident step:", wall(v_device), "us")
for name, t in (("pinned host", pin),):
    print(f"{name}: + export launch {wall(lambda: v_export(t))} us")
    for share in (0, 25, 30, 35, 40, 45):
        _lib.set_tuning("export_zero_share", share)
        row = [f"share under the product rows {share} %, under the pre-activation:"]
        for s2 in (0, 10, 15, 20, 25):
            _lib.set_tuning("export_zero_share2", s2)
            row.append(f"{s2} %: {wall(lambda: v_onecall(t))} us")
        print("   one call,", " ".join(row))
