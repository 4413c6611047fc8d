"""The host-landed `delta` step at twitch-RU size, 100 times, for rocprofv3 --kernel-trace --stats: which launch carries the link's time.
python tools/host_lab/host_trace.py <share %> <waves> <stores in flight> [<share under the pre-activation %>]"""
import os, sys
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
_lib.set_tuning("export_zero_share", int(sys.argv[1]))
_lib.set_tuning("export_zero_blocks", int(sys.argv[2]))
_lib.set_tuning("export_zero_inflight", int(sys.argv[3]))
_lib.set_tuning("export_zero_share2", int(sys.argv[4]) if len(sys.argv) > 4 else 0)
for _ in range(100):
    base.refresh("delta"); base.influence_rows(nodes, nodes, 1e-4, "delta", out=out, host=pin)
    torch.cuda.synchronize()
