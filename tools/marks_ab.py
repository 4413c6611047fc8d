"""pair_marks on / off at twitch-RU size for growing n_test (where should the automatic switch sit?).
python tools/marks_ab.py [powerlaw]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linkteller_amd import _lib, engine, graph, synth
pl = len(sys.argv) > 1 and sys.argv[1] == "powerlaw"
n, m = synth.TWITCH_SHAPES["twitch-RU"]["n"], synth.TWITCH_SHAPES["twitch-RU"]["e"]
adj = (synth.powerlaw_graph if pl else synth.erdos_renyi_graph)(n, m, seed=0)
a_hat = graph.first_order_gcn(adj)
x = torch.from_numpy(synth.gaussian_features(n, 512, seed=1)).cuda()
w = synth.gcn_weights(512, 256, 2, seed=42)
base = engine.Baseline(graph.HipGraph(a_hat), x, *[torch.from_numpy(w[k]).cuda() for k in ("W1", "b1", "W2", "b2")])
rng = np.random.RandomState(0)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return r, (time.time() - t) / reps * 1e3
for nt in (250, 500, 1000, 2000, 4000):
    nodes = rng.choice(n, nt, replace=False)
    row = []
    for mode in ("sparse", "delta"):
        res = {}
        for pm in (-1, 0):
            _lib.set_tuning("pair_marks", pm)
            res[pm] = timed(lambda: base.influence_rows(nodes, nodes, 1e-4, mode))
        row.append(f"{mode}: off {res[-1][1]:.3f} on {res[0][1]:.3f} ms same={bool(torch.equal(res[-1][0], res[0][0]))}")
    print(f"n_test {nt} ({nt * nt} pairs)  " + "   ".join(row))
_lib.set_tuning("pair_marks", None)
