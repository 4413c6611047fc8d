"""Where the host time of Attacker.influence_matrix() goes at BASELINE configs[1] (bench.py's `api_wall` leg): cProfile over N calls.
    python tools/api_profile.py [calls]"""
import argparse, cProfile, contextlib, io, os, pstats, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linkteller_amd import graph, synth
from linkteller_amd.attacker import Attacker
from linkteller_amd.gcn import GCN

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
adj, x, w = synth.twitch_like_problem("twitch-RU", hidden=256, n_classes=2, seed=0)
n, f = x.shape
a_hat = graph.first_order_gcn(adj)
model = GCN(f, 256, 2, 0.5)
model.load_state_dict({"gc1.weight": torch.from_numpy(w["W1"]), "gc1.bias": torch.from_numpy(w["b1"]),
                       "gc2.weight": torch.from_numpy(w["W2"]), "gc2.bias": torch.from_numpy(w["b2"])})
model.to(dev).eval()
wk = types.SimpleNamespace(features_2=torch.from_numpy(x).to(dev), adj_2=graph.sparse_mx_to_torch_sparse_tensor(a_hat).to(dev),
                           adj_ori=adj.tocsr(), n_nodes=n)
args = argparse.Namespace(dataset="twitch/RU", sample_type="unbalanced", n_test=500, sample_seed=42, influence=1e-4,
                          mode="vanilla-clean", attack_mode="efficient", influence_mode="delta")
atk = Attacker(args, model, wk)
with contextlib.redirect_stdout(io.StringIO()):
    atk.prepare_test_data()
for _ in range(20):
    atk.influence_matrix()
ts = []
for _ in range(200):
    t0 = time.perf_counter(); atk.influence_matrix(); ts.append(time.perf_counter() - t0)
print(f"median wall of a call: {np.median(ts) * 1e6:.1f} us (min {np.min(ts) * 1e6:.1f})")
pr = cProfile.Profile()
pr.enable()
for _ in range(calls):
    atk.influence_matrix()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("tottime")
st.print_stats(28)
print(f"per call (profiler on, {calls} calls):")
for line in s.getvalue().splitlines():
    print(line[:150])
