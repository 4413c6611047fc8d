"""Attacker.influence_matrix() at twitch-RU size under cProfile: the Python around the one library call.
python tools/host_lab/api_profile.py"""
import argparse, contextlib, cProfile, io, os, pstats, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from linkteller_amd import graph, synth
from linkteller_amd.attacker import Attacker
from linkteller_amd.gcn import GCN
dev = torch.device("cuda:0")
n, f, h, c = 4385, 3170, 256, 2
adj = synth.erdos_renyi_graph(n, 37304, seed=42)
a_hat = graph.first_order_gcn(adj)
x = torch.from_numpy(synth.twitch_like_features(n, f, seed=1)).to(dev)
w = synth.gcn_weights(f, h, c, seed=42)
model = GCN(f, h, c, 0.5)
model.load_state_dict({"gc1.weight": torch.from_numpy(w["W1"]), "gc1.bias": torch.from_numpy(w["b1"]),
                       "gc2.weight": torch.from_numpy(w["W2"]), "gc2.bias": torch.from_numpy(w["b2"])})
model.to(dev).eval()
wk = types.SimpleNamespace(features_2=x, adj_2=graph.sparse_mx_to_torch_sparse_tensor(a_hat).to(dev), adj_ori=adj.tocsr(), n_nodes=n)
args = argparse.Namespace(dataset="twitch/RU", sample_type="unbalanced", n_test=500, sample_seed=42, influence=1e-4,
                          mode="vanilla-clean", attack_mode="efficient", influence_mode="delta")
atk = Attacker(args, model, wk)
with contextlib.redirect_stdout(io.StringIO()):
    atk.prepare_test_data()
for _ in range(20):
    atk.influence_matrix()
ts = []
for _ in range(300):
    t = time.perf_counter(); atk.influence_matrix(); ts.append(time.perf_counter() - t)
print("influence_matrix(): median", round(float(np.median(ts)) * 1e6, 1), "us")
pr = cProfile.Profile()
pr.enable()
for _ in range(1000):
    atk.influence_matrix()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
