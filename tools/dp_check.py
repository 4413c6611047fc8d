"""FULL / SPARSE / DELTA on DP-perturbed twitch-like graphs (EdgeRand makes the served adjacency much denser):
python tools/dp_check.py <discrete|continuous> <eps>"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from linkteller_amd import dp, engine, graph, synth
kind, eps = sys.argv[1], float(sys.argv[2])
adj, x, w = synth.twitch_like_problem("twitch-RU", hidden=256, n_classes=2, seed=0)
served = dp.perturb_adj(adj, kind, eps, noise_seed=42)
a_hat = graph.first_order_gcn(served)
d = np.diff(a_hat.indptr)
print(kind, eps, 'nnz', a_hat.nnz, 'mean deg', round(d.mean(), 1), 'max', d.max(), 'rows > 128:', int((d > 128).sum()))
dev = torch.device('cuda', 0)
base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(dev), *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
nodes = np.random.RandomState(42).choice(adj.shape[0], 500, replace=False)
res = {}
for m in ('full', 'sparse', 'delta'):
    for _ in range(2):
        base.refresh(); res[m] = base.influence_rows(nodes, nodes, 1e-4, m)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        base.refresh(); res[m] = base.influence_rows(nodes, nodes, 1e-4, m)
    torch.cuda.synchronize(); print(m, round((time.time() - t0) / 5 * 1e3, 3), 'ms/step')
print('full == sparse:', bool(torch.equal(res['full'], res['sparse'])), ' |full - delta| max', float((res['full'] - res['delta']).abs().max()), 'max', float(res['delta'].max()))
