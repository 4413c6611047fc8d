"""Sanity/perf on an R-MAT graph (BASELINE configs[4] shape, scaled): full == sparse bitwise, timings per mode."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linkteller_amd import engine, graph, synth
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_test = int(sys.argv[2]) if len(sys.argv) > 2 else 128
modes = sys.argv[3].split(',') if len(sys.argv) > 3 else ['full', 'sparse', 'delta']
t0 = time.time()
adj = synth.rmat_graph(scale, synth.rmat_draws(scale), seed=42)
a_hat = graph.first_order_gcn(adj)
n = adj.shape[0]
print('n', n, 'nnz', a_hat.nnz, 'max deg', int(np.diff(a_hat.indptr).max()), ' host graph build', round(time.time() - t0, 1), 's')
x = torch.from_numpy(synth.gaussian_features(n, 256, seed=1)).cuda()
w = synth.gcn_weights(256, 256, 2, seed=42)
t0 = time.time()
base = engine.Baseline(graph.HipGraph(a_hat), x, *[torch.from_numpy(w[k]).cuda() for k in ("W1", "b1", "W2", "b2")])
torch.cuda.synchronize(); print('baseline create', round(time.time() - t0, 3), 's')
np.random.seed(42)
n_obs = int(sys.argv[4]) if len(sys.argv) > 4 else n_test      # observed nodes (BASELINE configs[4]: 4096, probes sharded 512 per GPU)
obs = np.random.choice(np.arange(n), n_obs, replace=False)
nodes = obs[:n_test] if n_test <= n_obs else np.random.choice(np.arange(n), n_test, replace=False)
res = {}
for m in modes:
    base.influence_rows(nodes, obs, 1e-4, m); torch.cuda.synchronize()
    t0 = time.time(); res[m] = base.influence_rows(nodes, obs, 1e-4, m); torch.cuda.synchronize()
    print(m, f'{len(nodes)} probes x {len(obs)} observed:', round((time.time() - t0) * 1e3, 3), 'ms')
for m in modes:
    if m == 'full':
        continue
    t0 = time.time(); base.refresh(); base.influence_rows(nodes, obs, 1e-4, m); torch.cuda.synchronize()
    print(m, 'incl. baseline refresh (X*W1 + layers):', round((time.time() - t0) * 1e3, 3), 'ms')
if 'full' in res and 'sparse' in res:
    print('full == sparse:', bool(torch.equal(res['full'], res['sparse'])))
if 'sparse' in res and 'delta' in res:
    print('max', float(res['delta'].max()), ' |sparse-delta| max', float((res['sparse'] - res['delta']).abs().max()),
          ' nonzero pairs', int((res['delta'] > 0).sum()), 'of', res['delta'].numel())
