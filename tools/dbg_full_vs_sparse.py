import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_golden, csr_from, golden_args
from linkteller_amd import engine, graph
g = load_golden('influence.npz')
for key in ['pl600','pl600hi','lap600','er300']:
    args = golden_args(g,key)
    served = csr_from(g, f"{key}.served") if f"{key}.served.n" in g else csr_from(g, f"{key}.adj")
    a_hat = graph.fetch_normalization(args["norm"])(served)
    x = torch.from_numpy(g[f"{key}.x"]).cuda()
    p = [torch.from_numpy(g[f"{key}.sd.{k}"]).cuda() for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias")]
    base = engine.Baseline(graph.HipGraph(a_hat), x, *p)
    nodes = g[f"{key}.ref32.test_nodes"]
    f = base.influence_rows(nodes, nodes, 1e-4, 'full').cpu().numpy()
    s = base.influence_rows(nodes, nodes, 1e-4, 'sparse').cpu().numpy()
    d = np.argwhere(f != s)
    print(key, 'n_test', len(nodes), 'mismatches', len(d), d[:10].tolist(), [ (float(f[i,j]), float(s[i,j])) for i,j in d[:5]])
