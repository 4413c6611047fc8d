import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_golden, csr_from, golden_args
from linkteller_amd import engine, graph
g = load_golden('influence.npz')
for key in ['pl600','lap600']:
    args = golden_args(g,key)
    served = csr_from(g, f"{key}.served") if f"{key}.served.n" in g else csr_from(g, f"{key}.adj")
    a_hat = graph.fetch_normalization(args["norm"])(served)
    x = torch.from_numpy(g[f"{key}.x"]).cuda()
    p = [torch.from_numpy(g[f"{key}.sd.{k}"]).cuda() for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias")]
    base = engine.Baseline(graph.HipGraph(a_hat), x, *p)
    nodes = g[f"{key}.ref32.test_nodes"]
    f = base.influence_rows(nodes, nodes, 1e-4, 'full').cpu().numpy()
    s = base.influence_rows(nodes, nodes, 1e-4, 'sparse').cpu().numpy()
    ref64 = g[f"{key}.ref64.influence_val"]; ref32 = g[f"{key}.ref32.influence_val"]
    print(key, 'bitwise full==sparse', np.array_equal(f, s), '|full-ref64|', np.abs(f-ref64).max(), '|ref32-ref64|', np.abs(ref32-ref64).max(), 'zeros ok', bool(np.all(f[ref64==0]==0)))
