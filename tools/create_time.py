import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from linkteller_amd import _lib, engine, graph, synth
import ctypes as C
adj = synth.rmat_graph(21, synth.rmat_draws(21), seed=42)
a_hat = graph.first_order_gcn(adj)
n = a_hat.shape[0]
torch.zeros(1).cuda(); torch.cuda.synchronize()
t0 = time.time(); _n, rp, ci, va = graph.csr_arrays(a_hat); t1 = time.time()
print("csr_arrays (host conversions)", round(t1 - t0, 3), "s")
t0 = time.time(); hg = graph.HipGraph(a_hat); torch.cuda.synchronize(); t1 = time.time()
print("HipGraph(a_hat) total", round(t1 - t0, 3), "s")
if rp is not None:
    out = C.c_void_p()
    t0 = time.time()
    rc = _lib.lib().lt_graph_create(n, a_hat.nnz, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, C.byref(out))
    torch.cuda.synchronize(); t1 = time.time()
    print("lt_graph_create alone", round(t1 - t0, 3), "s rc", rc)
x = torch.from_numpy(synth.gaussian_features(n, 256, seed=1)).cuda()
w = synth.gcn_weights(256, 256, 2, seed=42)
p = [torch.from_numpy(w[k]).cuda() for k in ("W1", "b1", "W2", "b2")]
torch.cuda.synchronize(); t0 = time.time(); base = engine.Baseline(hg, x, *p); torch.cuda.synchronize(); t1 = time.time()
print("Baseline create", round(t1 - t0, 3), "s")
