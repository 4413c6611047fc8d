"""fp32 noise of `full` on the hub-of-many-segments case of tests/test_gpu_round3.py, over ALL 100 probe rows: |ours - ref64| and
|ref32 - ref64| per row (max over the observed columns), their rms / max, and where the row maxima sit (the hub column or not)."""
import os, sys
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linkteller_amd import engine, graph, synth
from oracle import linkteller_oracle as O
n = 9500
rng = np.random.RandomState(4)
r = rng.randint(1, n, 30000); c = rng.randint(1, n, 30000)
keep = r != c
rows = np.concatenate([np.zeros(n - 1, int), r[keep]]); cols = np.concatenate([np.arange(1, n), c[keep]])
a = sp.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n, n)).tocsr()
a = ((a + a.T) > 0).astype(np.float32).tocsr()
a_hat = graph.first_order_gcn(a)
x = synth.twitch_like_features(n, 200, seed=6, density=0.03)
w = synth.gcn_weights(200, 256, 2, seed=8)
probes = np.concatenate([np.arange(1, 71), rng.choice(np.arange(200, n), 30, replace=False)]).astype(np.int32)
obs = np.concatenate([[0], np.arange(1, 40), rng.choice(np.arange(200, n), 60, replace=False)]).astype(np.int32)
dev = torch.device("cuda:0")
base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(dev), *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
f = base.influence_rows(probes, obs, 1e-4, "full").cpu().numpy().astype(np.float64)
d = base.influence_rows(probes, obs, 1e-4, "delta").cpu().numpy().astype(np.float64)
adj_o = O.to_torch_sparse(a_hat)
ref = {}
for dt in (torch.float32, torch.float64):
    P = {k: torch.from_numpy(w[k]).to(dt) for k in ("W1", "b1", "W2", "b2")}
    out = np.zeros((len(probes), len(obs)))
    with torch.no_grad():
        for i, v in enumerate(probes):
            gm = O.get_gradient_eps_mat(torch.from_numpy(x).to(dt), adj_o.to(dt), P, int(v), 1e-4)
            out[i] = gm[torch.as_tensor(obs.astype(np.int64))].norm(dim=1).double().numpy()
    ref[dt] = out
eo, et = np.abs(f - ref[torch.float64]), np.abs(ref[torch.float32] - ref[torch.float64])
print("max score", ref[torch.float64].max(), " delta vs ref64 max", np.abs(d - ref[torch.float64]).max())
for name, e in (("ours", eo), ("ref32", et)):
    rm = e.max(axis=1)
    print(f"{name}: per-row max: rms {np.sqrt((rm ** 2).mean()):.5f} median {np.median(rm):.5f} max {rm.max():.5f}; row maxima at the hub column: "
          f"{int((e.argmax(axis=1) == 0).sum())} of {len(rm)}; all-entries rms {np.sqrt((e ** 2).mean()):.6f}; hub column rms {np.sqrt((e[:, 0] ** 2).mean()):.5f}; "
          f"non-hub columns rms {np.sqrt((e[:, 1:] ** 2).mean()):.6f} max {e[:, 1:].max():.5f}")
