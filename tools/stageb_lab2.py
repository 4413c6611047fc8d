"""Which observed hubs cost what in stage B of a large call (R-MAT scale 21, 512 probes x 4096 observed): the biggest alone, all
but the biggest, only the moderate ones.  python tools/stageb_lab2.py"""
import os, sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linkteller_amd import _lib, engine, graph, synth
scale = 21
adj = synth.rmat_graph(scale, synth.rmat_draws(scale), seed=42)
a_hat = graph.first_order_gcn(adj)
n = adj.shape[0]
deg = np.diff(a_hat.indptr)
x = torch.from_numpy(synth.gaussian_features(n, 256, seed=1)).cuda()
w = synth.gcn_weights(256, 256, 2, seed=42)
base = engine.Baseline(graph.HipGraph(a_hat), x, *[torch.from_numpy(w[k]).cuda() for k in ("W1", "b1", "W2", "b2")])
rng = np.random.RandomState(42)
obs_all = rng.choice(n, 4096, replace=False)
probes = obs_all[:512]
small = np.flatnonzero(deg <= 128)
hub_idx = np.flatnonzero(deg[obs_all] > 128)
order = hub_idx[np.argsort(-deg[obs_all][hub_idx])]
print('observed hubs', len(hub_idx), 'degrees of the top 8', deg[obs_all][order[:8]], '| probe degrees top 8', np.sort(deg[probes])[-8:])
def variant(keep):
    o = obs_all.copy()
    drop = np.setdiff1d(hub_idx, keep)
    o[drop] = rng.choice(small, len(drop), replace=False)
    return o
def kernel_ms(name):
    tot, cnt = C.c_double(0), C.c_int64(0)
    _lib.check(_lib.lib().lt_profile_summary(_lib.KERNEL_IDS[name], C.byref(tot), C.byref(cnt)))
    return tot.value / max(cnt.value, 1)
small_h = hub_idx[deg[obs_all][hub_idx] < 1000]
sets = {"20 small hubs": small_h[:20], "60 small hubs": small_h[:60], "all hubs": hub_idx, "no hubs": np.array([], int), "top 1 only": order[:1], "top 8 only": order[:8], "all but top 8": order[8:],
        "hubs of < 1000 entries": hub_idx[deg[obs_all][hub_idx] < 1000]}
for mode in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("delta", "sparse")):
    for name, keep in sets.items():
        o = variant(keep)
        base.influence_rows(probes, o, 1e-4, mode); torch.cuda.synchronize()
        _lib.lib().lt_profile_reset(); _lib.lib().lt_profile_enable(0x1ff)
        for _ in range(3): base.influence_rows(probes, o, 1e-4, mode)
        torch.cuda.synchronize()
        if name == "hubs of < 1000 entries":      # the same without the eight biggest probes
            pr2 = probes[np.argsort(deg[probes])[:-8]]
            base.influence_rows(pr2, o, 1e-4, mode); torch.cuda.synchronize()
            _lib.lib().lt_profile_reset(); _lib.lib().lt_profile_enable(0x1ff)
            for _ in range(3): base.influence_rows(pr2, o, 1e-4, mode)
            torch.cuda.synchronize()
            print(f'{mode:6s} {name:24s} WITHOUT the 8 biggest probes: stage B {kernel_ms("item_stageB"):.3f} ms')
            _lib.lib().lt_profile_reset(); _lib.lib().lt_profile_enable(0x1ff)
            for _ in range(3): base.influence_rows(probes, o, 1e-4, mode)
            torch.cuda.synchronize()
        print(f'{mode:6s} {name:24s} ({len(keep):3d} hubs, {int(deg[obs_all][keep].sum()) if len(keep) else 0:7d} entries): stage B {kernel_ms("item_stageB"):.3f} ms')
        _lib.lib().lt_profile_enable(0)
