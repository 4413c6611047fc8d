"""Where SPARSE / DELTA stage B spends its time on a large R-MAT call (BASELINE configs[4] shard: 512 probes x 4096
observed): pair marks on / off, observed hubs in / out.  python tools/stageb_lab.py [scale] [n_probe] [n_obs]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linkteller_amd import _lib, engine, graph, synth
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n_probe = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n_obs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
adj = synth.rmat_graph(scale, synth.rmat_draws(scale), seed=42)
a_hat = graph.first_order_gcn(adj)
n = adj.shape[0]
deg = np.diff(a_hat.indptr)
x = torch.from_numpy(synth.gaussian_features(n, 256, seed=1)).cuda()
w = synth.gcn_weights(256, 256, 2, seed=42)
base = engine.Baseline(graph.HipGraph(a_hat), x, *[torch.from_numpy(w[k]).cuda() for k in ("W1", "b1", "W2", "b2")])
rng = np.random.RandomState(42)
obs_all = rng.choice(n, n_obs, replace=False)
probes = obs_all[:n_probe]
obs_nohub = obs_all.copy()
small = np.flatnonzero(deg <= 128)
obs_nohub[deg[obs_all] > 128] = rng.choice(small, int((deg[obs_all] > 128).sum()), replace=False)
print('observed hubs', int((deg[obs_all] > 128).sum()), 'their entries', int(deg[obs_all][deg[obs_all] > 128].sum()),
      'max', int(deg[obs_all].max()), '| probe hubs', int((deg[probes] > 128).sum()), 'sum |R_v|', int(deg[probes].sum()))
import ctypes as C
def kernel_ms(name):
    tot, cnt = C.c_double(0), C.c_int64(0)
    _lib.check(_lib.lib().lt_profile_summary(_lib.KERNEL_IDS[name], C.byref(tot), C.byref(cnt)))
    return tot.value / max(cnt.value, 1)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return r, (time.time() - t) / reps * 1e3
ref = {}
for name, obs in (("with hubs", obs_all), ("no observed hubs", obs_nohub)):
    for pm in (-1, 0):
        _lib.set_tuning("pair_marks", pm)
        for mode in ("sparse", "delta"):
            r, ms = timed(lambda: base.influence_rows(probes, obs, 1e-4, mode))
            _lib.lib().lt_profile_reset(); _lib.lib().lt_profile_enable(0x1ff)
            base.influence_rows(probes, obs, 1e-4, mode); torch.cuda.synchronize()
            kk = {k: round(kernel_ms(k), 3) for k in ("gemm", "item_stageA", "item_stageB")}
            _lib.lib().lt_profile_enable(0)
            key = (name, mode)
            same = '' if key not in ref else ' same bits: %s' % bool(torch.equal(ref[key], r))
            ref.setdefault(key, r)
            print(f'{name:18s} pair_marks={pm:2d} {mode:6s} {ms:8.3f} ms{same}  {kk}')
_lib.set_tuning("pair_marks", None)
