"""could the record route (k_delta_probe_finish) serve SOME probes of a graph with hub rows?  A node qualifies when
its incidence count is under the cap (4096) and none of its neighbours' rows is a hub row (> 128 entries: LT_ROW_SEG) -- its
record is then built from short rows only.  Host arithmetic on the bench graphs (no GPU).  python tools/records_histogram.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linkteller_amd import graph, synth

for name, pl in (("Erdos-Renyi (the headline graph)", False), ("power-law (workload_2: what MUSAE graphs look like)", True)):
    adj, _, _ = synth.twitch_like_problem("twitch-RU", hidden=256, n_classes=2, seed=0, powerlaw=pl)
    a = graph.first_order_gcn(adj).tocsr()
    n = a.shape[0]
    deg = np.diff(a.indptr)
    hub = deg > 128
    # incidences of node v: sum over r in col(v) of |col(r)| (the matrix is symmetric: col = row)
    inc = np.array([deg[a.indices[a.indptr[v]:a.indptr[v + 1]]].sum() for v in range(n)])
    adj_hub = np.array([hub[a.indices[a.indptr[v]:a.indptr[v + 1]]].any() for v in range(n)])
    ok = (inc <= 4096) & ~adj_hub
    print(f"== {name}: n {n}, nnz {a.nnz}, hub rows (> 128 entries) {int(hub.sum())}, max row {int(deg.max())}")
    print(f"   incidences per node: mean {inc.mean():.0f}, median {np.median(inc):.0f}, p90 {np.percentile(inc, 90):.0f}, max {inc.max()}; "
          f"over the 4096 cap: {100.0 * (inc > 4096).mean():.1f} %")
    print(f"   nodes with a hub row among their neighbours (or being one): {100.0 * adj_hub.mean():.1f} %")
    print(f"   nodes the record route could serve (under the cap, no hub neighbour): {100.0 * ok.mean():.1f} %")
    edges = [0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 1 << 30]
    h, _ = np.histogram(inc, bins=edges)
    print("   histogram of incidences:", ", ".join(f"<= {edges[i + 1] if i + 2 < len(edges) else 'inf'}: {100.0 * h[i] / n:.1f} %" for i in range(len(h))))
    np.random.seed(42)
    tn = np.random.choice(np.arange(n), 500, replace=False)
    print(f"   of the 500 test nodes of the bench: {int(ok[tn].sum())} qualify")
