"""BASELINE configs[4], one rank of 8 on one GPU (bench.py's `influence_shard`): 512 probes x 4096 observed nodes on the R-MAT
scale-21 graph, F = H = 256 -- `delta` with and without the loop-invariant baseline, device-resident node lists, median of 5.
python tools/shard_time.py [scale] [modes]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from linkteller_amd import engine, graph, synth
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 21
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["delta"]
dev = torch.device("cuda:0")
big = graph.first_order_gcn(synth.rmat_graph(scale, synth.rmat_draws(scale), seed=42))
n = big.shape[0]
gb = graph.HipGraph(big)
xb = torch.from_numpy(synth.gaussian_features(n, 256, seed=1)).to(dev)
wb = synth.gcn_weights(256, 256, 2, seed=42)
bb = engine.Baseline(gb, xb, *[torch.from_numpy(wb[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
ob_np = np.random.RandomState(42).choice(n, 4096, replace=False)
ob = torch.from_numpy(ob_np.astype(np.int32)).to(dev)
pb = ob[:512].contiguous()
out = torch.empty((512, 4096), dtype=torch.float32, device=dev)


def wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return round(float(np.median(ts)) * 1e3, 3)


if os.environ.get("LT_SHARD_TRACE"):      # rocprofv3: the build incl. the baseline alone, 10 times (tools/host_lab/shard_trace.sh)
    for m in modes:
        for _ in range(10):
            bb.refresh(m); bb.influence_rows(pb, ob, 1e-4, m, out=out)
        torch.cuda.synchronize()
    sys.exit(0)
for m in modes:
    a = wall(lambda: bb.influence_rows(pb, ob, 1e-4, m, out=out))
    b = wall(lambda: (bb.refresh(m), bb.influence_rows(pb, ob, 1e-4, m, out=out)))
    print(f"{m}: {a} ms, incl. baseline {b} ms   (checksum {float(out.double().sum()):.6f}, nonzero {int((out > 0).sum())})")


if "delta" in modes and bb.fp64_route() == 2:
    # one rank of 8 with the hub rows all 4096 probes reach split over the ranks (dist.SharedHubRows): 1/8 formed here, the others'
    # adopted from a buffer filled beforehand (the all-gather is not in the figure)
    from linkteller_amd import dist as lt_dist
    rows_all = bb.reached_rows(ob, lt_dist.HUB_ROW_MIN_ENTRIES)
    nh = rows_all.numel()
    per8 = (nh + 7) // 8
    allbuf = torch.empty((nh, 256), dtype=torch.float64, device=dev)
    bb.refresh("delta"); bb.form_rows_fp64(rows_all); bb.gather_rows_fp64(rows_all, allbuf)
    mine = rows_all[:per8].contiguous()
    send = torch.empty((per8, 256), dtype=torch.float64, device=dev)
    bb.refresh("delta"); bb.influence_rows(pb, ob, 1e-4, "delta", out=out)
    plain = out.clone()

    def hub_step():
        bb.refresh("delta"); bb.form_rows_fp64(mine); bb.gather_rows_fp64(mine, send); bb.scatter_rows_fp64(rows_all, allbuf)
        bb.influence_rows(pb, ob, 1e-4, "delta", out=out)
    c = wall(hub_step)
    parts = {"form my 1/8 of the hub rows": wall(lambda: (bb.refresh("delta"), bb.form_rows_fp64(mine))),
             "pack": wall(lambda: bb.gather_rows_fp64(mine, send)), "adopt all": wall(lambda: bb.scatter_rows_fp64(rows_all, allbuf))}
    print(f"delta, hub rows shared over 8 ranks: {nh} rows of >= {lt_dist.HUB_ROW_MIN_ENTRIES} entries, {per8} formed here, all-gather "
          f"{8 * per8 * 256 * 8 / 1e6:.1f} MB: incl. baseline {c} ms  (same bits: {bool(torch.equal(out, plain))}); {parts}")
