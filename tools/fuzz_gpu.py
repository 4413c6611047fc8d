"""Randomised cross-check on the GPU (run by hand: python tools/fuzz_gpu.py [cases] [seed]): graph families x
hidden widths x class counts x probe/observe lists, asserting
  full == sparse bit for bit, delta within 1e-5 of the fp64 oracle, full within 2x the oracle's own fp32 noise (+ the quantisation floor of a one-entry sample),
  exact zeros preserved, logits within 2e-5."""
import sys
import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, ".")
from linkteller_amd import _lib, engine, graph, synth    # noqa: E402
from oracle import linkteller_oracle as O                # noqa: E402


def oracle_matrix(a_hat, x, w, probes, observe, delta, dtype):
    adj_t = O.to_torch_sparse(a_hat).to(dtype)
    P = {k: torch.from_numpy(w[k]).to(dtype) for k in ("W1", "b1", "W2", "b2")}
    xt = torch.from_numpy(x).to(dtype)
    out = np.zeros((len(probes), len(observe)))
    with torch.no_grad():
        for i, v in enumerate(probes):
            g = O.get_gradient_eps_mat(xt, adj_t, P, int(v), delta)
            out[i] = g[torch.as_tensor(np.asarray(observe))].norm(dim=1).numpy()
    return out


def make_graph(kind, n, rng):
    if kind == "er":
        return synth.erdos_renyi_graph(n, int(n * rng.uniform(1.5, 8)), seed=int(rng.randint(1 << 30)))
    if kind == "pl":
        return synth.powerlaw_graph(n, int(n * rng.uniform(2, 8)), seed=int(rng.randint(1 << 30)), exponent=rng.uniform(1.8, 2.6))
    if kind == "star":
        rows = np.concatenate([np.zeros(n - 1, int), np.arange(1, n)])
        cols = np.concatenate([np.arange(1, n), np.roll(np.arange(1, n), 1)])
        a = sp.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n, n)).tocsr()
        return ((a + a.T) > 0).astype(np.float32).tocsr()
    if kind == "sparse_iso":      # many isolated nodes
        a = synth.erdos_renyi_graph(n, max(4, n // 4), seed=int(rng.randint(1 << 30)))
        return a
    raise ValueError(kind)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda", 0)
    worst = 0.0
    noisy = 0
    for it in range(cases):
        kind = rng.choice(["er", "pl", "pl", "star", "sparse_iso"])
        n = int(rng.choice([40, 130, 300, 700]))
        h = int(rng.choice([4, 10, 32, 100, 128, 132, 200, 256]))
        c = int(rng.choice([1, 2, 2, 3, 7, 8]))
        f = int(rng.choice([8, 33, 64]))
        norm = rng.choice(["FirstOrderGCN", "FirstOrderGCN", "AugNormAdj", "NormAdj"])
        adj = make_graph(kind, n, rng)
        a_hat = graph.fetch_normalization(norm)(adj).tocsr().astype(np.float32)
        a_hat.sort_indices()
        x = synth.gaussian_features(n, f, seed=it)
        w = synth.gcn_weights(f, h, c, seed=it + 1)
        base = engine.Baseline(graph.HipGraph(a_hat), torch.from_numpy(x).to(dev),
                               *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
        n_probe = int(rng.choice([1, 5, 17, 40, 70]))
        probes = rng.choice(n, min(n_probe, n), replace=bool(rng.randint(2)))
        observe = rng.choice(n, min(n, int(rng.choice([1, 30, 150]))), replace=False)
        ref64 = oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float64)
        ref32 = oracle_matrix(a_hat, x, w, probes, observe, 1e-4, torch.float32)
        scale = max(ref64.max(), 1e-6)
        res = {m: base.influence_rows(probes, observe, 1e-4, m).cpu().numpy().astype(np.float64) for m in ("full", "sparse", "delta")}
        # the routes of the item modes (pair marks, no per-probe bitmap + rows for big probes only, tiled layers, chunking)
        # picked at random: the same bits
        knobs = {"pair_marks": int(rng.choice([-1, 0])), "bits_max_bytes": int(rng.choice([1, 1 << 27])),
                 "item_bits": int(rng.choice([0, 1, 1])), "hub_short_side": int(rng.choice([0, 1])), "tiled_min_bytes": int(rng.choice([0, 1 << 25])),
                 "chunk_budget_bytes": int(rng.choice([1 << 14, 1 << 30])), "stageb_rows": int(rng.choice([0, 1])),
                 "z_on_demand": int(rng.choice([0, 1])), "pair_list": int(rng.choice([0, 1, 1]))}
        for k, v in knobs.items():
            _lib.set_tuning(k, v)
        try:
            base.refresh()
            for m in ("sparse", "delta"):
                alt = base.influence_rows(probes, observe, 1e-4, m).cpu().numpy().astype(np.float64)
                assert np.array_equal(alt, res[m]), (it, m, knobs)
        finally:
            for k in knobs:
                _lib.set_tuning(k, None)
        # the float64 matrix on the host by ONE call behind a refresh (lt_influence_rows_f64; on the fused route the head of the matrix is
        # zero-filled under the earlier launches, at random shares): the widened bits of the device matrix
        shares = {"export_zero_share": int(rng.choice([0, 35, 100])), "export_zero_share2": int(rng.choice([0, 15, 60])),
                  "export_zero_blocks": int(rng.choice([1, 16, 300]))}
        for k, v in shares.items():
            _lib.set_tuning(k, v)
        try:
            for m in ("delta", "sparse"):
                hostm = base.influence_matrix_host(probes, observe, 1e-4, m, refresh=True)
                assert hostm.dtype == np.float64 and np.array_equal(hostm, res[m]), (it, m, shares, "host matrix")
        finally:
            for k in shares:
                _lib.set_tuning(k, None)
        # the other fp64 route of `delta` (aggregate-first <-> S1d; feature rows <-> matrix cores): fp64 summation order only
        route_knobs = (("aggregate_first", 0), ("feature_delta", int(rng.choice([0, 1]))), ("defer_cref", int(rng.choice([0, 1]))),
                       ("s1_f32", int(rng.choice([0, 1]))))
        for k, v in route_knobs:
            _lib.set_tuning(k, v)
        try:
            base.refresh()
            other = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
            # on these S1d routes a graph without hub rows takes k_delta_probe_finish (round 4): the item kernels give its bits
            _lib.set_tuning("delta_fused", 0)
            unfused = base.influence_rows(probes, observe, 1e-4, "delta").cpu().numpy().astype(np.float64)
            assert np.array_equal(other, unfused), (it, kind, n, h, c, f, route_knobs, "delta_fused")
        finally:
            for k in ("aggregate_first", "feature_delta", "defer_cref", "s1_f32", "delta_fused"):
                _lib.set_tuning(k, None)
            base.refresh()
        # (this check is what caught plain fp32 storage of the feature route's product rows: seed 31337, case 59, 7e-5 of the largest
        # score; as 32-bit fixed point with a scale per row the routes agree to < 1e-6)
        assert np.abs(other - ref64).max() <= 1e-5 * scale and np.abs(other - res["delta"]).max() <= 2e-6 * scale, \
            (it, kind, n, h, c, f, route_knobs, np.abs(other - ref64).max() / scale, np.abs(other - res["delta"]).max() / scale, base.fp64_route())
        logits_ref = O.gcn_forward(torch.from_numpy(x).double(), O.to_torch_sparse(a_hat).double(),
                                   {k: torch.from_numpy(w[k]).double() for k in ("W1", "b1", "W2", "b2")}).numpy()
        tag = f"case {it}: {kind} n={n} H={h} C={c} F={f} {norm} probes={len(probes)} obs={len(observe)} maxdeg={int(np.diff(a_hat.indptr).max())}"
        assert np.array_equal(res["full"], res["sparse"]), tag
        ed = np.abs(res["delta"] - ref64).max() / scale
        assert ed <= 1e-5, (tag, ed)
        ef, e32 = np.abs(res["full"] - ref64).max(), np.abs(ref32 - ref64).max()
        if ef > 3.0 * e32 + 1e-4 * scale:
            print("NOISE?", tag, f"|full-ref64| {ef:.3e}  |ref32-ref64| {e32:.3e}  scale {scale:.3e}  nonzeros {(ref64 > 0).sum()}")
        # absolute floor: an fp32 finite difference carries ~ eps * |logit| / delta = 6e-4 * |logit| of noise whatever
        # the score is; a one-entry sample of the oracle's own noise can be anything
        # (round 4: the error of ONE entry is a handful of those steps -- one rounding of the difference to the ulp grid at every
        # CHANGED term of the observed row's sum, NOTES.md -- so the bound allows 8 of them and fails on the first case beyond it;
        # round 3 allowed 3 steps and tolerated one offending case in 50)
        floor = 6e-4 * max(1.0, float(np.abs(logits_ref).max()))
        if ef > 2.0 * e32 + 1e-3 * scale + 8.0 * floor:
            k = np.unravel_index(np.abs(res["full"] - ref64).argmax(), ref64.shape)
            print("FAIL", tag, f"|full-ref64| {ef:.3e}  |ref32-ref64| {e32:.3e}  scale {scale:.3e}  floor {floor:.3e}  at {k}: full {res['full'][k]:.6e} "
                  f"delta {res['delta'][k]:.6e} ref64 {ref64[k]:.6e} ref32 {ref32[k]:.6e}; ref32 errors sorted {np.sort(np.abs(ref32 - ref64).ravel())[-4:]}; "
                  f"full errors sorted {np.sort(np.abs(res['full'] - ref64).ravel())[-4:]}")
            raise SystemExit(1)
        for r in res.values():
            assert np.all(r[ref64 == 0] == 0), tag
        logits = base.logits().cpu().numpy().astype(np.float64)
        rl = O.gcn_forward(torch.from_numpy(x).double(), O.to_torch_sparse(a_hat).double(),
                           {k: torch.from_numpy(w[k]).double() for k in ("W1", "b1", "W2", "b2")}).numpy()
        assert np.abs(logits - rl).max() <= 2e-5 * max(1.0, np.abs(rl).max()), tag
        worst = max(worst, ed)
        print("ok", tag, f"delta err {ed:.1e}")
    print("all", cases, "cases ok; worst delta error", worst)


if __name__ == "__main__":
    main()
