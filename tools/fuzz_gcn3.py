"""Randomised cross-check of the 3-layer probe primitive (python tools/fuzz_gcn3.py [cases] [seed]): graph families x
widths x classes x probe / observe lists against the oracle's fp64 / fp32 finite difference of the 3-layer forward."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from fuzz_gpu import make_graph                              # noqa: E402
from linkteller_amd import engine, graph, synth, _lib       # noqa: E402
from oracle import linkteller_oracle as O                    # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda", 0)
    for it in range(cases):
        kind = rng.choice(["er", "pl", "pl", "star", "sparse_iso"])
        n = int(rng.choice([40, 130, 300, 700]))
        h1 = int(rng.choice([4, 10, 32, 100, 132, 256]))
        h2 = int(rng.choice([4, 16, 36, 128, 256]))
        c = int(rng.choice([1, 2, 2, 3, 8]))
        f = int(rng.choice([8, 33, 64]))
        norm = rng.choice(["FirstOrderGCN", "FirstOrderGCN", "AugNormAdj", "NormAdj"])
        a_hat = graph.fetch_normalization(norm)(make_graph(kind, n, rng)).tocsr().astype(np.float32)
        a_hat.sort_indices()
        x = synth.gaussian_features(n, f, seed=it)

        def u(shape, fan):
            s = 1.0 / np.sqrt(fan)
            return rng.uniform(-s, s, size=shape).astype(np.float32)

        P = dict(W1=u((f, h1), h1), b1=u((h1,), h1), W2=u((h1, h2), h2), b2=u((h2,), h2), W3=u((h2, c), c), b3=u((c,), c))
        if rng.randint(3) == 0:
            _lib.set_tuning("chunk_budget_bytes", int(rng.choice([50_000, 400_000])))
        base = engine.Baseline3(graph.HipGraph(a_hat), torch.from_numpy(x).to(dev),
                                *[torch.from_numpy(P[k]).to(dev) for k in ("W1", "b1", "W2", "b2", "W3", "b3")])
        probes = rng.choice(n, min(n, int(rng.choice([1, 5, 17, 40]))), replace=bool(rng.randint(2)))
        observe = rng.choice(n, min(n, int(rng.choice([1, 30, 150]))), replace=False)
        got = base.influence_rows(probes, observe, 1e-4).cpu().numpy().astype(np.float64)
        _lib.set_tuning("chunk_budget_bytes", None)
        adj_t = O.to_torch_sparse(a_hat)
        ref = {}
        for dt in (torch.float64, torch.float32):
            Pd = {k: torch.from_numpy(v).to(dt) for k, v in P.items()}
            xt = torch.from_numpy(x).to(dt)
            m = np.zeros((len(probes), len(observe)))
            with torch.no_grad():
                for i, v in enumerate(probes):
                    gm = O.get_gradient_eps_mat(xt, adj_t.to(dt), Pd, int(v), 1e-4, forward=O.gcn3_forward)
                    m[i] = gm[torch.as_tensor(np.asarray(observe))].norm(dim=1).numpy()
            ref[dt] = m
        r64, r32 = ref[torch.float64], ref[torch.float32]
        scale = max(r64.max(), 1e-6)
        e32 = np.abs(r32 - r64).max()
        ef = np.abs(got - r64).max()
        logits_ref = O.gcn3_forward(torch.from_numpy(x).double(), adj_t.double(), {k: torch.from_numpy(v).double() for k, v in P.items()}).numpy()
        floor = 6e-4 * max(1.0, float(np.abs(logits_ref).max()))
        tag = f"case {it}: {kind} n={n} H1={h1} H2={h2} C={c} F={f} {norm} probes={len(probes)} obs={len(observe)} maxdeg={int(np.diff(a_hat.indptr).max())}"
        assert np.all(got[r64 == 0] == 0), tag
        if ef > 10.0 * e32 + 1e-3 * scale + 3.0 * floor:
            print("FAIL", tag, f"|ours-ref64| {ef:.3e}  |ref32-ref64| {e32:.3e}  scale {scale:.3e}")
            raise SystemExit(1)
        logits = base.logits().cpu().numpy().astype(np.float64)
        assert np.abs(logits - logits_ref).max() <= 2e-5 * max(1.0, np.abs(logits_ref).max()), tag
        print("ok", tag, f"err {ef:.1e} (ref32 {e32:.1e})")
    print("all", cases, "cases ok")


if __name__ == "__main__":
    main()
