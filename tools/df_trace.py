"""Where does k_delta_probe_finish spend its time?  Needs the library built with -DLT_DF_TRACE
(make -C linkteller_amd/csrc clean all CXXEXTRA=-DLT_DF_TRACE); prints, per phase, when the waves of one launch pass it
(us after the first wave's entry: median / 90 % / last).  GPU box: python tools/df_trace.py [n_test]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linkteller_amd import _lib, engine, graph, synth  # noqa: E402


def main():
    n_test = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    dev = torch.device("cuda:0")
    adj, x_np, w = synth.twitch_like_problem("twitch-RU", hidden=256, n_classes=2, seed=0, powerlaw=False)
    a_hat = graph.first_order_gcn(adj)
    n = x_np.shape[0]
    np.random.seed(42)
    test_nodes = np.random.choice(np.arange(n), n_test, replace=False).astype(np.int32)
    hg = graph.HipGraph(a_hat)
    x = torch.from_numpy(x_np).to(dev)
    base = engine.Baseline(hg, x, *[torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
    base.enable_fp64()
    obs = torch.from_numpy(test_nodes).to(dev)
    out = torch.empty((n_test, n_test), dtype=torch.float32, device=dev)
    for _ in range(5):
        base.refresh("delta")
        base.influence_rows(obs, obs, 1e-4, "delta", out=out)
    torch.cuda.synchronize()
    h = _lib.lib()
    nb = min(n_test, 4096)
    buf = np.zeros(nb * 4 * 8, dtype=np.uint64)
    h.lt_debug_df_trace.argtypes = [C.c_void_p, C.c_int]
    h.lt_debug_df_trace.restype = C.c_int
    assert h.lt_debug_df_trace(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(nb, 4, 8).astype(np.float64) * 0.01      # 100 MHz -> us
    t0 = t[:, :, 0].min()
    names = ["entry", "stage A done", "barrier passed", "short positions done", "long position done", "end"]
    for k, nm in enumerate(names):
        x_ = (t[:, :, k] - t0).ravel()
        print(f"{nm:22s} first {x_.min():6.2f}  median {np.median(x_):6.2f}  p90 {np.percentile(x_, 90):6.2f}  last {x_.max():6.2f}")
    d = t[:, :, 5] - t[:, :, 0]
    print("per wave, entry -> answered: median %.2f  p90 %.2f  max %.2f" % (np.median(d), np.percentile(d, 90), d.max()))
    for k in range(1, 6):
        d = t[:, :, k] - t[:, :, k - 1]
        print(f"  phase {k}: median {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f}")


if __name__ == "__main__":
    main()
