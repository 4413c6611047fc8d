"""Dev: dumps the HIP influence matrices of golden fixtures (all modes, probe_kslice settings) for offline analysis."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from test_gpu_parity import _setup
from linkteller_amd import _lib
g = load_golden("influence.npz")
out = {}
for key in ("er300", "pl600", "pl600hi", "lap600", "rand400"):
    for pk in (0, 256):
        _lib.set_tuning("probe_kslice", pk)
        args, base = _setup(g, key, torch.device("cuda:0"))
        nodes = g[f"{key}.ref32.test_nodes"]
        for m in ("full", "delta"):
            out[f"{key}.pk{pk}.{m}"] = base.influence_rows(nodes, nodes, args["influence"], m).cpu().numpy()
np.savez_compressed("gpurun_out/dump_case.npz", **out)
print("ok")
