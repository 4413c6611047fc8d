"""The C-ABI library loads on a CPU-only box and exports exactly what include/linkteller_hip.h
declares; argument validation that happens before any device call is checked here too."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def lt():
    from linkteller_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def declared_functions():
    src = open(os.path.join(REPO, "include", "linkteller_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lt_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lt):
    names = declared_functions()
    assert len(names) >= 19
    handle = lt.lib()
    for n in names:
        assert hasattr(handle, n), f"{n} declared in the header but not exported"
        assert n in lt.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(lt.SIGNATURES) == names
    assert handle.lt_abi_version() == lt.ABI_VERSION == 5


def test_no_gpu_means_loud_failure_not_fallback(lt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lt.device_count() == 0
    from linkteller_amd import engine, graph
    import scipy.sparse as sp
    with pytest.raises(lt.LinkTellerHipError):
        graph.HipGraph(sp.identity(4, format="csr"))
    with pytest.raises(lt.LinkTellerHipError):
        engine.gemm(torch.zeros(2, 2), torch.zeros(2, 2))


def test_graph_create_validation(lt):
    h = lt.lib()
    out = C.c_void_p()

    def create(n, rowptr, col, val):
        rp = np.asarray(rowptr, dtype=np.int32)
        ci = np.asarray(col, dtype=np.int32)
        va = np.asarray(val, dtype=np.float32)
        return h.lt_graph_create(n, len(ci), rp.ctypes.data, ci.ctypes.data if len(ci) else None,
                                 va.ctypes.data if len(va) else None, C.byref(out))

    assert create(2, [1, 1, 2], [0, 1], [1, 1]) == -1 and b"rowptr[0]" in h.lt_last_error()
    assert create(2, [0, 2, 1], [0], [1]) == -1
    assert create(2, [0, 1, 2], [0, 5], [1, 1]) == -1 and b"out of range" in h.lt_last_error()
    assert create(2, [0, 2, 2], [1, 0], [1, 1]) == -1 and b"strictly increasing" in h.lt_last_error()
    assert create(2, [0, 2, 2], [1, 1], [1, 1]) == -1     # duplicate column
    assert h.lt_graph_create(2, 0, None, None, None, C.byref(out)) == -1
    # a graph large enough for the multi-threaded validation (nnz >= 2^20): offsets that are negative / beyond nnz /
    # decreasing in the part of rowptr a LATER thread starts in must be refused before any col[] is read through them
    n, deg = 1 << 17, 8
    rp = (np.arange(n + 1, dtype=np.int64) * deg).astype(np.int32)
    ci = np.tile(np.arange(deg, dtype=np.int32), n)
    va = np.ones(n * deg, dtype=np.float32)
    for row, bad in ((n - 5, -7), (n // 2 + 3, 2 ** 31 - 1), (3 * n // 4, 11)):
        rp2 = rp.copy()
        rp2[row] = bad
        assert h.lt_graph_create(n, len(ci), rp2.ctypes.data, ci.ctypes.data, va.ctypes.data, C.byref(out)) == -1
        assert b"monotone" in h.lt_last_error() or b"rowptr" in h.lt_last_error()
    assert h.lt_graph_info(None, None, None, None) == -1
    assert h.lt_graph_destroy(None) == 0


def test_workspace_queries_and_argument_errors(lt):
    h = lt.lib()
    assert h.lt_gcn2_workspace_bytes(4385, 3170, 256, 2) >= 4385 * 256 * 4 + 4385 * 2 * 4
    assert h.lt_gcn2_workspace_bytes(-1, 8, 256, 2) == 0
    assert h.lt_influence_workspace_bytes(None, 10, 10, 0) == 0
    assert h.lt_spmm_csr_f32(None, None, 0, 4, None, 0, None, 0, None) == -1
    assert h.lt_influence_rows(None, None, 1, None, 1, 1e-4, 0, None, 1, None, 0, None) == -1
    assert h.lt_gemm_f32(None, 1, None, 1, None, 1, 1, 1, 1, None) == -1
    tot, cnt = C.c_double(), C.c_int64()
    assert h.lt_profile_summary(99, C.byref(tot), C.byref(cnt)) == -1
    assert h.lt_profile_enable(0) == 0 and h.lt_profile_reset() == 0
    assert h.lt_profile_summary(0, C.byref(tot), C.byref(cnt)) == 0 and cnt.value == 0


def test_every_documented_tuning_key_is_accepted(lt):
    """The tuning keys listed in the header comment of lt_set_tuning are the keys the library knows (host-side state
    only: callable without a GPU); unknown keys and out-of-range values are refused with LT_ERR_INVALID."""
    src = open(os.path.join(REPO, "include", "linkteller_hip.h")).read()
    block = src[src.index("tuning knobs"):src.index("#define LT_TUNING_DEFAULT")]
    keys = re.findall(r'^ \*\s+"([a-z0-9_]+)"', block, flags=re.M)
    assert len(keys) >= 12 and len(set(keys)) == len(keys)
    for k in keys:
        lt.set_tuning(k, None)                      # restoring the default is always valid
    # every key the library parses is documented
    core = open(os.path.join(REPO, "linkteller_amd", "csrc", "lt_core.hip")).read()
    parsed = set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', core))
    assert parsed == set(keys), parsed ^ set(keys)
    h = lt.lib()
    assert h.lt_set_tuning(b"no_such_knob", 1) == -1 and b"unknown key" in h.lt_last_error()
    assert h.lt_set_tuning(b"full_p", 12) == -1
    assert h.lt_set_tuning(b"chunk_budget_bytes", 0) == -1
    assert h.lt_set_tuning(None, 1) == -1
    for k in keys:
        lt.set_tuning(k, None)


def test_header_is_plain_c_and_links_from_c(lt, tmp_path):
    """The boundary is a C ABI: include/linkteller_hip.h compiles as C99 with -pedantic (and as C++), and a C program
    that binds a few entry points links against the shared object and runs without a GPU (host-only calls)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "c_caller.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "linkteller_hip.h"\n'
                   'int main(void) {\n'
                   '    int n = -1;\n'
                   '    if (lt_abi_version() != LT_ABI_VERSION) return 1;\n'
                   '    if (lt_device_count(&n) != LT_OK || n < 0) return 2;\n'
                   '    if (lt_set_tuning("full_p", 12) != LT_ERR_INVALID) return 3;\n'
                   '    if (strstr(lt_last_error(), "full_p") == NULL) return 4;\n'
                   '    if (lt_set_tuning("full_p", LT_TUNING_DEFAULT) != LT_OK) return 5;\n'
                   '    lt_graph *g = NULL;\n'
                   '    if (lt_graph_create(-1, 0, NULL, NULL, NULL, &g) != LT_ERR_INVALID || g != NULL) return 6;\n'
                   '    printf("ok %d\\n", n);\n    return 0;\n}\n')
    inc = os.path.join(REPO, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-fsyntax-only", str(src)], check=True)
    if shutil.which("g++"):
        subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)], check=True)
    exe = tmp_path / "c_caller"
    libdir = os.path.dirname(lt.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-llinkteller_hip",
                        f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("ok"), (run.returncode, run.stdout, run.stderr[-500:])
