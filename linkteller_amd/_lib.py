"""ctypes binding of ``liblinkteller_hip.so`` (the C ABI declared in include/linkteller_hip.h).

There is deliberately no CPU fallback: every compute entry point of this package goes through
this library, and loading fails loudly if it has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C linkteller_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblinkteller_hip.so")

LT_OK = 0
MODE_FULL, MODE_SPARSE, MODE_DELTA = 0, 1, 2
MODES = {"full": MODE_FULL, "sparse": MODE_SPARSE, "delta": MODE_DELTA}


class LinkTellerHipError(RuntimeError):
    """A negative lt_status came back across the C ABI (mapped to RuntimeError)."""


_lib = None

# name -> (restype, argtypes); must list every symbol include/linkteller_hip.h declares
SIGNATURES = {
    "lt_last_error": (C.c_char_p, []),
    "lt_abi_version": (C.c_int, []),
    "lt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "lt_set_tuning": (C.c_int, [C.c_char_p, C.c_longlong]),
    "lt_graph_create": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_void_p)]),
    "lt_graph_destroy": (C.c_int, [C.c_void_p]),
    "lt_graph_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                C.POINTER(C.c_int32)]),
    "lt_graph_records_host": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.POINTER(C.c_int64)]),
    "lt_gemm_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                              C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "lt_spmm_csr_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32,
                                  C.c_void_p, C.c_int64, C.c_void_p]),
    "lt_spmm_route": (C.c_int, [C.c_void_p, C.c_int32]),
    "lt_spmm_gather_ceiling_bytes": (C.c_size_t, [C.c_void_p]),
    "lt_spmm_gather_ceiling": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p,
                                         C.c_int64, C.c_void_p]),
    "lt_gcn2_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "lt_gcn2_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "lt_baseline_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.POINTER(C.c_void_p)]),
    "lt_baseline_refresh": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lt_baseline_attach_s1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "lt_baseline_refresh_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "lt_baseline_enable_fp64": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lt_baseline_attach_s1d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "lt_baseline_refresh_rows_fp64": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "lt_baseline_fp64_route": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "lt_baseline_destroy": (C.c_int, [C.c_void_p]),
    "lt_baseline_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "lt_influence_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "lt_influence_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float,
                                    C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lt_graph_reached_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lt_baseline_form_rows_fp64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "lt_baseline_gather_rows_fp64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "lt_baseline_scatter_rows_fp64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "lt_influence_rows_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float,
                                        C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lt_influence_rows_vec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float,
                                        C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lt_wide_combine": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int32,
                                  C.c_int32, C.c_void_p]),
    "lt_baseline3_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                      C.POINTER(C.c_void_p)]),
    "lt_baseline3_refresh": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lt_baseline3_destroy": (C.c_int, [C.c_void_p]),
    "lt_baseline3_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "lt_influence3_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "lt_influence3_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_void_p,
                                     C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lt_lapgraph_select": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_double), C.c_void_p]),
    "lt_baseline3_enable_fp64": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lt_influence3_rows_mode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lt_node_check": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lt_export_rows_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "lt_profile_calls": (C.c_int, [C.POINTER(C.c_int64)]),
    "lt_profile_enable": (C.c_int, [C.c_int]),
    "lt_profile_reset": (C.c_int, []),
    "lt_profile_summary": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}
KERNEL_IDS = {"gemm": 0, "layer1": 1, "layer2": 2, "perturb": 3, "full_stageA": 4, "full_stageB": 5,
              "item_stageA": 6, "item_stageB": 7, "spmm": 8, "fp64_product": 9, "fp64_spmm": 10, "item_bits": 11}
ABI_VERSION = 5


def lib():
    """The loaded library (loads on first use; raises if the shared object is missing)."""
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64.so.7 / libhsa-runtime64.so.1; whichever copy is loaded
        # first serves the whole process.  Load torch's first so that this library, torch's
        # allocator and torch's streams all talk to ONE HIP runtime.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise LinkTellerHipError(
                f"{LIB_PATH} not found: the HIP extension is not built.  Build it with "
                f"`make -C {os.path.join(_HERE, 'csrc')}` (needs hipcc); there is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


LT_ERR_INDEX = -6


def check(status: int, what: str = ""):
    if status != LT_OK:
        msg = lib().lt_last_error().decode("utf-8", "replace")
        if status == LT_ERR_INDEX:      # the reference raises IndexError at grad_mat[test_nodes[j]] (attacker.py:226-229)
            raise IndexError(f"{what or 'liblinkteller_hip'}: {msg}")
        raise LinkTellerHipError(f"{what or 'liblinkteller_hip'} failed with status {status}: {msg}")


TUNING_DEFAULT = -(1 << 63)


def set_tuning(key: str, value=None):
    """Route-selection knob of the library (include/linkteller_hip.h); ``None`` restores the default."""
    check(lib().lt_set_tuning(key.encode(), TUNING_DEFAULT if value is None else int(value)), "lt_set_tuning")


def device_count() -> int:
    n = C.c_int(0)
    check(lib().lt_device_count(C.byref(n)), "lt_device_count")
    return n.value


def require_gpu():
    """Fail loudly when there is nothing to run on (no silent eager/CPU path)."""
    import torch
    if not torch.cuda.is_available() or device_count() == 0:
        raise LinkTellerHipError("no HIP device visible: linkteller_amd runs its hot path only on an "
                                 "MI355X-class GPU through liblinkteller_hip.so (no CPU fallback)")
