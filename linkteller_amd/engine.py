"""Thin torch-facing wrappers over the C ABI: torch owns device memory and streams, the kernels
are liblinkteller_hip's.  Every function enqueues on torch's current stream and never syncs."""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib
from .graph import HipGraph, as_hip_graph


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current stream (of the current device) as the ``void *`` the C ABI takes.  The raw getter is what torch's own
    launchers use: ~0.3 us against ~3 us for building a ``torch.cuda.Stream`` object -- three of them sat in front of the first
    launch of every ``Attacker.influence_matrix()`` call (tools/api_profile.py)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise _lib.LinkTellerHipError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _require_finite(**tensors):
    """ReLU is one ``v_max_f32`` in the kernels: a NaN pre-activation becomes 0 where torch keeps NaN (DESIGN.md
    section 3), so non-finite inputs would be hidden instead of propagated.  They are refused up front (one device
    reduction per tensor when a baseline is created, never on the per-step path)."""
    for name, t in tensors.items():
        if not bool(torch.isfinite(t).all()):
            raise ValueError(f"{name} holds non-finite values (NaN / Inf): the reference would propagate them through "
                             f"every score; refusing rather than masking them")


def _workspace(nbytes: int, device) -> torch.Tensor:
    # torch's caching allocator returns >= 512-byte aligned blocks
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def gemm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a @ b on the matrix cores in exact fp32 (reference gcn/layers.py:31 torch.mm)."""
    a, b = _f32(a, "a"), _f32(b, "b")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[0]:
        raise ValueError(f"shape mismatch {tuple(a.shape)} @ {tuple(b.shape)}")
    m, k = a.shape
    n = b.shape[1]
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().lt_gemm_f32(a.data_ptr(), k, b.data_ptr(), n, out.data_ptr(), n, m, n, k, _stream()),
               "lt_gemm_f32")
    return out


def spmm(adj, dense: torch.Tensor, bias=None, relu: bool = False) -> torch.Tensor:
    """adj @ dense (+ bias)(relu)  (reference gcn/layers.py:32-36 torch.spmm + bias)."""
    g = as_hip_graph(adj)
    s = _f32(dense, "dense")
    if s.dim() != 2 or s.shape[0] != g.n:
        raise ValueError(f"dense must be [{g.n}, k], got {tuple(s.shape)}")
    k = s.shape[1]
    out = torch.empty((g.n, k), dtype=torch.float32, device=s.device)
    bptr = None
    if bias is not None:
        bias = _f32(bias, "bias")
        bptr = bias.data_ptr()
    _lib.check(_lib.lib().lt_spmm_csr_f32(g.handle, s.data_ptr(), k, k, bptr, int(bool(relu)),
                                          out.data_ptr(), k, _stream()), "lt_spmm_csr_f32")
    return out


def gcn2_forward(adj, x, w1, b1, w2, b2) -> torch.Tensor:
    """Logits of the 2-layer GCN in eval mode (reference gcn/models.py:19-24)."""
    g = as_hip_graph(adj)
    x, w1, b1, w2, b2 = (_f32(t, n) for t, n in ((x, "x"), (w1, "W1"), (b1, "b1"), (w2, "W2"), (b2, "b2")))
    n, f = x.shape
    h, c = w1.shape[1], w2.shape[1]
    if n != g.n or w1.shape[0] != f or w2.shape[0] != h or b1.numel() != h or b2.numel() != c:
        raise ValueError("inconsistent GCN shapes")
    if not fused_shapes(h, c):      # wider than the fused kernels: the same layers unfused (GraphConvolution.forward x 2)
        return spmm(g, gemm(spmm(g, gemm(x, w1), b1, relu=True), w2), b2)
    out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    nbytes = _lib.lib().lt_gcn2_workspace_bytes(n, f, h, c)
    ws = _workspace(nbytes, x.device)
    _lib.check(_lib.lib().lt_gcn2_forward(g.handle, x.data_ptr(), f, f, w1.data_ptr(), b1.data_ptr(), h,
                                          w2.data_ptr(), b2.data_ptr(), c, out.data_ptr(), c,
                                          ws.data_ptr(), ws.numel(), _stream()), "lt_gcn2_forward")
    return out


class Baseline:
    """Unperturbed forward state (S1, Z1, S2, logits) for the probe loop; see lt_baseline_create."""

    def __init__(self, adj, x, w1, b1, w2, b2):
        self.graph: HipGraph = as_hip_graph(adj)
        self.x, self.w1, self.b1, self.w2, self.b2 = (
            _f32(t, n) for t, n in ((x, "x"), (w1, "W1"), (b1, "b1"), (w2, "W2"), (b2, "b2")))
        n, f = self.x.shape
        self.n, self.f, self.h, self.c = n, f, self.w1.shape[1], self.w2.shape[1]
        if n != self.graph.n or self.w1.shape[0] != f or self.w2.shape[0] != self.h:
            raise ValueError("inconsistent GCN shapes")
        if self.graph.device_index != self.x.device.index:
            raise ValueError(f"the graph lives on cuda:{self.graph.device_index}, the features on {self.x.device}")
        _require_finite(features=self.x, W1=self.w1, b1=self.b1, W2=self.w2, b2=self.b2)
        h = C.c_void_p()
        _lib.check(_lib.lib().lt_baseline_create(self.graph.handle, self.x.data_ptr(), f, f, self.w1.data_ptr(),
                                                 self.b1.data_ptr(), self.h, self.w2.data_ptr(),
                                                 self.b2.data_ptr(), self.c, _stream(), C.byref(h)),
                   "lt_baseline_create")
        self._h = h
        self._finalizer = weakref.finalize(self, _lib.lib().lt_baseline_destroy, h)
        self._ws = {}
        self._host_scratch = {}
        self._fp64 = False
        self._shard = None          # (row_begin, row_end, per) while the sharded refresh of the fp32 product is on
        self._s1_full = None        # torch-owned S1 storage once attached (kept alive for the handle's lifetime)
        self._send = None
        self._shard64 = None        # the same for the fp64 product of `delta` (lt_baseline_refresh_rows_fp64)
        self._s1d_full = None
        self._send64 = None

    @property
    def handle(self):
        return self._h

    def refresh(self, mode=None):
        """The borrowed inputs changed: everything derived from them is recomputed by the next call that reads it.
        ``mode`` (the mode of the calls that follow, when known) lets a multi-GPU run move only what that mode reads:
        with ``shard_refresh`` on, `full` / `sparse` compute this rank's rows of the fp32 X W1 and all-gather S1; `delta`
        with the fp64 pre-activation does the same for the fp64 product (``shard_refresh_fp64``) and touches nothing
        fp32.  Without a mode both sharded products (if enabled) are refreshed."""
        m = None if mode is None else (_lib.MODES[mode] if isinstance(mode, str) else int(mode))
        only64 = m == _lib.MODE_DELTA and self._fp64
        only32 = m is not None and not only64
        from . import dist as lt_dist
        if self._shard is not None and not only64:
            b, e, _ = self._shard
            _lib.check(_lib.lib().lt_baseline_refresh_rows(self._h, b, e, self._send.data_ptr(), _stream()),
                       "lt_baseline_refresh_rows")
            lt_dist.all_gather_into(self._s1_full, self._send)
        else:
            _lib.check(_lib.lib().lt_baseline_refresh(self._h, _stream()), "lt_baseline_refresh")
        if self._shard64 is not None and not only32:
            b, e, _ = self._shard64
            _lib.check(_lib.lib().lt_baseline_refresh_rows_fp64(self._h, b, e, self._send64.data_ptr(), _stream()),
                       "lt_baseline_refresh_rows_fp64")
            lt_dist.all_gather_into(self._s1d_full, self._send64)

    def enable_fp64(self):
        """The fp64-accumulated pre-activation `delta` evaluates its ReLU kink test on (lt_baseline_enable_fp64):
        allocates S1d / Z1d (2 * n * Hp * 8 bytes -- 8.6 GB at n = 2 M, H = 256) and computes them once."""
        if not self._fp64:
            _lib.check(_lib.lib().lt_baseline_enable_fp64(self._h, _stream()), "lt_baseline_enable_fp64")
            self._fp64 = True
        return self

    def fp64_route(self) -> int:
        """1: the fp64 product comes from the feature rows' differences to a reference row (one pass over X; sharding it
        buys nothing), 0: it runs on the f64 matrix cores, -1: fp64 not enabled."""
        r = C.c_int32(-1)
        _lib.check(_lib.lib().lt_baseline_fp64_route(self._h, C.byref(r)), "lt_baseline_fp64_route")
        return r.value

    # ---- the on-demand pre-activation by row list (fp64_route() == 2): hub rows shared between ranks, dist.share_hub_rows ----
    def reached_rows(self, probe_nodes, min_entries: int) -> torch.Tensor:
        """Sorted int32 device list of the rows of >= min_entries entries that the probes reach in one hop (lt_graph_reached_rows;
        one host sync for the count: call once per node list, not per step)."""
        dev = self.x.device
        probes = _as_nodes(probe_nodes, self.n, dev, "probe_nodes")
        flags = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)
        rows = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().lt_graph_reached_rows(self.graph.handle, probes.data_ptr(), probes.numel(), int(min_entries),
                                                    flags.data_ptr(), rows.data_ptr(), count.data_ptr(), _stream()),
                   "lt_graph_reached_rows")
        return torch.sort(rows[: int(count.item())]).values.contiguous()

    def form_rows_fp64(self, rows: torch.Tensor):
        self.enable_fp64()
        _lib.check(_lib.lib().lt_baseline_form_rows_fp64(self._h, rows.data_ptr(), rows.numel(), _stream()), "lt_baseline_form_rows_fp64")

    def gather_rows_fp64(self, rows: torch.Tensor, dst: torch.Tensor):
        _lib.check(_lib.lib().lt_baseline_gather_rows_fp64(self._h, rows.data_ptr(), rows.numel(), dst.data_ptr(), _stream()),
                   "lt_baseline_gather_rows_fp64")

    def scatter_rows_fp64(self, rows: torch.Tensor, src: torch.Tensor):
        _lib.check(_lib.lib().lt_baseline_scatter_rows_fp64(self._h, rows.data_ptr(), rows.numel(), src.data_ptr(), _stream()),
                   "lt_baseline_scatter_rows_fp64")

    def shard_refresh_fp64(self, enable=True):
        """Multi-GPU: shard the fp64 product X W1 of `delta` over the ranks (rows of k_gemm_f64acc_128 + one all-gather
        of S1d) instead of recomputing it on every rank.  Bits are those of the replicated product."""
        from . import dist as lt_dist
        rank, world = lt_dist.world()
        if not enable or not lt_dist.collectives_on():
            if self._shard64 is not None:
                _lib.check(_lib.lib().lt_baseline_attach_s1d(self._h, None, 0, _stream()), "lt_baseline_attach_s1d")
                torch.cuda.current_stream().synchronize()
                self._s1d_full = self._send64 = None
            self._shard64 = None
            return self
        self.enable_fp64()
        b, e, per = lt_dist.shard_bounds(self.n, rank, world)
        hp = (self.h + 3) // 4 * 4
        if self._s1d_full is None or self._s1d_full.shape[0] != world * per:
            s1d = torch.zeros((world * per, hp), dtype=torch.float64, device=self.x.device)
            _lib.check(_lib.lib().lt_baseline_attach_s1d(self._h, s1d.data_ptr(), hp, _stream()), "lt_baseline_attach_s1d")
            self._s1d_full = s1d
            self._send64 = torch.zeros((per, hp), dtype=torch.float64, device=self.x.device)
        self._shard64 = (b, e, per)
        return self

    def shard_refresh(self, enable=True):
        """Multi-GPU: shard the loop-invariant X W1 over the ranks of the initialised process group
        instead of replicating it.  S1 moves to a torch-owned [world * ceil(n / world), Hp] tensor (the all-gather's
        output) that the library reads in place.  Bits are those of the replicated product."""
        from . import dist as lt_dist
        rank, world = lt_dist.world()
        if not enable or not lt_dist.collectives_on():
            self._shard = None
            return self
        b, e, per = lt_dist.shard_bounds(self.n, rank, world)
        hp = (self.h + 3) // 4 * 4
        if self._s1_full is None or self._s1_full.shape[0] != world * per:
            s1_full = torch.zeros((world * per, hp), dtype=torch.float32, device=self.x.device)
            _lib.check(_lib.lib().lt_baseline_attach_s1(self._h, s1_full.data_ptr(), hp, _stream()), "lt_baseline_attach_s1")
            torch.cuda.current_stream().synchronize()        # the previous storage may be released now
            self._s1_full = s1_full
            self._send = torch.zeros((per, hp), dtype=torch.float32, device=self.x.device)
        self._shard = (b, e, per)
        return self

    def logits(self) -> torch.Tensor:
        out = torch.empty((self.n, self.c), dtype=torch.float32, device=self.x.device)
        _lib.check(_lib.lib().lt_baseline_logits(self._h, out.data_ptr(), _stream()), "lt_baseline_logits")
        return out

    def influence_rows(self, probe_nodes, observe_nodes, delta: float, mode="full", out=None, host=None) -> torch.Tensor:
        """[n_probe, n_obs] fp32 on the device: ||(f(X + delta e_v x_v^T) - f(X))[u]||_2 / delta.  ``host``: a pinned (or
        device) float64 [n_probe, n_obs] tensor that receives the same matrix widened (valid once the stream has drained).
        Node lists given as int32 CUDA tensors are checked on the device: call ``engine.node_check()`` after synchronising
        (see ``_as_nodes``)."""
        dev = self.x.device
        probes = _as_nodes(probe_nodes, self.n, dev, "probe_nodes")
        obs = _as_nodes(observe_nodes, self.n, dev, "observe_nodes")
        m = _lib.MODES[mode] if isinstance(mode, str) else int(mode)
        if m == _lib.MODE_DELTA and not self._fp64:
            # kink test on an fp64-accumulated pre-activation (one-off cost, kept fresh by refresh())
            self.enable_fp64()
        npb, nob = probes.numel(), obs.numel()
        if out is None:
            out = torch.empty((npb, nob), dtype=torch.float32, device=dev)
        elif out.shape != (npb, nob) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 [n_probe, n_obs] tensor")
        key = (npb, nob, m)
        need = _lib.lib().lt_influence_workspace_bytes(self._h, npb, nob, m)   # host arithmetic; depends on the tuning knobs too
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = _workspace(need, dev)
            self._ws = {key: ws}   # keep only the latest: sizes repeat across steps
        if host is not None:
            # lt_influence_rows_f64: the float64 matrix is written as part of the same call (the fused `delta` route's blocks
            # export their own rows; other routes end with the export launch)
            _lib.check(_lib.lib().lt_influence_rows_f64(self._h, probes.data_ptr(), npb, obs.data_ptr(), nob, float(delta), m,
                                                        out.data_ptr(), nob, host.data_ptr(), max(nob, 1), ws.data_ptr(),
                                                        ws.numel(), _stream()), "lt_influence_rows_f64")
            return out
        _lib.check(_lib.lib().lt_influence_rows(self._h, probes.data_ptr(), npb, obs.data_ptr(), nob,
                                                float(delta), m, out.data_ptr(), nob, ws.data_ptr(),
                                                ws.numel(), _stream()), "lt_influence_rows")
        return out

    def influence_matrix_host(self, probe_nodes, observe_nodes, delta: float, mode="delta", refresh=False):
        """[n_probe, n_obs] float64 on the HOST (the reference's ``influence_val``, attacker.py:216-229) by ONE library call:
        the rows are formed and land in pinned host memory (a block of torch's pinned-memory cache, owned by the returned
        array) without an export launch of their own where the route allows; one stream wait; then the device-side node-id
        check (IndexError, as the reference raises).  ``refresh``: mark the loop-invariant baseline stale first (``refresh(mode)``)
        -- the call then recomputes it, and on the fused `delta` route the matrix's zeros cross PCIe under the launch that forms
        the product rows while the probes' blocks send the touched positions only."""
        dev = self.x.device
        probes = _as_nodes(probe_nodes, self.n, dev, "probe_nodes")
        obs = _as_nodes(observe_nodes, self.n, dev, "observe_nodes")
        npb, nob = probes.numel(), obs.numel()
        host = torch.empty((npb, nob), dtype=torch.float64, pin_memory=True)
        key = ("scratch", npb, nob)
        out = self._host_scratch.get(key)
        if out is None:
            out = torch.empty((npb, nob), dtype=torch.float32, device=dev)
            self._host_scratch = {key: out}
        if refresh:
            self.refresh(mode)
        if npb and nob:
            self.influence_rows(probes, obs, delta, mode, out=out, host=host)
            torch.cuda.current_stream(dev).synchronize()
            node_check()
        return host.numpy()


    def influence_rows_vec(self, probes: torch.Tensor, obs: torch.Tensor, delta: float, m: int, out: torch.Tensor,
                           vec: torch.Tensor) -> torch.Tensor:
        """lt_influence_rows_vec (a slice of a wide model, ``WideBaseline``): the norms into ``out`` and the pairs' unscaled
        difference vectors into ``vec`` ([n_probe, n_obs, C] dense in the storage of ``vec``).  Device int32 node tensors."""
        if m == _lib.MODE_DELTA and not self._fp64:
            self.enable_fp64()
        npb, nob = probes.numel(), obs.numel()
        if vec.numel() < npb * nob * self.c or vec.dtype != torch.float32 or not vec.is_contiguous():
            raise ValueError("vec must be a contiguous float32 tensor of at least n_probe * n_obs * C elements")
        key = (npb, nob, m)
        need = _lib.lib().lt_influence_workspace_bytes(self._h, npb, nob, m)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = _workspace(need, self.x.device)
            self._ws = {key: ws}
        _lib.check(_lib.lib().lt_influence_rows_vec(self._h, probes.data_ptr(), npb, obs.data_ptr(), nob, float(delta), m,
                                                    out.data_ptr(), nob, vec.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
                   "lt_influence_rows_vec")
        return vec


class WideBaseline:
    """The probe primitive for 2-layer models wider than one pass of the fused kernels (hidden width > 256 or more than 8
    classes; the reference has no such limit: gcn/layers.py:14-36, main.py:30 --hidden).  Same interface as ``Baseline``.

    A perturbation reaches the logits as a SUM over the hidden units, so the model is served slice by slice: for every slice
    s of <= 256 hidden units and every slice t of <= 8 classes a ``Baseline`` on (W1[:, s], b1[s], W2[s, t], b2[t]) whose
    ``lt_influence_rows_vec`` call yields the pairs' difference vectors, and per class slice one ``lt_wide_combine`` that adds
    the hidden slices' vectors in slice order, divides by delta and accumulates the squared norm over the classes in class
    order (include/linkteller_hip.h).  All three modes: `sparse` (= `full`: per slice the fp32 finite difference of the
    slice's own forward, exact zeros outside the 2-hop set) and `delta` (the exact propagation, fp64 kink test per slice).
    ~5 launches per (s, t) per matrix, independent of the number of probes (round 3 looped ~5 launches per PROBE and had
    no `delta`).  The fp32 product X W1[:, s] is formed once per hidden slice and shared by its class slices (round 6); the
    fp64 parts of `delta` are per (s, t)."""

    supports_sharding = False       # dist.choose_baseline_sharding: replicated on every rank, no timing loop
    VEC_BUDGET_BYTES = 256 << 20    # difference vectors of one probe chunk, all hidden slices together

    def __init__(self, adj, x, w1, b1, w2, b2):
        self.graph: HipGraph = as_hip_graph(adj)
        self.x, self.w1, self.b1, self.w2, self.b2 = (
            _f32(t, n) for t, n in ((x, "x"), (w1, "W1"), (b1, "b1"), (w2, "W2"), (b2, "b2")))
        n, f = self.x.shape
        self.n, self.f, self.h, self.c = n, f, self.w1.shape[1], self.w2.shape[1]
        if n != self.graph.n or self.w1.shape[0] != f or self.w2.shape[0] != self.h or self.b1.numel() != self.h or self.b2.numel() != self.c:
            raise ValueError("inconsistent GCN shapes")
        if self.graph.device_index != self.x.device.index:
            raise ValueError(f"the graph lives on cuda:{self.graph.device_index}, the features on {self.x.device}")
        _require_finite(features=self.x, W1=self.w1, b1=self.b1, W2=self.w2, b2=self.b2)
        self.h_slices = [(s0, min(s0 + 256, self.h)) for s0 in range(0, self.h, 256)]
        self.c_slices = [(t0, min(t0 + 8, self.c)) for t0 in range(0, self.c, 8)]
        if len(self.h_slices) > 32:
            raise NotImplementedError(f"hidden width {self.h}: more than 32 slices of 256 (lt_wide_combine)")
        self._subs = None           # [(s, t)] -> (Baseline, (W1 slice, b1 slice, W2 slice, b2 slice)); built on first use
        self._logits = None
        self._fp64 = False
        self._buf = {}

    def _build(self):
        if self._subs is not None:
            return
        subs = {}
        for si, (s0, s1) in enumerate(self.h_slices):
            for ti, (t0, t1) in enumerate(self.c_slices):
                parts = (self.w1[:, s0:s1].contiguous(), self.b1[s0:s1].contiguous(),
                         self.w2[s0:s1, t0:t1].contiguous(), self.b2[t0:t1].contiguous())
                subs[(si, ti)] = (Baseline(self.graph, self.x, *parts), parts)
        self._subs = subs
        # ONE product X W1[:, s] per hidden slice, shared by the slice's class slices (a 121-class model has 16 of them): every
        # (s, t) baseline reads S1 from the slice's torch-owned storage (lt_baseline_attach_s1), the first class slice's
        # baseline forms it there (lt_baseline_refresh_rows over all rows), the others are told it is current
        self._s1 = {}
        for si, (s0, s1) in enumerate(self.h_slices):
            hp = (s1 - s0 + 3) // 4 * 4
            t = torch.zeros((self.n, hp), dtype=torch.float32, device=self.x.device)
            self._s1[si] = t
            for ti in range(len(self.c_slices)):
                sub = subs[(si, ti)][0]
                # (marked current first: attaching copies the baseline's own S1 over, and none has been formed yet)
                _lib.check(_lib.lib().lt_baseline_refresh_rows(sub._h, 0, 0, t.data_ptr(), _stream()), "lt_baseline_refresh_rows")
                _lib.check(_lib.lib().lt_baseline_attach_s1(sub._h, t.data_ptr(), hp, _stream()), "lt_baseline_attach_s1")
                sub._s1_full = t
        self._share_products()

    def _share_products(self):
        for si in range(len(self.h_slices)):
            t = self._s1[si]
            for ti in range(len(self.c_slices)):
                sub = self._subs[(si, ti)][0]
                _lib.check(_lib.lib().lt_baseline_refresh_rows(sub._h, 0, self.n if ti == 0 else 0, t.data_ptr(), _stream()),
                           "lt_baseline_refresh_rows")

    def refresh(self, mode=None):
        """The borrowed inputs changed: the slices are re-cut from them, everything derived is recomputed on next use."""
        self._logits = None
        if self._subs is None:
            return
        for (si, ti), (sub, (w1s, b1s, w2s, b2s)) in self._subs.items():
            (s0, s1), (t0, t1) = self.h_slices[si], self.c_slices[ti]
            w1s.copy_(self.w1[:, s0:s1]); b1s.copy_(self.b1[s0:s1])
            w2s.copy_(self.w2[s0:s1, t0:t1]); b2s.copy_(self.b2[t0:t1])
        m = _lib.MODES[mode] if isinstance(mode, str) else mode
        if m == _lib.MODE_DELTA and all(sub._fp64 for sub, _ in self._subs.values()):
            # `delta` reads the fp64 pre-activation only: no fp32 X W1[:, s] at all (75 us per hidden slice at twitch size, half of a
            # 0.34 ms build with H = 512) -- every slice is marked stale, and an fp32 reader that comes later recomputes lazily
            for sub, _ in self._subs.values():
                _lib.check(_lib.lib().lt_baseline_refresh(sub._h, _stream()), "lt_baseline_refresh")
        else:
            self._share_products()      # (marks every slice's layers and fp64 parts stale as lt_baseline_refresh does)

    def shard_refresh(self, enable=True):
        return self

    shard_refresh_fp64 = shard_refresh

    def enable_fp64(self):
        self._fp64 = True
        return self

    def fp64_route(self) -> int:
        return -1

    def logits(self) -> torch.Tensor:
        """GraphConvolution.forward x 2 on the unfused HIP layers (gcn/layers.py:30-36: any width)."""
        if self._logits is None:
            h1 = spmm(self.graph, gemm(self.x, self.w1), self.b1, relu=True)
            self._logits = spmm(self.graph, gemm(h1, self.w2), self.b2)
        return self._logits.clone()

    def influence_rows(self, probe_nodes, observe_nodes, delta: float, mode="delta", out=None) -> torch.Tensor:
        m = _lib.MODES[mode or "delta"] if isinstance(mode, str) or mode is None else int(mode)
        if m == _lib.MODE_FULL:
            m = _lib.MODE_SPARSE            # the same quantity, bit for bit (lt_influence_rows_vec has no FULL form)
        dev = self.x.device
        probes = _as_nodes(probe_nodes, self.n, dev, "probe_nodes")
        obs = _as_nodes(observe_nodes, self.n, dev, "observe_nodes")
        npb, nob = probes.numel(), obs.numel()
        if out is None:
            out = torch.empty((npb, nob), dtype=torch.float32, device=dev)
        elif out.shape != (npb, nob) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 [n_probe, n_obs] tensor")
        if npb == 0 or nob == 0:
            return out
        self._build()
        n_s = len(self.h_slices)
        # probes in chunks: the slices' difference vectors ([chunk, n_obs, 8] fp32 per hidden slice) stay within a fixed budget
        # (unchunked they took n_slices x n_probe x n_obs x 32 bytes -- 16 GB at 4096^2 pairs and 32 slices)
        chunk = max(1, min(npb, self.VEC_BUDGET_BYTES // (n_s * nob * 8 * 4)))
        key = (chunk, nob)
        if self._buf.get("key") != key:
            self._buf = {"key": key, "norm": torch.empty((chunk, nob), dtype=torch.float32, device=dev),
                         "vec": [torch.empty((chunk, nob, 8), dtype=torch.float32, device=dev) for _ in self.h_slices]}
        ptrs = (C.c_void_p * n_s)()
        for p0 in range(0, npb, chunk):
            pc = probes[p0:p0 + chunk]
            nb = pc.numel()
            orow = out[p0:p0 + nb]
            for ti, (t0, t1) in enumerate(self.c_slices):
                ct = t1 - t0
                for si in range(n_s):
                    sub = self._subs[(si, ti)][0]
                    if m == _lib.MODE_DELTA:
                        sub.enable_fp64()
                    vec = self._buf["vec"][si]          # used as [nb, nob, ct] dense
                    sub.influence_rows_vec(pc, obs, delta, m, self._buf["norm"], vec)
                    ptrs[si] = vec.data_ptr()
                _lib.check(_lib.lib().lt_wide_combine(ptrs, n_s, nb * nob, ct, float(delta), orow.data_ptr(), int(ti == 0),
                                                      int(ti == len(self.c_slices) - 1), _stream()), "lt_wide_combine")
        return out


def fused_shapes(h: int, c: int) -> bool:
    """What lt_baseline_create / lt_influence_rows are built for (one pass of the row kernels, fused layer-2 epilogue)."""
    return h <= 256 and c <= 8


def baseline_for(adj, x, w1, b1, w2, b2):
    """``Baseline`` (the fused probe primitive) when the layer widths allow it, else ``WideBaseline``."""
    return (Baseline if fused_shapes(w1.shape[1], w2.shape[1]) else WideBaseline)(adj, x, w1, b1, w2, b2)


class Baseline3:
    """Unperturbed forward state of the 3-layer model (reference gcn/models.py:28-46) + its probe primitive
    (lt_baseline3_create / lt_influence3_rows): the reference's fp32 finite difference, evaluated on the rows a
    probe reaches in 1 / 2 / 3 hops."""

    def __init__(self, adj, x, w1, b1, w2, b2, w3, b3):
        self.graph: HipGraph = as_hip_graph(adj)
        self.x, self.w1, self.b1, self.w2, self.b2, self.w3, self.b3 = (
            _f32(t, n) for t, n in ((x, "x"), (w1, "W1"), (b1, "b1"), (w2, "W2"), (b2, "b2"), (w3, "W3"), (b3, "b3")))
        n, f = self.x.shape
        self.n, self.f = n, f
        self.h1, self.h2, self.c = self.w1.shape[1], self.w2.shape[1], self.w3.shape[1]
        if (n != self.graph.n or self.w1.shape[0] != f or self.w2.shape[0] != self.h1 or self.w3.shape[0] != self.h2
                or self.b1.numel() != self.h1 or self.b2.numel() != self.h2 or self.b3.numel() != self.c):
            raise ValueError("inconsistent GCN3 shapes")
        if self.graph.device_index != self.x.device.index:
            raise ValueError(f"the graph lives on cuda:{self.graph.device_index}, the features on {self.x.device}")
        _require_finite(features=self.x, W1=self.w1, b1=self.b1, W2=self.w2, b2=self.b2, W3=self.w3, b3=self.b3)
        h = C.c_void_p()
        _lib.check(_lib.lib().lt_baseline3_create(self.graph.handle, self.x.data_ptr(), f, f, self.w1.data_ptr(),
                                                  self.b1.data_ptr(), self.h1, self.w2.data_ptr(), self.b2.data_ptr(),
                                                  self.h2, self.w3.data_ptr(), self.b3.data_ptr(), self.c, _stream(),
                                                  C.byref(h)), "lt_baseline3_create")
        self._h = h
        self._finalizer = weakref.finalize(self, _lib.lib().lt_baseline3_destroy, h)
        self._ws = {}
        self._fp64 = False

    def refresh(self):
        _lib.check(_lib.lib().lt_baseline3_refresh(self._h, _stream()), "lt_baseline3_refresh")

    def logits(self) -> torch.Tensor:
        out = torch.empty((self.n, self.c), dtype=torch.float32, device=self.x.device)
        _lib.check(_lib.lib().lt_baseline3_logits(self._h, out.data_ptr(), _stream()), "lt_baseline3_logits")
        return out

    def enable_fp64(self):
        """fp64 pre-activations of the first two layers for the exact (`delta`) propagation (lt_baseline3_enable_fp64)."""
        if not self._fp64:
            _lib.check(_lib.lib().lt_baseline3_enable_fp64(self._h, _stream()), "lt_baseline3_enable_fp64")
            self._fp64 = True
        return self

    def influence_rows(self, probe_nodes, observe_nodes, delta: float, mode="sparse", out=None) -> torch.Tensor:
        """`sparse` (= `full`): the reference's fp32 finite difference on the rows a probe reaches in 1 / 2 / 3 hops;
        `delta`: the perturbation propagated exactly through the three layers (lt_influence3_rows_mode)."""
        m = _lib.MODES[mode or "sparse"] if isinstance(mode, str) or mode is None else int(mode)
        if m == _lib.MODE_DELTA:
            self.enable_fp64()
        dev = self.x.device
        probes = _as_nodes(probe_nodes, self.n, dev, "probe_nodes")
        obs = _as_nodes(observe_nodes, self.n, dev, "observe_nodes")
        npb, nob = probes.numel(), obs.numel()
        if out is None:
            out = torch.empty((npb, nob), dtype=torch.float32, device=dev)
        elif out.shape != (npb, nob) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 [n_probe, n_obs] tensor")
        key = (npb, nob)
        need = _lib.lib().lt_influence3_workspace_bytes(self._h, npb, nob)   # depends on the tuning knobs too
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = _workspace(need, dev)
            self._ws = {key: ws}
        _lib.check(_lib.lib().lt_influence3_rows_mode(self._h, probes.data_ptr(), npb, obs.data_ptr(), nob, float(delta), m,
                                                      out.data_ptr(), nob, ws.data_ptr(), ws.numel(), _stream()),
                   "lt_influence3_rows_mode")
        return out


def export_rows_f64(rows: torch.Tensor):
    """The finished rows as a float64 numpy array on the host (the reference's ``influence_val = np.zeros(...)`` filled by
    n_test**2 ``.item()`` round trips, attacker.py:216-229) -- ONE launch that widens on the device and writes straight into
    pinned host memory over PCIe (lt_export_rows_f64), one wait.  The array owns a block of torch's pinned-memory cache
    (recycled when the array is released: a fresh result per call costs no allocation).  Ends with the device-side node-id
    check of everything the stream has run (IndexError, as the reference raises)."""
    if rows.dim() != 2 or rows.dtype != torch.float32 or not rows.is_cuda:
        raise TypeError("rows must be a 2-d float32 CUDA tensor")
    r, c = rows.shape
    host = torch.empty((r, c), dtype=torch.float64, pin_memory=True)
    if rows.stride(1) != 1:
        rows = rows.contiguous()
    stream = torch.cuda.current_stream(rows.device)
    _lib.check(_lib.lib().lt_export_rows_f64(rows.data_ptr(), rows.stride(0) if r > 1 else max(c, 1), r, c, host.data_ptr(), max(c, 1),
                                             C.c_void_p(stream.cuda_stream)), "lt_export_rows_f64")
    stream.synchronize()
    node_check()
    return host.numpy()


def node_check():
    """Raise IndexError when a probe / observed list of a COMPLETED call held a node id outside [0, n) (lt_node_check: the
    lists are device memory, so the kernels check them; call after synchronising)."""
    _lib.check(_lib.lib().lt_node_check(None, None), "lt_node_check")


def _as_nodes(nodes, n, device, name) -> torch.Tensor:
    """Node list -> int32 device tensor.  Host lists are range-checked here (IndexError, as the reference's indexing raises).  An
    int32 CUDA tensor is taken as it is -- no copy, no host round trip -- and its ids are checked by the KERNELS of the call that
    reads it: an id outside [0, n) is replaced by node 0 there and raises a flag the caller collects with ``engine.node_check()``
    AFTER synchronising (``influence_matrix_host`` / ``export_rows_f64`` do).  The flag is one per process (two words of mapped host
    memory): a caller that never collects it gets LT_ERR_INDEX from the NEXT probe call of any baseline instead."""
    if isinstance(nodes, torch.Tensor) and nodes.is_cuda and nodes.dtype == torch.int32:
        return nodes.contiguous()          # fast path: the ids are checked on the device (lt_node_check / LT_ERR_INDEX)
    t = torch.as_tensor(nodes).to(torch.int64).reshape(-1).cpu()
    if t.numel() and (int(t.min()) < 0 or int(t.max()) >= n):
        raise IndexError(f"{name} out of range [0, {n})")
    return t.to(torch.int32).to(device)
