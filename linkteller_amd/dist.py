"""Probe-dimension sharding: row i of the influence matrix depends on probe
i only, so ranks take contiguous slices of the probe list and one all-gather of the row slabs
(RCCL over xGMI when the backend is ``nccl``) rebuilds the matrix.  No other collective exists on
the path.  Works on CPU tensors with ``gloo`` too (that is how the N>1 path is tested without GPUs).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def force_collectives() -> bool:
    """``LT_FORCE_COLLECTIVES=1`` with an initialised process group: every collective of the path (the all-gather of row
    slabs, the sharded refresh's all-gather of X W1, the policy's all-reduce) is issued even at world size 1.  A one-GPU box
    then executes the very calls an 8-GPU run makes -- through RCCL when the backend is ``nccl`` -- instead of the
    single-process shortcuts (tests/test_gpu_round4.py asserts librccl is mapped)."""
    return os.environ.get("LT_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized()


def collectives_on() -> bool:
    """True when the N > 1 code path is the one to run: several ranks, or the forcing hook above."""
    return world()[1] > 1 or force_collectives()


def shard_bounds(n_items: int, rank: int, world_size: int):
    """[begin, end) of this rank's slice; every rank gets ceil(n/W) slots, the tail is padding."""
    per = (n_items + world_size - 1) // world_size
    b = min(rank * per, n_items)
    return b, min(b + per, n_items), per


def all_gather_rows(local_rows: torch.Tensor, n_total: int, async_op: bool = False):
    """local_rows: this rank's [end-begin, n_obs] slab (may be empty), or the full padded
    [ceil(n/W), n_obs] slab (then it is sent as is and must not be overwritten before the collective
    has completed).  Returns [n_total, n_obs] on every rank.  One ``all_gather_into_tensor`` on a padded contiguous slab.
    ``async_op=True`` returns ``(tensor, work)``: the collective runs on the communicator's stream and
    the caller's stream is not made to wait (``work.wait()`` before reading the tensor)."""
    rank, ws = world()
    if not collectives_on():
        return (local_rows, None) if async_op else local_rows
    _, _, per = shard_bounds(n_total, rank, ws)
    n_obs = local_rows.shape[1]
    if local_rows.shape[0] == per and local_rows.is_contiguous():
        slab = local_rows                  # already the padded slab: no staging copy
    else:
        slab = torch.zeros((per, n_obs), dtype=local_rows.dtype, device=local_rows.device)
        slab[: local_rows.shape[0]] = local_rows
    if slab.is_cuda and dist.get_backend() == "gloo":
        # gloo has no all-gather for device tensors (test hook: several ranks on one GPU): stage through the host
        full_h = torch.empty((ws * per, n_obs), dtype=slab.dtype)
        dist.all_gather_into_tensor(full_h, slab.cpu())
        full = full_h.to(slab.device)[:n_total]
        return (full, None) if async_op else full
    full = torch.empty((ws * per, n_obs), dtype=local_rows.dtype, device=local_rows.device)
    if async_op:
        work = dist.all_gather_into_tensor(full, slab, async_op=True)
        return full[:n_total], work
    dist.all_gather_into_tensor(full, slab)
    return full[:n_total]


def all_gather_into(full: torch.Tensor, shard: torch.Tensor):
    """``full[world * k, ...] <- all ranks' shard[k, ...]`` on the current stream order (one all_gather_into_tensor)."""
    if shard.is_cuda and dist.get_backend() == "gloo":      # test hook, see all_gather_rows
        full_h = torch.empty(full.shape, dtype=full.dtype)
        dist.all_gather_into_tensor(full_h, shard.cpu())
        full.copy_(full_h)
        return
    dist.all_gather_into_tensor(full, shard)


def shard_probes_policy() -> str:
    """LT_SHARD_PROBES: '1' every call shards its probes over the ranks and all-gathers the row slabs, '0' every rank builds
    the whole matrix itself (no collective), 'auto' (default) times both once per workload and keeps the faster -- the same
    branch on every rank.  A build whose loop-invariant part dominates (n_test = 500 in `delta`: one pass over X + the
    pre-activation, repeated by every rank whatever its share of the probes) gains less from a rank's shorter probe list
    than the all-gather costs: sharded, it would be SLOWER than one GPU."""
    v = os.environ.get("LT_SHARD_PROBES", "auto").lower()
    return v if v in ("0", "1", "auto") else "auto"


_probe_choice = {}


def choose_probe_sharding(key, sharded, local, trials: int = 5, warm: int = 2, sync=None) -> bool:
    """Whether the workload `key` shards its probes (True) or every rank builds all rows (False), by shard_probes_policy().
    `sharded` / `local` run one build each way (the callers' real step: refresh + rows (+ all-gather)); 'auto' times
    `trials` of each after `warm`, takes the MAX over the ranks (one all-reduce per strategy) and keeps the faster -- every
    rank gets the same answer, remembered per key.  `sync`: drains the device (default torch.cuda.synchronize when CUDA is
    initialised).  The pattern of _choose_baseline_sharding."""
    import time
    if not collectives_on():
        return False
    pol = shard_probes_policy()
    if pol != "auto":
        return pol == "1"
    if key in _probe_choice:
        return _probe_choice[key][0]
    if sync is None:
        sync = torch.cuda.synchronize if (torch.cuda.is_available() and torch.cuda.is_initialized()) else (lambda: None)
    times = []
    for fn in (local, sharded):
        for _ in range(warm):
            fn()
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(trials):
            fn()
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if dist.get_backend() != "gloo":
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times.append(float(t.item()) / trials)
    use = times[1] < times[0]
    _probe_choice[key] = (use, times[0], times[1])
    return use


def probe_sharding_report(key):
    """(sharded?, seconds per build local, seconds per build sharded) of a decided workload, or None."""
    return _probe_choice.get(key)


HUB_ROW_MIN_ENTRIES = int(os.environ.get("LT_HUB_ROW_MIN_ENTRIES", "1024"))   # rows at least this long are worth an exchange (2 KB of fp64
                                                                             # per row against >= 1 MB of gathers); the env is a test hook


class SharedHubRows:
    """The hub rows ALL ranks' probes reach, split over the ranks (BASELINE configs[4]: R-MAT, n_test = 4096 over 8 ranks).

    On the on-demand route of `delta` (engine.Baseline.fp64_route() == 2) a rank forms the fp64 pre-activation on the rows its
    own probes reach; on a heavy-tailed graph most of that work lies in hub rows every rank reaches (84 % of a rank's gathers
    in ~3 800 rows at configs[4]) -- replicated W times.  Every rank holds the whole probe list (attacker.py:220: one list), so
    all of them compute the same sorted list of reached rows of >= HUB_ROW_MIN_ENTRIES entries ONCE per node list (here), and
    after every refresh (``exchange``) rank k forms rows [k per, (k + 1) per) of it, one all-gather moves the 2-KB rows, every
    rank adopts all of them; the probe calls that follow skip valid rows.  A row's bits do not depend on the rank that formed it
    (its gathers and its product are functions of the row alone), so the matrix equals the single-rank one bit for bit."""

    def __init__(self, base, all_probe_nodes, min_entries: int = HUB_ROW_MIN_ENTRIES):
        rank, ws = world()
        self.base = base
        self.rows = base.reached_rows(all_probe_nodes, min_entries)
        n = self.rows.numel()
        self.n_rows = n
        self.b, self.e, self.per = shard_bounds(n, rank, ws)
        dev = self.rows.device
        hp = (base.h + 3) // 4 * 4
        self.mine = self.rows[self.b:self.e].contiguous()
        padded = torch.full((ws * self.per,), -1, dtype=torch.int32, device=dev)      # (ids out of range are skipped by the scatter)
        padded[:n] = self.rows
        self.padded = padded
        self.send = torch.zeros((max(self.per, 1), hp), dtype=torch.float64, device=dev)
        self.recv = torch.empty((max(ws * self.per, 1), hp), dtype=torch.float64, device=dev)

    def exchange(self):
        """After a refresh, before the probe call: form this rank's share, all-gather, adopt every rank's rows."""
        if self.n_rows == 0:
            return
        if self.mine.numel():
            self.base.form_rows_fp64(self.mine)
            self.base.gather_rows_fp64(self.mine, self.send)
        if collectives_on():
            all_gather_into(self.recv, self.send)
            self.base.scatter_rows_fp64(self.padded, self.recv)

    @property
    def collective_bytes(self):
        return int(self.recv.numel() * 8)


def shard_baseline_policy() -> str:
    """LT_SHARD_BASELINE: '0' replicate X W1 on every rank, '1' shard it + all-gather S1, 'auto' (default) time both
    at first use and keep the faster (the product is small at twitch size: whether the all-gather beats recomputing
    depends on the link latency, not on arithmetic)."""
    import os
    v = os.environ.get("LT_SHARD_BASELINE", "auto").lower()
    return v if v in ("0", "1", "auto") else "auto"


def choose_baseline_sharding(base, trials: int = 5, mode: str = "full") -> bool:
    use = _choose_baseline_sharding(base, trials, mode)
    if collectives_on():
        base.refresh(mode)          # whatever was decided: the products the mode reads are current on every rank
    return use


def _choose_baseline_sharding(base, trials, mode) -> bool:
    """Apply shard_baseline_policy() to an engine.Baseline FOR THE MODE THE CALLER WILL USE; every rank takes the same
    decision (max over ranks of the measured time of a refresh + the first reader of that mode, replicated vs sharded).
    `full` / `sparse` read the fp32 product S1 = X W1 (sharded: lt_baseline_refresh_rows + all-gather of S1); `delta` with
    the fp64 pre-activation reads nothing fp32, so what is sharded there is the fp64 product (lt_baseline_refresh_rows_fp64
    + all-gather of S1d) -- unless the features are sparse differences to a reference row (engine.Baseline.fp64_route()
    == 1: the product is one pass over X on every rank, cheaper than any collective).  Returns True when a sharded refresh
    is on."""
    import time
    rank, ws = world()
    multi = collectives_on()
    pol = shard_baseline_policy()
    is_delta = mode == "delta"
    if not getattr(base, "supports_sharding", True):
        # engine.WideBaseline (layers wider than one pass of the fused kernels): its product is a per-slice loop with no
        # sharded refresh, and the timing loop below would call modes it may not serve -- replicated on every rank
        return False
    shard = base.shard_refresh_fp64 if is_delta else base.shard_refresh
    other = base.shard_refresh if is_delta else base.shard_refresh_fp64
    other(False)
    if is_delta and multi:
        base.enable_fp64()
        # The route depends on a word of mapped host memory that k_s1d_feature_rows sets asynchronously (rows denser than
        # its probe saw): read it only once the stream has drained, and let every rank take the SAME branch (the smallest
        # route any rank reports: 0, the matrix-core product, is always valid) -- ranks that disagree here would sit in
        # different collectives below.
        if base.x.is_cuda:
            torch.cuda.current_stream(base.x.device).synchronize()
        route = torch.tensor([base.fp64_route()], dtype=torch.int32)
        if dist.get_backend() != "gloo":
            route = route.to(base.x.device)
        dist.all_reduce(route, op=dist.ReduceOp.MIN)
        if int(route.item()) in (1, 2):
            # 1: the fp64 product is one pass over X; 2: the pre-activation is formed on the rows a rank's own probes reach
            # (aggregate-first, no S1d at all) -- nothing worth exchanging either way
            shard(False)
            return False
    if not multi or pol == "0":
        shard(False)
        return False
    if pol == "1":
        shard(True)
        return True
    times = []
    nodes = torch.zeros(1, dtype=torch.int32, device=base.x.device)

    def refresh_and_use():
        # the replicated refresh is lazy (it only marks things stale): the first reader makes them materialise -- the
        # logits for the fp32 layers, a one-pair call for the fp64 pre-activation -- and costs the same in both settings
        base.refresh(mode)
        if is_delta:
            base.influence_rows(nodes, nodes, 1e-4, "delta")
        else:
            base.logits()
    for enable in (False, True):
        shard(enable)
        for _ in range(2):
            refresh_and_use()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(trials):
            refresh_and_use()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if dist.get_backend() != "gloo":
            t = t.to(base.x.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times.append(float(t.item()))
    use = times[1] < times[0]
    shard(use)
    return use
