"""world_size-2 gloo test of the probe sharding + all-gather (the only collective on the path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from linkteller_amd import dist as lt_dist


def test_shard_bounds_cover_and_pad():
    for n in (0, 1, 7, 500, 2000, 4096):
        for ws in (1, 2, 3, 8):
            spans = [lt_dist.shard_bounds(n, r, ws) for r in range(ws)]
            assert [s[2] for s in spans] == [spans[0][2]] * ws
            covered = [i for b, e, _ in spans for i in range(b, e)]
            assert covered == list(range(n))
            assert all(e - b <= per for b, e, per in spans)


def _worker(rank, ws, port, n, n_obs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    full = torch.arange(n * n_obs, dtype=torch.float32).reshape(n, n_obs) * 0.5 + 1.0   # "influence rows"
    b, e, _ = lt_dist.shard_bounds(n, rank, ws)
    got = lt_dist.all_gather_rows(full[b:e].clone(), n)
    got2, work = lt_dist.all_gather_rows(full[b:e].clone() * 2, n, async_op=True)     # pipelined form used by bench.py
    work.wait()
    q.put((rank, bool(torch.equal(got, full)) and bool(torch.equal(got2, full * 2)), tuple(got.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,n_obs", [(500, 500), (7, 5), (1, 3)])
def test_all_gather_rows_two_ranks(n, n_obs):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, n_obs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and all(r[2] == (n, n_obs) for r in res)


def test_single_process_is_identity():
    x = torch.ones(3, 4)
    assert lt_dist.world() == (0, 1)
    assert lt_dist.all_gather_rows(x, 3) is x
    y, work = lt_dist.all_gather_rows(x, 3, async_op=True)
    assert y is x and work is None


def _cli_worker(rank, ws, port, tmp, q):
    """What `torchrun -m linkteller_amd.main` does per rank before the attack: join the group (gloo hook), then the
    result file is written by rank 0 alone (attacker.compute_and_save)."""
    import argparse
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws),
                      LOCAL_RANK=str(rank), LT_DIST_BACKEND="gloo")
    from linkteller_amd import main as lt_main
    from linkteller_amd.attacker import Attacker
    assert lt_main.init_distributed() is True
    assert lt_main.init_distributed() is False                  # already initialised: the caller keeps ownership
    ok = lt_dist.world() == (rank, ws)
    os.chdir(tmp)
    atk = Attacker.__new__(Attacker)                            # only the result writer is exercised here
    atk.args = argparse.Namespace(mode="vanilla-clean", attack_mode="efficient", sample_type="unbalanced", n_test=4,
                                  sample_seed=42)
    atk.dataset = "twitch/ES/RU"
    atk.compute_and_save([0.9, 0.8, 0.4], [0.1, 0.5, 0.0, 0.0])
    dist.barrier()
    q.put((rank, ok, os.path.exists(os.path.join(tmp, atk.result_filename()))))
    dist.barrier()
    dist.destroy_process_group()


def test_cli_ranks_join_group_and_rank0_writes(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    dirs = [str(tmp_path / f"r{r}") for r in range(2)]          # one working directory per rank: who wrote is visible
    for d in dirs:
        os.makedirs(d)
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, dirs[r], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, False)]


def _policy_worker(rank, ws, port, q):
    """choose_probe_sharding under gloo: both ranks take the same branch (MAX over ranks of the measured times), the choice is
    remembered per key, LT_SHARD_PROBES pins it."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    calls = {"s": 0, "l": 0}

    def sharded():          # fast on rank 0, slow on rank 1: the MAX over ranks must decide
        calls["s"] += 1
        time.sleep(0.001 if rank == 0 else 0.02)

    def local():
        calls["l"] += 1
        time.sleep(0.006)
    os.environ.pop("LT_SHARD_PROBES", None)
    a = lt_dist.choose_probe_sharding("w1", sharded, local, trials=3, warm=1)
    n_after = dict(calls)
    b = lt_dist.choose_probe_sharding("w1", sharded, local, trials=3, warm=1)      # remembered: no further timing
    rep = lt_dist.probe_sharding_report("w1")
    c = lt_dist.choose_probe_sharding("w2", lambda: time.sleep(0.001), lambda: time.sleep(0.01), trials=3, warm=1)
    os.environ["LT_SHARD_PROBES"] = "1"
    d = lt_dist.choose_probe_sharding("w1", sharded, local)
    os.environ["LT_SHARD_PROBES"] = "0"
    e = lt_dist.choose_probe_sharding("w2", sharded, local)
    q.put((rank, a, b, c, d, e, n_after == calls, rep is not None and rep[1] < rep[2]))
    dist.barrier()
    dist.destroy_process_group()


def test_probe_sharding_policy_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_policy_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # w1: sharded is slower on the slowest rank -> every rank builds all rows; w2: sharded faster; the pins override
    assert res == [(0, False, False, True, True, False, True, True), (1, False, False, True, True, False, True, True)]


def test_probe_sharding_policy_single_process():
    assert lt_dist.choose_probe_sharding("x", lambda: 1 / 0, lambda: 1 / 0) is False      # no group: nothing is timed


class _FakeBase:
    """What dist.SharedHubRows needs of an engine.Baseline, on CPU tensors: Z[r] = r * (1, 2, ..., Hp) for a formed row."""
    def __init__(self, n, h, reached):
        self.n, self.h = n, h
        self._reached = torch.tensor(sorted(reached), dtype=torch.int32)
        self.z = torch.zeros((n, (h + 3) // 4 * 4), dtype=torch.float64)
        self.valid = torch.zeros(n, dtype=torch.bool)
        self.formed = []

    def reached_rows(self, nodes, min_entries):
        return self._reached.clone()

    def form_rows_fp64(self, rows):
        for r in rows.tolist():
            self.z[r] = float(r) * torch.arange(1, self.z.shape[1] + 1, dtype=torch.float64)
            self.valid[r] = True
            self.formed.append(r)

    def gather_rows_fp64(self, rows, dst):
        dst[: rows.numel()] = self.z[rows.long()]

    def scatter_rows_fp64(self, rows, src):
        for i, r in enumerate(rows.tolist()):
            if 0 <= r < self.n:
                self.z[r] = src[i]
                self.valid[r] = True


def _hub_worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    ok = True
    for reached in ([3, 17, 4, 90, 55, 21, 8], [5], list(range(10, 26)), []):
        base = _FakeBase(100, 6, reached)
        hub = lt_dist.SharedHubRows(base, None, min_entries=1)
        hub.exchange()
        want = torch.zeros_like(base.z)
        for r in reached:
            want[r] = float(r) * torch.arange(1, want.shape[1] + 1, dtype=torch.float64)
        b, e, per = lt_dist.shard_bounds(len(reached), rank, ws)
        ok = ok and torch.equal(base.z, want) and sorted(base.formed) == sorted(reached)[b:e]      # only its share was formed here
        ok = ok and bool(base.valid[torch.tensor(sorted(reached), dtype=torch.long)].all()) and int(base.valid.sum()) == len(reached)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_shared_hub_rows_partition_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hub_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
