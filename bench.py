#!/usr/bin/env python3
"""bench.py -- influence-matrix build throughput (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 runs in this process.  N > 1: when the torchrun environment (RANK / WORLD_SIZE) is present this process is one rank
of N (``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``); otherwise ``python bench.py --gpus N``
starts N fresh child processes itself -- BEFORE anything here touches the GPU -- one per device, RCCL (``nccl``) over
127.0.0.1, and exits with their status.

One *step* = one complete influence-matrix build for the workload: the loop-invariant baseline forward of the mode, then
every probe and the n_test x n_test influence norms, plus -- for N > 1 -- the single all-gather of row slabs, ENDING with
``influence_val`` on the host as float64 (``lt_influence_rows_f64``): the region the reference itself times
(attacker.py:213->231; round 6 -- up to round 5 the step of ``value`` ended on the device: that figure stays in the line as
``value_device``).  Nothing is cached across steps.  Probes are sharded over ranks; the problem (n_test = 500 on the twitch-RU-shaped
graph) is fixed, so this is STRONG scaling, as the BASELINE.json metric ("n_test=500 at 1/2/4/8 GPU") is.  Inputs are
synthetic (no dataset on the box) and resident in HBM before timing.

Timing: W warm-up steps, then ``--blocks`` (default 5) timed blocks of EXACTLY K steps each, every block bracketed by a
barrier + torch.cuda.synchronize() on both sides and reduced with MAX over ranks; ``value`` / ``ms_per_step`` are the
MEDIAN block ("median of >= 5"), all block times are in the line.

Next to ``value`` the line carries ``api_wall``: the drop-in API itself -- ``Attacker.influence_matrix()`` on the same workload
(the same region plus the Python surface: state_dict walk, cached node lists) -- ``value_host``, ``d2h_us``,
``ms_per_step_export_launch`` (round 5's host path: an export launch behind the step).

``value`` is measured in ``--mode delta`` (default): the mode whose scores, AUC and AP meet north_star's 1e-4 against the
reference.  `full` (every probe a full perturbed forward, the reference's fp32 finite difference)
and `sparse` (bit-identical to it) are reported next to it under ``other_modes`` and never substituted for it; the line
carries a ``parity`` object (GPU rows against the CPU reference path run in fp32 and fp64 in this same process) and the
process exits non-zero when that check fails.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 matrix peak
# v_mfma_f64_16x16x4_f64 issues in 64 cycles (measured: tools/fold_test/mfma_peak.hip) = 32 FLOP/clk/SIMD, half the F32
# row of the guide's matrix-core table: 32 x 4 SIMD x 256 CU x 2.4 GHz
FP64_MFMA_PEAK_TFLOPS = 78.6
INT8_MFMA_PEAK_TOPS = 5000.0   # MI355X_MICROARCH.md: I8 MFMA = 2x the BF16 rate per clock (BF16 ~2.5 PF dense)


def dense_product_roofline(rows_x, f, h, n, us, traffic=None):
    """Roofline object of the dense-feature fp64 product at `us` microseconds per refresh: on the int8 matrix cores as an error-free
    split ("i8_split", default where the shapes allow -- 14 digit-pair products of 2*rows*F*H integer ops each) or on the f64 cores."""
    sec = us * 1e-6 if us else None
    flop = 2.0 * rows_x * f * h
    i8 = os.environ.get("LT_I8_SPLIT", "1") != "0" and n >= 256 and 256 <= f < 32768 and h >= 64 and h % 64 == 0
    if i8:
        ops = 14.0 * flop
        return {"kernel": "k_gemm_i8split (+ k_i8_w_max, k_i8_w_digits, k_sum_slabs_f64_q)", "bound": "mfma",
                "achieved": round(ops / sec / 1e12, 1) if sec else None, "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                "frac": round(ops / sec / 1e12 / INT8_MFMA_PEAK_TOPS, 4) if sec else None, "traffic": traffic, "avg_launch_us": us,
                "units_per_launch": f"X[{rows_x}x{f}] * W1[{f}x{h}] as 14 int8 digit-pair products (v_mfma_i32_32x32x32_i8), exact integer sums, "
                                    f"one rounding per K slice; the time includes W1's digits and the slab sum",
                "algorithmic_flop_per_launch": flop, "fp64_equivalent_TFLOPs": round(flop / sec / 1e12, 2) if sec else None,
                "f64_mfma_peak_TFLOPs": FP64_MFMA_PEAK_TFLOPS}
    return {"kernel": "k_gemm_f64acc_128 (+ k_sum_slabs_f64)", "bound": "mfma", "achieved": round(flop / sec / 1e12, 2) if sec else None,
            "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flop / sec / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4) if sec else None,
            "traffic": traffic, "avg_launch_us": us,
            "units_per_launch": f"X[{rows_x}x{f}] * W1[{f}x{h}], fp32 operands, fp64 accumulation (v_mfma_f64_16x16x4_f64)",
            "algorithmic_flop_per_launch": flop}

# kernel name (prefix) -> profile class, for the PMC passes
KERNEL_CLASS = (("k_s1d_feature_rows", "fp64_product"), ("k_ref_product", "fp64_product"), ("k_ref_vector", "fp64_product"), ("k_gemm_f64", "fp64_product"),
                ("k_sum_slabs_f64", "fp64_product"), ("k_spmm_f64", "fp64_spmm"), ("k_rows_tiled_f64", "fp64_spmm"),
                ("k_rows_tiled_xf64", "fp64_spmm"), ("k_z_mark", "fp64_spmm"), ("k_y_long", "fp64_spmm"),
                ("k_item_bits", "item_bits"), ("k_pm_", "item_bits"), ("k_delta_records", "item_bits"), ("k_item_stageA", "item_stageA"),
                ("k_delta_probe_finish", "item_stageB"), ("k_rows_tiled_gathers_only", "gather_ceiling"),
                ("k_item_stageB", "item_stageB"), ("k_full_stageA", "full_stageA"), ("k_full_long_combine", "full_stageA"),
                ("k_full_stageB", "full_stageB"), ("k_gemm_f32_mfma", "gemm"), ("k_sum_slabs", "gemm"),
                ("k_rows_tiled", "spmm"), ("k_spmm_long_combine", "spmm"), ("k_spmm_rows", "spmm"), ("k_spmm_seg", "spmm"))
PRIMARY = {"fp64_product": ("k_s1d_feature_rows", "k_gemm_f64acc_128", "k_gemm_f64acc", "k_gemm_f64_rows<double>"),
           "fp64_spmm": ("k_spmm_f64<", "k_rows_tiled_f64", "k_rows_tiled_xf64"),
           "item_bits": ("k_item_bits",), "item_stageA": ("k_item_stageA",), "item_stageB": ("k_delta_probe_finish", "k_item_stageB<", "k_item_stageB_rows"),
           "full_stageA": ("k_full_stageA_lds<2, 32, 0>", "k_full_stageA_lds", "k_full_stageA"), "full_stageB": ("k_full_stageB",),
           "gemm": ("k_gemm_f32_mfma_128",), "spmm": ("k_rows_tiled<", "k_spmm_rows")}


def source_signature():
    """sha256 over the kernel sources: PMC traffic figures kept in profiles/ are stamped with it and ignored when stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(REPO, "linkteller_amd", "csrc", "*.hip")) +
                     glob.glob(os.path.join(REPO, "linkteller_amd", "csrc", "*.h"))):
        h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def stamped_traffic(kernel_key):
    """Fallback when the in-run PMC passes are unavailable: profiles/pmc_traffic.json (copied from the in-run passes of a kept bench line), used only
    when it was collected for the kernel sources in the tree."""
    try:
        d = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        if d.get("source_signature") != source_signature():
            return None
        return d.get("kernels", {}).get(kernel_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def flush_c_stdio():
    try:
        C.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps; the median block is reported")
    p.add_argument("--workload", default="twitch-RU", choices=["twitch-RU", "twitch-ES"])
    p.add_argument("--n-test", type=int, default=500)
    p.add_argument("--hidden", type=int, default=256)
    p.add_argument("--classes", type=int, default=2, help="output classes (twitch: 2)")
    p.add_argument("--mode", default="delta", choices=["full", "sparse", "delta"])
    p.add_argument("--powerlaw", action="store_true", help="hub-heavy graph instead of Erdos-Renyi")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU reference sample")
    p.add_argument("--no-extras", action="store_true", help="skip other_modes / standalone SpMM legs")
    p.add_argument("--no-scaling-workloads", action="store_true",
                   help="skip the strong-scaling legs (n_test=2000 and the R-MAT n_test=4096 build at this rank count)")
    p.add_argument("--spmm-scale", type=int, default=21,
                   help="R-MAT scale of the HBM-resident SpMM leg (BASELINE configs[4]: 21; 0 = skip)")
    p.add_argument("--only-spmm", action="store_true", help="run only the R-MAT SpMM leg")
    p.add_argument("--no-api-wall", action="store_true", help="skip the drop-in API leg (Attacker.influence_matrix wall time)")
    p.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes (roofline.traffic)")
    p.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return p.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without torchrun: start the ranks ourselves (nothing in this process has touched the GPU yet)
# ------------------------------------------------------------------------------------------------------------------
def launch_children(a):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for p in procs:
            p.wait()
            rc = rc or p.returncode
            if p.returncode:                 # a rank failed: its peers would wait in a collective for ever
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
    except KeyboardInterrupt:
        for q in procs:
            q.terminate()
        rc = 130
    return rc


# ------------------------------------------------------------------------------------------------------------------
# in-run HBM traffic: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE child runs of this file (MI355X_MICROARCH.md,
# "HBM": counters are KiB; on gfx950 FETCH_SIZE reports half the bytes of wide reads -> 2 * FETCH_SIZE + WRITE_SIZE)
# ------------------------------------------------------------------------------------------------------------------
def pmc_inrun(a):
    import csv
    import glob
    import shutil
    import tempfile
    if a.no_pmc or shutil.which("rocprofv3") is None or any(k.startswith("ROCPROF") for k in os.environ):
        return None, "in-run PMC passes skipped"
    root = tempfile.mkdtemp(prefix="lt_pmc_", dir="/tmp")
    acc = {}
    t0 = time.time()
    try:
        # (third pass, round 5: the L2's own hit / miss counts -- what the SpMM's gather-ceiling argument rests on)
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum"):
            out = os.path.join(root, ctr.replace(" ", "+"))
            cmd = ["rocprofv3", "--pmc", *ctr.split(), "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--mode", a.mode, "--workload", a.workload, "--n-test", str(a.n_test), "--hidden", str(a.hidden),
                   "--classes", str(a.classes), "--spmm-scale", str(0 if a.no_extras else a.spmm_scale)] + (["--powerlaw"] if a.powerlaw else [])
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               timeout=420)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                if ctr.startswith("TCC_"):      # the hit / miss pass is an extra: the traffic figures stand without it
                    continue
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode})"
            for fn in files:
                for row in csv.DictReader(open(fn)):
                    name = row["Kernel_Name"].replace("void ", "").strip()
                    cn = row["Counter_Name"]
                    if cn not in ctr.split():
                        continue
                    d = acc.setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "TCC_HIT_sum": 0.0, "TCC_MISS_sum": 0.0, "n": {}})
                    d[cn] += float(row["Counter_Value"])
                    d["n"][cn] = d["n"].get(cn, 0) + 1
    except Exception as e:      # noqa: BLE001 -- a profiler hiccup must not cost the benchmark line
        return None, f"in-run PMC passes failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(root, ignore_errors=True)
    classes = {}
    for name, d in acc.items():
        cls = next((c for pre, c in KERNEL_CLASS if name.startswith(pre)), None)
        if cls is None:
            continue
        k = classes.setdefault(cls, {"bytes": 0.0, "primary": 0, "kernels": {}, "hit": 0.0, "miss": 0.0, "fetch": 0.0})
        byts = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        k["bytes"] += byts
        k["hit"] += d["TCC_HIT_sum"]
        k["miss"] += d["TCC_MISS_sum"]
        k["fetch"] += 2.0 * d["FETCH_SIZE"] * 1024.0
        nd = max(d["n"].values()) if d["n"] else 0
        k["kernels"][name.split("(")[0][:60]] = {"dispatches": nd, "hbm_bytes_per_dispatch": int(byts / max(nd, 1))}
    for cls, k in classes.items():       # launches of the class = dispatches of its primary kernel
        for pre in PRIMARY.get(cls, ()):
            n = sum(max(d["n"].values()) for name, d in acc.items() if name.startswith(pre) and d["n"])
            if n:
                k["primary"] = n
                break
        k["hbm_bytes_per_launch"] = int(k["bytes"] / k["primary"]) if k["primary"] else None
        k["read_bytes_per_launch"] = int(k["fetch"] / k["primary"]) if k["primary"] else None
        k["l2_hit_frac"] = round(k["hit"] / (k["hit"] + k["miss"]), 4) if (k["hit"] + k["miss"]) > 0 else None
    return classes, f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum + TCC_MISS_sum, separate child runs of this command, {time.time() - t0:.0f} s"


def kernel_ms(name):
    from linkteller_amd import _lib
    tot, cnt = C.c_double(0), C.c_int64(0)
    _lib.check(_lib.lib().lt_profile_summary(_lib.KERNEL_IDS[name], C.byref(tot), C.byref(cnt)))
    return tot.value, cnt.value


def profiled_calls():
    """Calls of the probe primitive sampled by the profile since lt_profile_enable (lt_profile_calls)."""
    from linkteller_amd import _lib
    c = C.c_int64(0)
    _lib.check(_lib.lib().lt_profile_calls(C.byref(c)))
    return c.value


def spmm_bytes(n, nnz, h, elem=4):
    """int32 CSR + fp32 values, S read once, result written once."""
    return nnz * 8 + (n + 1) * 4 + 2 * n * h * elem


def main():
    a = parse()
    if "RANK" not in os.environ and a.gpus > 1 and not a.pmc_child:
        sys.exit(launch_children(a))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and not a.pmc_child:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    # HBM traffic of the kernels this line prices, measured now (rank 0 of a 1-GPU run; child processes, so that this
    # process has not touched the GPU when they start)
    pmc, pmc_note = (None, "N > 1: not collected")
    if world == 1 and not a.pmc_child and not a.only_spmm and os.environ.get("LT_FORCE_COLLECTIVES") != "1":
        pmc, pmc_note = pmc_inrun(a)

    import torch
    import torch.distributed as dist

    # test hooks (used to exercise the N > 1 flow on a 1-GPU box): LT_BENCH_DEVICE pins every rank to one
    # device, LT_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU)
    if os.environ.get("LT_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["LT_BENCH_DEVICE"])
    backend = os.environ.get("LT_BENCH_BACKEND", "nccl")
    if world > 1 and os.environ.get("LT_BENCH_DEVICE") is None and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} HIP devices are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LT_FORCE_COLLECTIVES=1 (linkteller_amd/dist.py): a process group is created even at world size 1 and every collective of
    # the N > 1 path is issued -- on a one-GPU box RCCL then executes the exact calls an 8-GPU run makes
    force = os.environ.get("LT_FORCE_COLLECTIVES") == "1" and not a.pmc_child
    multi = world > 1 or force
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:      # every rank would pick a different port and the rendezvous would hang until the store timeout
                raise SystemExit("WORLD_SIZE > 1 but MASTER_PORT is not set (launch through torchrun or `bench.py --gpus N`)")
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # RCCL prints a version banner through C stdio when its communicator comes up; with stdout redirected that sits in
        # a buffer until the process exits -- i.e. BEHIND the JSON line.  Bring the communicator up now and flush it out, on
        # every rank, so that the line rank 0 prints at the end is the last thing on stdout.
        dist.barrier()
        flush_c_stdio()

    from linkteller_amd import _lib, engine, graph, synth
    from linkteller_amd import dist as lt_dist

    def traffic_of(cls, stamped_key=None):
        if pmc and pmc.get(cls, {}).get("hbm_bytes_per_launch"):
            return pmc[cls]["hbm_bytes_per_launch"]
        return stamped_traffic(stamped_key) if stamped_key else None

    rmat_cache = {}

    def rmat_problem(scale):
        """(A_hat scipy CSR, its device handle, host seconds) of the R-MAT graph of BASELINE configs[4]; built once per process."""
        if scale not in rmat_cache:
            t0 = time.perf_counter()
            big = graph.first_order_gcn(synth.rmat_graph(scale, synth.rmat_draws(scale), seed=42))
            rmat_cache[scale] = (big, graph.HipGraph(big), time.perf_counter() - t0)
        return rmat_cache[scale]

    def rmat_baseline(gb, nb_, hcols):
        """The configs[4] model on the R-MAT graph: F = 256 Gaussian features, H = hcols, C = 2 (kept for the scaling leg)."""
        key = ("baseline", nb_, hcols)
        if key not in rmat_cache:
            xb = torch.from_numpy(synth.gaussian_features(nb_, 256, seed=1)).to(dev)
            wb = synth.gcn_weights(256, hcols, 2, seed=42)
            rmat_cache[key] = engine.Baseline(gb, xb, *[torch.from_numpy(wb[k]).to(dev) for k in ("W1", "b1", "W2", "b2")])
        return rmat_cache[key]

    def influence_shard(gb, nb_, scale, hcols):
        """The influence build on the R-MAT graph at the shape ONE of 8 ranks gets in BASELINE configs[4] (n_test = 4096 ->
        512 probes x 4096 observed nodes, F = H = 256): the product's default mode, with and without the loop-invariant
        baseline (X*W1 and the fp64 pre-activation), and the bit-faithful `sparse`."""
        bb = rmat_baseline(gb, nb_, hcols)
        rs = np.random.RandomState(42)
        ob = rs.choice(nb_, 4096, replace=False)
        pb = ob[:512]
        shard = {"workload": f"R-MAT scale {scale}, F=256 H={hcols}: 512 probes x 4096 observed (BASELINE configs[4], one rank of 8)"}

        def wall(fn, reps_=3):
            fn(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps_):
                fn()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t) / reps_ * 1e3, 3)
        for m_ in ("delta", "sparse"):
            shard[f"{m_}_ms"] = wall(lambda: bb.influence_rows(pb, ob, 1e-4, m_))
            shard[f"{m_}_incl_baseline_ms"] = wall(lambda: (bb.refresh(m_), bb.influence_rows(pb, ob, 1e-4, m_)))
        shard["pairs_per_s_delta_incl_baseline"] = round(512 * 4096 / (shard["delta_incl_baseline_ms"] * 1e-3), 1)
        if bb.fp64_route() == 2:
            # The same rank with the hub rows ALL 4096 probes reach split over 8 ranks (dist.SharedHubRows): this GPU plays rank 0
            # -- it forms 1/8 of those rows and packs them; the all-gather is EXCLUDED (its output, the other ranks' rows, is
            # prepared once outside the timed region); it adopts all of them, then runs its 512 probes.
            rows_all = bb.reached_rows(ob, lt_dist.HUB_ROW_MIN_ENTRIES)
            nh = int(rows_all.numel())
            per8 = (nh + 7) // 8
            hp = (hcols + 3) // 4 * 4
            allbuf = torch.empty((max(nh, 1), hp), dtype=torch.float64, device=dev)
            bb.refresh("delta")
            bb.form_rows_fp64(rows_all)
            bb.gather_rows_fp64(rows_all, allbuf)
            mine = rows_all[:per8].contiguous()
            send = torch.empty((max(per8, 1), hp), dtype=torch.float64, device=dev)

            def hub_step():
                bb.refresh("delta")
                bb.form_rows_fp64(mine)
                bb.gather_rows_fp64(mine, send)
                bb.scatter_rows_fp64(rows_all, allbuf)
                return bb.influence_rows(pb, ob, 1e-4, "delta")
            plain = bb.influence_rows(pb, ob, 1e-4, "delta").clone()
            shard["delta_incl_baseline_hub_rows_shared_ms"] = wall(hub_step)
            shard["hub_rows_shared"] = {
                "rows_all_ranks_reach": nh, "min_entries": lt_dist.HUB_ROW_MIN_ENTRIES, "rows_formed_by_this_rank": int(mine.numel()),
                "all_gather_bytes": int(8 * per8 * hp * 8), "same_bits_as_the_unshared_build": bool(torch.equal(hub_step(), plain)),
                "predicted_speedup_at_8_ranks_before_collectives": None,
                "note": "one rank of 8 emulated on this GPU: 1/8 of the hub rows formed here, the other ranks' rows adopted from a buffer "
                        "filled outside the timed region (the all-gather itself is not in the figure)"}
        return shard

    def tiled_route(gb_, hcols):
        return bool(_lib.lib().lt_spmm_route(gb_.handle, hcols))

    def spmm_rmat_leg(scale, hcols, reps=10, with_shard=True):
        """Standalone SpMM (lt_spmm_csr_f32) on an R-MAT graph whose S exceeds every cache: the 'SpMM HBM GB/s' half of
        the metric, at BASELINE configs[4] size by default.  Kernel time from HIP events on the launch stream."""
        big, gb, host_s = rmat_problem(scale)
        sb = torch.randn((big.shape[0], hcols), device=dev)
        for _ in range(2):
            engine.spmm(gb, sb)
        if a.pmc_child:
            if tiled_route(gb, hcols):      # the gather-ceiling kernel too: its L2 hit fraction next to the real kernel's
                sink = torch.empty(int(_lib.lib().lt_spmm_gather_ceiling_bytes(gb.handle)), dtype=torch.uint8, device=dev)
                for _ in range(2):
                    _lib.check(_lib.lib().lt_spmm_gather_ceiling(gb.handle, sb.data_ptr(), hcols, hcols, 8, sink.data_ptr(), sink.numel(), None,
                                                                 hcols, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lt_spmm_gather_ceiling")
            torch.cuda.synchronize()
            return None
        _lib.lib().lt_profile_enable(1 << _lib.KERNEL_IDS["spmm"])
        for _ in range(reps):
            engine.spmm(gb, sb)
        torch.cuda.synchronize()
        tot, cnt = kernel_ms("spmm")
        _lib.lib().lt_profile_enable(0)
        sec = tot / cnt * 1e-3
        byts = spmm_bytes(big.shape[0], big.nnz, hcols)
        # the gather ceiling (lt_spmm_gather_ceiling): the timed kernel with everything but its gathers removed -- same work
        # items, order, column stream, slice placement, piece size; then the same with the result rows stored as well
        ceiling = None
        if tiled_route(gb, hcols):
            sink = torch.empty(int(_lib.lib().lt_spmm_gather_ceiling_bytes(gb.handle)), dtype=torch.uint8, device=dev)
            outc = torch.empty((big.shape[0], hcols), dtype=torch.float32, device=dev)

            def ceiling_ms(in_flight, with_stores, reps_=5):
                def one():
                    _lib.check(_lib.lib().lt_spmm_gather_ceiling(gb.handle, sb.data_ptr(), hcols, hcols, in_flight, sink.data_ptr(), sink.numel(),
                                                                 outc.data_ptr() if with_stores else None, hcols,
                                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lt_spmm_gather_ceiling")
                one()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps_):
                    one()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / reps_
            g_ms = {u: ceiling_ms(u, False) for u in (8, 16)}
            gs_ms = {u: ceiling_ms(u, True) for u in (8, 16)}
            best, best_s = min(g_ms.values()), min(gs_ms.values())
            ceiling = {"gathers_only_ms": round(best, 3), "gathers_plus_result_stores_ms": round(best_s, 3),
                       "by_gathers_in_flight": {str(u): round(v, 3) for u, v in g_ms.items()},
                       "kernel_ms": round(sec * 1e3, 3),
                       "kernel_reaches_of_gather_ceiling": round(best / (sec * 1e3), 4),
                       "kernel_reaches_of_gather_plus_store_ceiling": round(best_s / (sec * 1e3), 4),
                       "ceiling_as_frac_of_hbm_roofline": round(byts / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "note": "k_rows_tiled with its val stream, fmaf chains and result rows removed (XOR-folded loads): what the memory "
                               "system needs for THIS index stream at THIS hit distribution -- no row-gather SpMM issuing these gathers runs "
                               "faster; tools/spmm_lab holds the other geometries tried (8 / 32 / 64 lanes per item are all slower)"}
            del sink, outc
        del sb
        shard = influence_shard(gb, big.shape[0], scale, hcols) if with_shard else None
        tr = traffic_of("spmm", f"spmm_rmat{scale}")
        tiled = tiled_route(gb, hcols)          # the route the call actually took
        pm_s = (pmc or {}).get("spmm", {})
        pm_c = (pmc or {}).get("gather_ceiling", {})
        rd = pm_s.get("read_bytes_per_launch")
        return {"influence_shard": shard, "gather_ceiling": ceiling,
                "kernel": ("k_rows_tiled (+ k_spmm_long_combine for the hub rows)" if tiled else "k_spmm_rows (+ k_spmm_segments / k_spmm_long_combine for the hub rows)"),
                "bound": "hbm",
                # the HBM-side rate of the bytes the kernel really moves (PMC traffic over the measured duration)
                "traffic_GBps": round(tr / sec / 1e9, 1) if tr else None,
                "traffic_frac_of_peak": round(tr / sec / 1e9 / HBM_PEAK_GBS, 4) if tr else None,
                "achieved": round(byts / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(byts / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": tr,
                "traffic_over_algorithmic": round(tr / byts, 2) if tr else None,
                # counters instead of a host-side histogram (round 5): the L2's hit fraction over the launch, the bytes it asked the
                # memory side for (TCC_EA0_RDREQ x 128 B = 2 x FETCH_SIZE: Infinity Cache AND HBM -- the L2's counters cannot tell the two
                # apart, "DRAM" in their names is the memory controller's side of the fabric) and that read rate
                "l2_hit_frac": pm_s.get("l2_hit_frac"), "l2_hit_frac_gather_ceiling_kernel": pm_c.get("l2_hit_frac"),
                "dram_bytes": rd, "dram_bytes_note": "memory-side read bytes per launch (Infinity Cache + HBM, not separable from the L2)",
                "dram_read_GBps": round(rd / sec / 1e9, 1) if rd else None,
                "algorithmic_bytes_per_launch": int(byts), "avg_launch_us": round(sec * 1e6, 1),
                "units_per_launch": f"one SpMM A_hat[{big.shape[0]}^2, nnz={big.nnz}] x S[{big.shape[0]}x{hcols}] fp32 "
                                    f"(R-MAT scale {scale}, max row {int(np.diff(big.indptr).max())})",
                "gathered_bytes_per_launch": int(big.nnz) * hcols * 4, "host_graph_build_s": round(host_s, 1),
                # every gathered byte passes L2 (~20 TB/s chip-wide when it hits): the cap on the ALGORITHMIC rate of a
                # row-gather SpMM whatever the input's locality; measured 1.37 TB/s on a 2 M-node banded graph
                "l2_gather_bound": {"L2_gather_GBps": 20000.0,
                                    "max_algorithmic_GBps": round(20000.0 * byts / (int(big.nnz) * hcols * 4), 1),
                                    "frac_of_that": round(byts / sec / 1e9 / (20000.0 * byts / (int(big.nnz) * hcols * 4)), 4)},
                "note": "algorithmic bytes = SURVEY 8(d) (CSR once, S once, result once); a row-gather SpMM moves nnz*H*4 "
                        "bytes of gathered rows through L2, and what it cannot hold comes over the fabric: `traffic` "
                        "(FETCH_SIZE*2 + WRITE_SIZE, PMC) over avg_launch_us is the real HBM-side rate"}

    if a.only_spmm:
        print(json.dumps({"roofline_spmm": spmm_rmat_leg(a.spmm_scale, a.hidden, reps=4, with_shard=False)}))
        return

    # ---------------- workload (identical on every rank: seeded) ----------------
    adj, x_np, w = synth.twitch_like_problem(a.workload, hidden=a.hidden, n_classes=a.classes, seed=0,
                                             powerlaw=a.powerlaw)
    a_hat = graph.first_order_gcn(adj)
    n, f = x_np.shape
    h, c = a.hidden, a.classes
    nnz = a_hat.nnz
    np.random.seed(42)
    test_nodes = np.random.choice(np.arange(n), a.n_test, replace=False)

    hg = graph.HipGraph(a_hat)
    x = torch.from_numpy(x_np).to(dev)
    params = [torch.from_numpy(w[k]).to(dev) for k in ("W1", "b1", "W2", "b2")]
    base = engine.Baseline(hg, x, *params)
    if a.mode == "delta":
        base.enable_fp64()
    # N > 1: the loop-invariant product the mode reads is sharded over the ranks + all-gathered, or replicated
    baseline_sharded = lt_dist.choose_baseline_sharding(base, mode=a.mode)
    fp64_route = base.fp64_route()
    b0, b1_, per = lt_dist.shard_bounds(a.n_test, rank, world)
    probes = torch.from_numpy(test_nodes[b0:b1_].astype(np.int32)).to(dev)
    obs = torch.from_numpy(test_nodes.astype(np.int32)).to(dev)
    local = torch.empty((b1_ - b0, a.n_test), dtype=torch.float32, device=dev)
    delta = 1e-4
    pending = []
    probe_sharded = multi          # decided below (LT_SHARD_PROBES=auto: sharded + all-gather against every rank building all rows)

    def step(mode, bs=None):
        """One influence-matrix build.  For N > 1 the all-gather of step k is left in flight on the
        communicator's stream while step k+1 computes (steps are independent; every step's matrix is
        complete before the closing barrier + synchronize)."""
        bs = bs or base
        if multi and not probe_sharded:
            # the probe dimension is not worth sharding for this workload (decided once, the same on every rank): every rank
            # builds all rows, no collective
            bs.refresh(mode)
            bs.influence_rows(obs, obs, delta, mode, out=local_all)
            return local_all
        if multi:
            # a fresh padded slab per step: the collective of step k may still be reading its slab
            # while step k+1 computes
            slab = torch.empty((per, a.n_test), dtype=torch.float32, device=dev)
            if b1_ - b0 < per:
                slab[b1_ - b0:].zero_()
            out = slab[: b1_ - b0]
        else:
            slab = out = local
        bs.refresh(mode)
        bs.influence_rows(probes, obs, delta, mode, out=out)
        full, work = lt_dist.all_gather_rows(slab, a.n_test, async_op=True)
        if work is not None:
            pending.append(work)
        return full

    def probe_sharded_set(v):
        nonlocal probe_sharded
        probe_sharded = v

    def wall_median(fn, steps, blocks=5, warm=3):
        """Seconds per call of fn: median over `blocks` synchronize-bracketed blocks of `steps` calls (the side legs)."""
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(blocks):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps)
        return float(np.median(ts))

    def drain():
        while pending:
            pending.pop().wait()

    def barrier():
        drain()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(el):
        if not multi:
            return el
        t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(mode, steps, warmup, profile_mask=0, blocks=1, bs=None):
        """`blocks` timed blocks of exactly `steps` steps (barrier + synchronize on both sides, MAX over ranks);
        returns (list of block seconds, the last matrix)."""
        for _ in range(warmup):
            step(mode, bs)
        _lib.lib().lt_profile_enable(profile_mask)
        out = []
        full = None
        for _ in range(blocks):
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                full = step(mode, bs)
            barrier()
            out.append(max_over_ranks(time.perf_counter() - t0))
        return out, full

    def step_to_host(mode, bs=None):
        """One influence-matrix build that ends where the reference's timed region ends (attacker.py:213 -> 231): influence_val as
        float64 on the HOST.  One rank: ONE library call forms the rows and lands them in pinned host memory
        (lt_influence_rows_f64), one stream wait.  Several ranks: the step above (rows + all-gather), then the export launch."""
        bs = bs or base
        if (multi and probe_sharded) or not hasattr(bs, "influence_matrix_host"):
            full_ = step(mode, bs)
            drain()
            return engine.export_rows_f64(full_)
        return bs.influence_matrix_host(obs if multi else probes, obs, delta, mode, refresh=True)

    def timed_host(mode, steps, warmup, blocks=1, bs=None):
        # (no HIP events inside these steps: the dominant class is bracketed in the device-resident region above -- left on, the mask of
        # that region put an event pair, ~5 us of stream time, into EVERY host-landed step up to the middle of round 6)
        _lib.lib().lt_profile_enable(0)
        for _ in range(warmup):
            step_to_host(mode, bs)
        out = []
        m = None
        for _ in range(blocks):
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                m = step_to_host(mode, bs)
            barrier()
            out.append(max_over_ranks(time.perf_counter() - t0))
        return out, m

    local_all = torch.empty((a.n_test, a.n_test), dtype=torch.float32, device=dev) if multi else None
    probe_choice = None
    if multi:
        def _sharded_once():
            probe_sharded_set(True)
            step(a.mode)
            drain()

        def _local_once():
            probe_sharded_set(False)
            step(a.mode)
        key_ps = ("bench", a.n_test, a.mode)
        probe_sharded = lt_dist.choose_probe_sharding(key_ps, _sharded_once, _local_once)
        rep = lt_dist.probe_sharding_report(key_ps)
        probe_choice = {"policy": lt_dist.shard_probes_policy(), "sharded": bool(probe_sharded),
                        **({"ms_per_step_every_rank_all_rows": round(rep[1] * 1e3, 4), "ms_per_step_sharded_all_gather": round(rep[2] * 1e3, 4)}
                           if rep else {})}
    n_probe_local = (b1_ - b0) if (probe_sharded or not multi) else a.n_test
    items_local = int(a_hat.tocsc()[:, test_nodes[b0:b1_] if (probe_sharded or not multi) else test_nodes].nnz)   # sum over this rank's probes of |R_v|

    if a.pmc_child:
        # the kernels of the value mode and of `full`, a few launches each, for the counter passes
        for m in dict.fromkeys((a.mode, "full")):
            for _ in range(3):
                step(m)
        torch.cuda.synchronize()
        if a.spmm_scale:
            spmm_rmat_leg(a.spmm_scale, h)
        return

    # ---------------- the timed region ----------------
    # Only the dominant kernel class carries HIP events inside it (one pair per step); bracketing every launch costs
    # ~40 us of stream time per step, so the full per-kernel table comes from a second, instrumented pass right after.
    classes = {"full": ["gemm", "full_stageA", "full_stageB"],
               "sparse": ["gemm", "layer1", "layer2", "item_bits", "item_stageA", "item_stageB"],
               "delta": ["fp64_product", "fp64_spmm", "item_bits", "item_stageA", "item_stageB"]}[a.mode]
    probe_t, _ = timed(a.mode, max(3, a.steps // 10), a.warmup, profile_mask=-1)     # which class dominates on this box
    first = {}
    for k in classes:
        tot, cnt = kernel_ms(k)
        first[k] = tot
    dom_name = max(first, key=first.get)
    # (an event pair costs ~5 us of stream time -- 9 % of a 0.058 ms step: the dominant class is bracketed on every
    # PROFILE_EVERY-th step of the timed region, its average launch duration is the mean of those samples)
    PROFILE_EVERY = 5
    _lib.set_tuning("profile_every", PROFILE_EVERY)
    block_s, full = timed(a.mode, a.steps, 2, profile_mask=1 << _lib.KERNEL_IDS[dom_name], blocks=max(1, a.blocks))
    _lib.set_tuning("profile_every", None)
    dom_tot, dom_cnt = kernel_ms(dom_name)
    # "profile_every" samples whole calls (= steps here): the class's time per step is its total over the SAMPLED STEPS,
    # however many scopes it opens in a step (chunked calls, the three sites of the fp64 product)
    dom_steps = profiled_calls()
    elapsed_dev = float(np.median(block_s))
    ms_per_step_dev = elapsed_dev / a.steps * 1e3
    value_dev = a.n_test * a.n_test * a.steps / elapsed_dev
    # THE headline: the same K steps, each ending with influence_val on the host as float64 -- the region the reference times
    # (attacker.py:213 -> 231).  The device-resident figure above stays in the line as `value_device`.
    block_h, m_host = timed_host(a.mode, a.steps, 3, blocks=max(1, a.blocks))
    elapsed = float(np.median(block_h))
    ms_per_step = elapsed / a.steps * 1e3
    value = a.n_test * a.n_test * a.steps / elapsed
    host_equals_device = bool(m_host is not None and np.array_equal(m_host, full.cpu().numpy().astype(np.float64)))

    elapsed_i, _ = timed(a.mode, a.steps, 0, profile_mask=-1)
    elapsed_i = elapsed_i[0]
    per_kernel = {}
    for k in classes:
        tot, cnt = kernel_ms(k)
        if cnt:
            per_kernel[k] = {"launches_per_step": round(cnt / a.steps, 2), "us_per_step": round(tot / a.steps * 1e3, 2),
                             "share_of_step": round(tot / a.steps / (elapsed_i / a.steps * 1e3), 3)}
    # the same table for the HOST-landed steps (`value`): the product rows' and the pre-activation's launches carry the zero-fill waves
    # of the float64 matrix there, and the probes' launch its transfer
    per_kernel_host = {}
    if not (multi and probe_sharded) and hasattr(base, "influence_matrix_host"):
        _lib.lib().lt_profile_enable(-1)
        for _ in range(a.steps):
            step_to_host(a.mode)
        for k in classes:
            tot, cnt = kernel_ms(k)
            if cnt:
                per_kernel_host[k] = {"launches_per_step": round(cnt / a.steps, 2), "us_per_step": round(tot / a.steps * 1e3, 2)}
    _lib.lib().lt_profile_enable(0)
    # what an empty event pair reads on this stream (median of 50): the overhead every HIP-event duration above carries
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    for e0, e1 in evs:
        e0.record(); e1.record()
    torch.cuda.synchronize()
    event_pair_us = round(float(np.median([e0.elapsed_time(e1) for e0, e1 in evs])) * 1e3, 2)
    if dom_cnt:      # duration of the dominant class as measured INSIDE the timed region (per step = per launch group)
        per_kernel[dom_name]["us_per_step_instrumented_pass"] = per_kernel[dom_name]["us_per_step"]
        per_kernel[dom_name]["us_per_step"] = round(dom_tot / max(dom_steps, 1) * 1e3, 2)
        per_kernel[dom_name]["samples_in_timed_region"] = int(dom_steps)
        per_kernel[dom_name]["scopes_in_those_samples"] = int(dom_cnt)

    def roofline_of(cls, us, mode):
        """Roofline object of one kernel class at `us` microseconds per launch group (one per step)."""
        sec = us * 1e-6
        tr = traffic_of(cls, {"full_stageA": "full_stageA", "gemm": "gemm"}.get(cls)) if (world == 1) else None
        hp = (h + 3) // 4 * 4
        if cls == "full_stageA":
            flop = 2.0 * n_probe_local * nnz * h
            unfused = nnz * 8 + (n + 1) * 4 + n_probe_local * 2 * n * h * 4
            return {"kernel": "k_full_stageA_lds", "bound": "fp32_fma", "achieved": round(flop / sec / 1e12, 2), "peak": FP32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(flop / sec / 1e12 / FP32_PEAK_TFLOPS, 4), "traffic": tr, "avg_launch_us": us,
                    "units_per_launch": f"{n_probe_local} perturbed layer-1 passes over A_hat[{n}x{n}, nnz={nnz}] x S1'[{n}x{h}] = {n_probe_local}*nnz*H FMAs",
                    "algorithmic_flop_per_launch": flop,
                    "unfused_hbm_figure": {"bytes_per_launch": int(unfused), "GBps_at_this_duration": round(unfused / sec / 1e9, 1),
                                           "note": "SURVEY 8(d) batched formula; exceeds the HBM peak because the fused kernel never moves these bytes"}}
        if cls == "gemm":
            rows_x = (lt_dist.shard_bounds(n, rank, world)[1] - lt_dist.shard_bounds(n, rank, world)[0]) if baseline_sharded else n
            flop = 2.0 * (rows_x + n_probe_local) * f * h
            return {"kernel": "k_gemm_f32_mfma_128 + k_gemm_f32_mfma_deep<gather> (+ k_sum_slabs)", "bound": "mfma", "achieved": round(flop / sec / 1e12, 2),
                    "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flop / sec / 1e12 / FP32_PEAK_TFLOPS, 4), "traffic": tr,
                    "avg_launch_us": us, "units_per_launch": f"per step: X[{rows_x}x{f}] * W1[{f}x{h}] and X'[{n_probe_local} probes] * W1, exact fp32 MFMA"}
        if cls == "fp64_product" and fp64_route == 1:
            out_b = 8 if os.environ.get("LT_S1_F32") == "0" else 4      # the rows leave as 32-bit fixed point unless s1_f32 = 0
            alg = n * f * 4 + f * h * 4 + n * hp * out_b
            rec = os.environ.get("LT_RECORDS_EARLY", "1") != "0" and mode == "delta"
            return {"kernel": "k_s1d_feature_rows (the reference vector's product and its slice sum ride in the same launch)",
                    "bound": "hbm", "achieved": round(alg / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": tr, "algorithmic_bytes_per_launch": int(alg), "avg_launch_us": us,
                    **({"also_in_this_launch": "the record blocks of the call's probe chunk (round 5, `records_early`: they left the "
                                               "pre-activation's launch, the step lost 0.7 us; this launch alone reads 21.4 us = 0.37 "
                                               "without them, profiles/r04_step_kernel_stats.csv)"} if rec else {}),
                    "units_per_launch": f"one pass over X[{n}x{f}] fp32 -> S1d = X*W1 [{n}x{h}], fp64-accumulated (feature rows as differences to a "
                                        f"reference row); bytes = N*F*4 + F*H*4 + N*H*{out_b}"}
        if cls == "fp64_product":
            rows_x = (lt_dist.shard_bounds(n, rank, world)[1] - lt_dist.shard_bounds(n, rank, world)[0]) if baseline_sharded else n
            return dense_product_roofline(rows_x, f, h, n, us, tr)
        if cls in ("fp64_spmm", "layer1"):
            alg = spmm_bytes(n, nnz, hp, 8 if cls == "fp64_spmm" else 4)
            return {"kernel": "k_spmm_f64" if cls == "fp64_spmm" else "k_layer1", "bound": "hbm", "achieved": round(alg / sec / 1e9, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": tr, "algorithmic_bytes_per_launch": int(alg),
                    "avg_launch_us": us, "units_per_launch": f"one SpMM A_hat[nnz={nnz}] x S[{n}x{hp}] ({'fp64' if cls == 'fp64_spmm' else 'fp32'}); SURVEY 8(d) bytes; "
                                                             f"operands are L2 / Infinity-Cache resident at this size"}
        if cls == "item_stageA":
            elem = 8 if mode == "delta" else 4
            alg = items_local * (hp * elem + c * 4 + 8) + n_probe_local * hp * elem
            return {"kernel": "k_item_stageA", "bound": "hbm", "achieved": round(alg / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": tr, "algorithmic_bytes_per_launch": int(alg), "avg_launch_us": us,
                    "units_per_launch": f"{items_local} items (probe v, row r in R_v): one pre-activation row of {hp} each -> {c} values"}
        if cls == "item_stageB":
            obs_entries = int(np.diff(a_hat.indptr)[test_nodes].sum())
            alg = n_probe_local * a.n_test * 4 + items_local * c * 4 + obs_entries * 8
            return {"kernel": "k_delta_probe_finish (stage A + B of a probe in one block, from its incidence record; graphs with hub rows: k_item_stageB_rows / _hubs)",
                    "bound": "hbm", "achieved": round(alg / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": tr, "algorithmic_bytes_per_launch": int(alg), "avg_launch_us": us,
                    "units_per_launch": f"{n_probe_local} x {a.n_test} (probe, observed) pairs: the observed rows' CSR once, the items' layer-2 inputs once, "
                                        f"the matrix once -- a latency-bound kernel of cache-resident operands"}
        return {"kernel": cls, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": tr, "avg_launch_us": us}

    roofline = roofline_of(dom_name, per_kernel[dom_name]["us_per_step"], a.mode) if dom_name in per_kernel else None
    if roofline is not None:
        roofline["traffic_source"] = pmc_note if pmc else (pmc_note + "; profiles/pmc_traffic.json when its source stamp matches")
        roofline["every_class"] = {k: {kk: vv for kk, vv in roofline_of(k, v["us_per_step"], a.mode).items()
                                       if kk in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us")}
                                   for k, v in per_kernel.items()}

    # ---------------- the drop-in API: the region the reference itself times --------------------------------------------
    # attacker.py:213 -> 231 ("time for predicting edges"): from the call to influence_val on the HOST as float64.  Here that is
    # Attacker.influence_matrix(): one state_dict walk, the cached device node lists, refresh + lt_influence_rows, ONE export
    # launch that widens on the device and writes into pinned host memory (lt_export_rows_f64), one stream wait.
    api = None
    if rank == 0 and world == 1 and not a.no_api_wall and not a.powerlaw:
        import argparse as _ap
        import types as _types
        from linkteller_amd.attacker import Attacker
        from linkteller_amd.gcn import GCN
        model = GCN(f, h, c, 0.5)
        model.load_state_dict({"gc1.weight": torch.from_numpy(w["W1"]), "gc1.bias": torch.from_numpy(w["b1"]),
                               "gc2.weight": torch.from_numpy(w["W2"]), "gc2.bias": torch.from_numpy(w["b2"])})
        model.to(dev).eval()
        adj_t = graph.sparse_mx_to_torch_sparse_tensor(a_hat).to(dev)
        wk = _types.SimpleNamespace(features_2=x, adj_2=adj_t, adj_ori=adj.tocsr(), n_nodes=n)
        args_ = _ap.Namespace(dataset="twitch/RU", sample_type="unbalanced", n_test=a.n_test, sample_seed=42, influence=delta,
                              mode="vanilla-clean", attack_mode="efficient", influence_mode=a.mode)
        atk = Attacker(args_, model, wk)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            atk.prepare_test_data()
        for _ in range(5):
            m_api = atk.influence_matrix()
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            m_api = atk.influence_matrix()
            ts.append(time.perf_counter() - t0)
        api_s = float(np.median(ts))
        same_nodes = bool(np.array_equal(np.asarray(atk.test_nodes), test_nodes))
        # the export alone (HIP events on the launch stream), and the copy engine's plain fp32 D2H for comparison
        pin64 = torch.empty((a.n_test, a.n_test), dtype=torch.float64, pin_memory=True)
        pin32 = torch.empty((a.n_test, a.n_test), dtype=torch.float32, pin_memory=True)

        def ev_median(fn, reps=30):
            out_ = []
            for i in range(reps + 3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                if i >= 3:
                    out_.append(e0.elapsed_time(e1) * 1e3)
            return round(float(np.median(out_)), 2)
        st_ = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        d2h_us = ev_median(lambda: _lib.check(_lib.lib().lt_export_rows_f64(full.data_ptr(), a.n_test, a.n_test, a.n_test, pin64.data_ptr(),
                                                                           a.n_test, st_)))
        d2h_copy_us = ev_median(lambda: pin32.copy_(full, non_blocking=True))
        # the round-5 form of the host-landed step for comparison: rows on the device, then the export launch, then the wait
        def step_export():
            return engine.export_rows_f64(step(a.mode))
        bl = []
        for _ in range(3):
            step_export()
        for _ in range(max(1, a.blocks)):
            barrier()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step_export()
            barrier()
            bl.append(time.perf_counter() - t0)
        sth = float(np.median(bl)) / a.steps
        api = {"what": "Attacker.influence_matrix(): the region the reference times at attacker.py:213->231 (`time for predicting edges`), "
                       "influence_val on the host as float64; median of 40 calls after 5",
               "api_wall_ms": round(api_s * 1e3, 4), "value_host": round(a.n_test ** 2 / api_s, 1), "unit": "node-pairs/s",
               "api_wall_min_ms": round(float(np.min(ts)) * 1e3, 4), "api_wall_p90_ms": round(float(np.percentile(ts, 90)) * 1e3, 4),
               "d2h_us": d2h_us, "d2h_bytes": a.n_test * a.n_test * 8,
               "d2h_note": "lt_export_rows_f64: one launch widens fp32 -> fp64 on the device and writes into pinned host memory over PCIe "
                           "(HIP events around the launch)",
               "d2h_copy_engine_fp32_us": d2h_copy_us,
               "ms_per_step_export_launch": round(sth * 1e3, 4),
               "export_launch_note": "the device-resident step + lt_export_rows_f64 as a launch of its own + the stream wait (round 5's host "
                                     "path); `value` takes lt_influence_rows_f64, whose probe blocks write their own rows to the host",
               "matrix_equals_step": bool(same_nodes and np.array_equal(m_api, full.cpu().numpy().astype(np.float64))),
               "same_test_nodes_as_step": same_nodes}
        del pin64, pin32, atk, model

    extras = {}
    mats = {}
    if rank == 0 and not a.no_extras and world == 1:
        # other evaluations of the same matrix (never substituted for `value`)
        ref_value = full.clone()
        extras["max_score"] = float(ref_value.max().item())
        mats = {a.mode: ref_value}
        for m in ("full", "sparse", "delta"):
            if m == a.mode:
                continue
            mask = (1 << _lib.KERNEL_IDS["full_stageA"]) if m == "full" else 0
            bl, res = timed(m, a.steps, 2, profile_mask=mask, blocks=3)
            el = float(np.median(bl))
            mats[m] = res.clone()
            om = {"pairs_per_s": round(a.n_test ** 2 * a.steps / el, 1), "ms_per_step": round(el / a.steps * 1e3, 4),
                  "max_abs_diff_vs_value_mode": float((res - ref_value).abs().max().item())}
            if m == "full":
                tot, cnt = kernel_ms("full_stageA")
                _lib.lib().lt_profile_enable(0)
                if cnt:
                    om["roofline"] = roofline_of("full_stageA", round(tot / cnt * 1e3, 2), "full")
            extras.setdefault("other_modes", {})[m] = om
        if "full" in mats and "sparse" in mats:
            extras["other_modes"]["sparse_equals_full_bit_for_bit"] = bool(torch.equal(mats["full"], mats["sparse"]))
        if a.mode == "delta" and fp64_route == 1:
            # the same mode with the fp64 product forced onto the f64 matrix cores (what dense features -- Gaussian,
            # embeddings -- take): the general-case figure next to the one the twitch features' structure allows
            _lib.set_tuning("feature_delta", 0)
            try:
                bl, res = timed("delta", a.steps, 2, profile_mask=1 << _lib.KERNEL_IDS["fp64_product"], blocks=3)
                el = float(np.median(bl))
                tot, cnt = kernel_ms("fp64_product")
                _lib.lib().lt_profile_enable(0)
                us = round(tot / cnt * 1e3, 2) if cnt else None
                extras["other_modes"]["delta_dense_features"] = {
                    "pairs_per_s": round(a.n_test ** 2 * a.steps / el, 1), "ms_per_step": round(el / a.steps * 1e3, 4),
                    "max_abs_diff_vs_value_mode": float((res - ref_value).abs().max().item()),
                    "roofline": dense_product_roofline(n, f, h, n, us),
                    "note": "feature_delta = 0: X*W1 at fp64 grade on the matrix cores (round 6: the error-free split on the int8 cores; 0.163 ms "
                            "on the f64 cores, LT_I8_SPLIT=0), as for features that are not sparse differences"}
            finally:
                _lib.set_tuning("feature_delta", None)
                base.refresh()
        # the same build on a hub-heavy graph of the same size (the real MUSAE graphs are heavy-tailed; the
        # headline graph is Erdos-Renyi as in SURVEY 8(d)): reported next to `value`, never instead of it
        if not a.powerlaw:
            adj_h, _, _ = synth.twitch_like_problem(a.workload, hidden=a.hidden, n_classes=a.classes, seed=0, powerlaw=True)
            ah = graph.first_order_gcn(adj_h)
            base_h = engine.Baseline(graph.HipGraph(ah), x, *params)
            out_h = torch.empty((a.n_test, a.n_test), dtype=torch.float32, device=dev)
            res_h = {}
            for m in ("delta", "full", "sparse"):
                res_h[m] = wall_median(lambda: (base_h.refresh(m), base_h.influence_rows(obs, obs, delta, m, out=out_h)), 20)
            extras["workload_2"] = {"workload": f"n_test={a.n_test} {a.workload}-shaped POWER-LAW graph (same N, E; what real MUSAE graphs look like)",
                                    "max_degree": int(np.diff(ah.indptr).max()),
                                    "value": round(a.n_test ** 2 / res_h[a.mode], 1), "unit": "node-pairs/s", "mode": a.mode,
                                    "ms_per_step": round(res_h[a.mode] * 1e3, 4),
                                    **{f"{m}_ms_per_step": round(res_h[m] * 1e3, 4) for m in res_h if m != a.mode}}
            del base_h, out_h
        # BASELINE configs[3]: the same model served on the LapGraph-perturbed graph (eps = 5, noise_seed = 42; worker.py:281-335):
        # the graph the DP defence publishes is sparse (~E edges, heavy-tailed), the ground truth stays the clean adj_ori
        if not a.powerlaw:
            import contextlib
            import io
            from linkteller_amd import dp as lt_dp
            buf_ = io.StringIO()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(buf_):
                n_l, noise_l, keep_l = lt_dp._lapgraph_inputs(adj, 5.0, 42, "laplace", 1e-5)
            t_draw = time.perf_counter() - t0
            t0 = time.perf_counter()
            top_dev = lt_dp._lapgraph_select_hip(adj, noise_l, keep_l)
            torch.cuda.synchronize()
            t_dev = time.perf_counter() - t0
            t0 = time.perf_counter()
            noise_h = noise_l * np.tri(n_l, n_l, k=-1, dtype=bool)
            import scipy.sparse as _sp
            cells_h = np.asarray(_sp.tril(adj, k=-1) + noise_h).ravel()
            top_host = np.argpartition(cells_h, -keep_l)[-keep_l:]
            t_host = time.perf_counter() - t0
            same_cells = bool(np.array_equal(np.sort(top_dev), np.sort(top_host)))
            del noise_l, noise_h, cells_h
            mat_l = _sp.csr_matrix((np.ones(keep_l, dtype=np.int32), (top_dev // n_l, top_dev % n_l)), shape=(n_l, n_l))
            adj_l = ((mat_l + mat_l.T) > 0).astype(np.float32).tocsr()
            al = graph.first_order_gcn(adj_l)
            base_l = engine.Baseline(graph.HipGraph(al), x, *params)
            out_l = torch.empty((a.n_test, a.n_test), dtype=torch.float32, device=dev)
            res_l = {}
            for m in ("delta", "sparse", "full"):
                res_l[m] = wall_median(lambda: (base_l.refresh(m), base_l.influence_rows(obs, obs, delta, m, out=out_l)), 20)
            extras["workload_4"] = {"workload": f"n_test={a.n_test} on the LapGraph-served graph (eps=5, noise_seed=42; BASELINE configs[3]): "
                                                f"{int(adj_l.nnz // 2)} edges kept of {int(adj.nnz // 2)}, max degree {int(np.diff(al.indptr).max())}",
                                    "value": round(a.n_test ** 2 / res_l[a.mode], 1), "unit": "node-pairs/s", "mode": a.mode,
                                    "ms_per_step": round(res_l[a.mode] * 1e3, 4),
                                    **{f"{m}_ms_per_step": round(res_l[m] * 1e3, 4) for m in res_l if m != a.mode},
                                    "lapgraph_select": {"device_s": round(t_dev, 4), "host_argpartition_s": round(t_host, 4),
                                                        "numpy_laplace_draw_s": round(t_draw, 4), "same_cells": same_cells,
                                                        "note": "lt_lapgraph_select (upload of the N x N float64 noise + add + 8-pass radix select + "
                                                                "index read-back) against the reference's host path (tril + add + argpartition); "
                                                                "the seeded numpy draw is common to both"}}
            del base_l, out_l
        # BASELINE configs[2] on ONE GPU (the config itself shards n_test = 2000 over 8): the same graph, 2000 probes x 2000 observed
        if a.n_test != 2000 and n >= 2000:
            np.random.seed(42)
            nodes3 = torch.from_numpy(np.random.choice(np.arange(n), 2000, replace=False).astype(np.int32)).to(dev)
            out3 = torch.empty((2000, 2000), dtype=torch.float32, device=dev)
            res3 = {}
            for m in ("delta", "sparse", "full"):
                res3[m] = wall_median(lambda: (base.refresh(m), base.influence_rows(nodes3, nodes3, delta, m, out=out3)), 10)
            extras["workload_3"] = {"workload": f"n_test=2000 on the same graph, one GPU (BASELINE configs[2] shards it over 8)",
                                    "value": round(2000 ** 2 / res3[a.mode], 1), "unit": "node-pairs/s", "mode": a.mode,
                                    **{f"{m}_ms_per_step": round(res3[m] * 1e3, 4) for m in res3}}
            del out3
            base.refresh()
        # the 3-layer model (gcn/models.py:28-46, --n-layer 3) at n_test = 500, and `balanced-full`'s chunk shape
        # (attacker.py:250-284: every node a probe, all N observed; 1024 probes per chunk) on the 2-layer model
        if not a.powerlaw:
            rs5 = np.random.RandomState(7)

            def u5(shape, fan_out):
                s_ = 1.0 / np.sqrt(fan_out)
                return torch.from_numpy(rs5.uniform(-s_, s_, size=shape).astype(np.float32)).to(dev)
            h2_ = 64
            b3 = engine.Baseline3(hg, x, u5((f, h), h), u5((h,), h), u5((h, h2_), h2_), u5((h2_,), h2_), u5((h2_, c), c), u5((c,), c))
            out5 = torch.empty((a.n_test, a.n_test), dtype=torch.float32, device=dev)
            g3 = {m: wall_median(lambda: (b3.refresh(), b3.influence_rows(obs, obs, delta, m, out=out5)), 10) for m in ("delta", "sparse")}
            del b3
            nb5 = min(1024, n)
            pr5 = torch.from_numpy(np.random.RandomState(5).choice(n, nb5, replace=False).astype(np.int32)).to(dev)
            ev5 = torch.arange(n, dtype=torch.int32, device=dev)
            out5 = torch.empty((nb5, n), dtype=torch.float32, device=dev)
            bf = {m: wall_median(lambda: (base.refresh(m), base.influence_rows(pr5, ev5, delta, m, out=out5)), 10) for m in ("delta", "sparse")}
            extras["workload_5"] = {
                "gcn3": {"workload": f"GCN3 {f}-{h}-{h2_}-{c} (--n-layer 3, gcn/models.py:28-46), n_test={a.n_test} on the headline graph",
                         "value": round(a.n_test ** 2 / g3[a.mode if a.mode in g3 else "delta"], 1), "unit": "node-pairs/s",
                         **{f"{m}_ms_per_step": round(g3[m] * 1e3, 4) for m in g3}},
                "balanced_full_chunk": {"workload": f"balanced-full (attacker.py:250-284): one chunk of {nb5} probes x all {n} nodes observed, 2-layer model",
                                        "value": round(nb5 * n / bf[a.mode if a.mode in bf else "delta"], 1), "unit": "node-pairs/s",
                                        **{f"{m}_ms_per_step": round(bf[m] * 1e3, 4) for m in bf}},
                "note": "step = loop-invariant baseline of the mode + the probes, device-resident (as workload_2 .. 4)"}
            del out5
            base.refresh()
        # standalone SpMM (lt_spmm_csr_f32) on this graph (cache-resident) and on the R-MAT graph of configs[4]
        def time_spmm(g_, s_, reps=20):
            for _ in range(3):
                engine.spmm(g_, s_)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                engine.spmm(g_, s_)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps
        s1 = torch.randn((n, h), device=dev)
        t = time_spmm(hg, s1, 200)
        extras["spmm_twitch"] = {"us": round(t * 1e6, 2), "algorithmic_GBps": round(spmm_bytes(n, nnz, h) / t / 1e9, 1),
                                 "note": "operands (9.6 MB) are L2/Infinity-Cache resident; wall time incl. launch"}
        if a.spmm_scale:
            extras["roofline_spmm"] = spmm_rmat_leg(a.spmm_scale, h)

    # ---------------- strong-scaling workloads: the configs where sharding matters, at THIS rank count ----------------
    # One driver command per N then yields the 1/2/4/8 curve of all three: the n_test = 500 build of
    # `value` is a 65 us job whose pass over X every rank repeats (it cannot scale), n_test = 2000 and the R-MAT shard can.
    def collective_us(slab_rows, n_cols, reps=20):
        """The all-gather of row slabs ALONE, at the shape one step sends: HIP events on the compute stream around the
        blocking all_gather_into_tensor (the stream waits for the communicator's), median."""
        if not multi:
            return None
        slab = torch.zeros((slab_rows, n_cols), dtype=torch.float32, device=dev)
        ts = []
        for i in range(reps + 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lt_dist.all_gather_rows(slab, world * slab_rows)
            e1.record()
            torch.cuda.synchronize()
            if i >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        return round(float(np.median(ts)), 2)

    def sharded_build(bs, nodes_np, mode, steps, blocks=3, warm=2, strategy=None):
        strategy = {} if strategy is None else strategy
        """(seconds per step, probes per rank): this rank's ceil(n / W) probes x all n observed nodes incl. the loop-invariant
        baseline of the mode, then the single all-gather -- barrier + synchronize around every block, MAX over ranks."""
        n_t = len(nodes_np)
        q0, q1, per_ = lt_dist.shard_bounds(n_t, rank, world)
        pr = torch.from_numpy(nodes_np[q0:q1].astype(np.int32)).to(dev)
        ob = torch.from_numpy(nodes_np.astype(np.int32)).to(dev)

        hub = None
        if multi and mode == "delta" and bs.fp64_route() == 2 and os.environ.get("LT_SHARE_HUB_ROWS", "1") != "0":
            hub = lt_dist.SharedHubRows(bs, ob)
            strategy["hub_rows_shared"] = {"rows": hub.n_rows, "rows_per_rank": hub.per, "all_gather_bytes": hub.collective_bytes}

        def one():
            slab = torch.empty((per_, n_t), dtype=torch.float32, device=dev)
            if q1 - q0 < per_:
                slab[q1 - q0:].zero_()
            bs.refresh(mode)
            if hub is not None:
                hub.exchange()
            bs.influence_rows(pr, ob, delta, mode, out=slab[: q1 - q0])
            full_, work = lt_dist.all_gather_rows(slab, n_t, async_op=True)
            if work is not None:
                pending.append(work)
            return full_
        all_rows = torch.empty((n_t, n_t), dtype=torch.float32, device=dev)

        def one_local():
            bs.refresh(mode)
            bs.influence_rows(ob, ob, delta, mode, out=all_rows)
            return all_rows
        key_ = ("bench", n_t, mode, id(bs))
        use = lt_dist.choose_probe_sharding(key_, lambda: (one(), drain()), one_local, trials=3, warm=1) if multi else False
        strategy.update({"sharded": bool(use), "policy": lt_dist.shard_probes_policy()})
        rep_ = lt_dist.probe_sharding_report(key_)
        if rep_:
            strategy.update({"ms_per_step_every_rank_all_rows": round(rep_[1] * 1e3, 4), "ms_per_step_sharded_all_gather": round(rep_[2] * 1e3, 4)})
        if multi and not use:
            one, per_ = one_local, n_t
        for _ in range(warm):
            one()
        ts = []
        for _ in range(blocks):
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            barrier()
            ts.append(max_over_ranks(time.perf_counter() - t0) / steps)
        return float(np.median(ts)), per_

    scaling = None
    if not a.no_scaling_workloads and (multi or not a.no_extras):
        scaling = {"note": "strong scaling: the same problem at every rank count, probes sharded contiguously over the ranks, one all-gather "
                           "of row slabs per step; ms_per_step = loop-invariant baseline of the mode + this rank's probes + the all-gather, "
                           "MAX over ranks, median block; collective_us = that all-gather alone (HIP events)",
                   "expectation": "configs[1] / [2] in the default mode CANNOT scale, by construction, and a first real N > 1 run must not be read as "
                                  "a regression: the step is a ~45 us job of which ~33 us is the loop-invariant baseline every rank repeats (one pass "
                                  "over X + the pre-activation) and the rest one latency-bound block per probe.  One-GPU emulation of one rank's step "
                                  "(tools/shard_step_time.py, profiles/r05_shard_step.txt, before any collective): n_test = 500: 44.5 us at 1 rank -> "
                                  "40.8 / 40.9 / 39.9 at 2 / 4 / 8 ranks (1.1x); n_test = 2000: 66.8 -> 54.1 / 48.8 / 43.7 (1.5x); the all-gather "
                                  "adds 26-38 us (measured under RCCL at world size 1).  configs[4] is the config where sharding pays: one rank's "
                                  "512 x 4096 share of the R-MAT build is roofline_spmm.influence_shard (2.8-2.9 ms against 14.0 for all 4096 probes "
                                  "on one GPU = 4.9x before the collective; the 3 800 hub rows every rank's probes reach are 84 % of a rank's "
                                  "loop-invariant part and are replicated: DESIGN.md section 7 sizes sharding them)",
                   "mode": a.mode, "n_gpus": world}
        scaling["configs[1]"] = {"workload": f"n_test={a.n_test} (the `value` workload; ends on the host)", "ms_per_step": round(ms_per_step, 4),
                                 "pairs_per_s": round(value, 1), "probes_per_rank": n_probe_local, "collective_us": collective_us(per, a.n_test),
                                 "strategy": probe_choice}
        if n >= 2000:
            np.random.seed(42)
            nodes3 = np.random.choice(np.arange(n), 2000, replace=False)
            st3 = {}
            t3, per3 = sharded_build(base, nodes3, a.mode, 10, strategy=st3)
            scaling["configs[2]"] = {"workload": "n_test=2000 on the same graph (BASELINE configs[2])", "ms_per_step": round(t3 * 1e3, 4),
                                     "pairs_per_s": round(2000 ** 2 / t3, 1), "probes_per_rank": per3,
                                     "collective_us": collective_us(lt_dist.shard_bounds(2000, rank, world)[2], 2000), "strategy": st3}
            base.refresh()
        if a.spmm_scale:
            big, gb, _ = rmat_problem(a.spmm_scale)
            bb = rmat_baseline(gb, big.shape[0], h)
            if a.mode == "delta":
                bb.enable_fp64()
            lt_dist.choose_baseline_sharding(bb, mode=a.mode)
            nodes5 = np.random.RandomState(42).choice(big.shape[0], 4096, replace=False)
            st5 = {}
            t5, per5 = sharded_build(bb, nodes5, a.mode, 3, blocks=3, warm=1, strategy=st5)
            scaling["configs[4]"] = {"workload": f"R-MAT scale {a.spmm_scale} (N={big.shape[0]}, nnz={big.nnz}) F=256 H={h} n_test=4096 (BASELINE configs[4])",
                                     "ms_per_step": round(t5 * 1e3, 3), "pairs_per_s": round(4096 ** 2 / t5, 1), "probes_per_rank": per5,
                                     "collective_us": collective_us(lt_dist.shard_bounds(4096, rank, world)[2], 4096), "strategy": st5}
            del bb
        rmat_cache.clear()

    def rccl_info():
        """Which collective library this process has mapped (/proc/self/maps): the evidence that RCCL itself executed."""
        libs = set()
        try:
            for line in open("/proc/self/maps"):
                nm = line.rsplit("/", 1)[-1].strip()
                if "rccl" in nm or "nccl" in nm:
                    libs.add(nm)
        except OSError:
            pass
        # (librccl.so is mapped by `import torch.distributed` alone; the `nccl-*` shared-memory segment only exists once a
        # communicator has been brought up in this process)
        return {"backend": (dist.get_backend() if multi else None), "forced_at_world_size_1": bool(force and world == 1),
                "librccl_mapped": any("rccl" in x for x in libs),
                "communicator_segment_mapped": any(x.startswith("nccl-") for x in libs), "libs": sorted(libs)}

    # ---------------- CPU reference path (oracle), bounded sample, rank 0 / N=1 only -----------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import linkteller_oracle as O
        xt = torch.from_numpy(x_np)
        adj_t = O.to_torch_sparse(O.first_order_gcn(adj))
        P = {k: torch.from_numpy(w[k]) for k in ("W1", "b1", "W2", "b2")}
        # thread count: the fastest of a short calibration (one probe each); tiny ops do not
        # scale to hundreds of host cores, so "all cores" would understate the CPU path
        ncpu = os.cpu_count() or 1
        best = (None, 1e30)
        calib = {}
        for th in sorted({t for t in (1, 8, 16, 32, 64, ncpu) if t <= ncpu}):      # (ncpu: BASELINE.md section 3's set_num_threads(os.cpu_count()))
            torch.set_num_threads(th)
            tc = time.perf_counter()
            O.influence_matrix(xt, adj_t, P, test_nodes, delta, probe_range=range(0, 1))
            tc = time.perf_counter() - tc
            calib[th] = tc
            if tc < best[1]:
                best = (th, tc)
        torch.set_num_threads(best[0])
        done, t0 = 0, time.perf_counter()
        rows = []
        while done < a.n_test and (time.perf_counter() - t0) < a.cpu_seconds:
            k = min(4, a.n_test - done)
            m = O.influence_matrix(xt, adj_t, P, test_nodes, delta, probe_range=range(done, done + k))
            rows.append(m[done:done + k])
            done += k
        el = time.perf_counter() - t0
        cpu32 = np.vstack(rows)
        cpu = {"value": round(done * a.n_test / el, 1), "unit": "node-pairs/s", "cores": torch.get_num_threads(),
               "kind": "port",
               "sample": f"first {done} of {a.n_test} probes (x {a.n_test} observed nodes) of the same workload, "
                         f"reference op sequence incl. per-probe baseline forward and per-pair .item(), {el:.1f} s",
               "host_cores": ncpu, "cpu_model": cpu_model(),
               "single_thread": {"value": round(a.n_test / calib[1], 1), "unit": "node-pairs/s", "cores": 1,
                                 "sample": f"1 probe x {a.n_test} observed nodes, {calib[1]:.2f} s"},
               "all_cores": {"value": round(a.n_test / calib[ncpu], 1), "unit": "node-pairs/s", "cores": ncpu,
                             "sample": f"1 probe x {a.n_test} observed nodes at torch.set_num_threads(os.cpu_count()) as BASELINE.md section 3 "
                                       f"words it, {calib[ncpu]:.2f} s"},
               "threads_tried": {str(k_): round(a.n_test / v_, 1) for k_, v_ in sorted(calib.items())},
               "cores_note": "`value` is the sample at the FASTEST thread count of the calibration (tiny ops do not scale to hundreds of host "
                             "cores); the all-cores and single-thread figures stand next to it"}
        # ---- parity gate: the same reference path evaluated in fp64 on the first rows of the sample (untimed) ----
        kq = min(done, 8)
        P64 = {k_: v.double() for k_, v in P.items()}
        cpu64 = O.influence_matrix(xt.double(), adj_t.double(), P64, test_nodes, delta, probe_range=range(0, kq))[:kq]
        gap32 = float(np.abs(cpu32[:kq] - cpu64).max())
        scale = float(cpu64.max())
        mats_now = {a.mode: full}
        mats_now.update(mats)
        parity = {"rows_checked": kq, "max_score": scale, "cpu_fp32_vs_cpu_fp64": gap32, "ok": True,
                  "rule": "delta: |gpu - cpu_fp64| <= 1e-4 * max score (north_star); full / sparse (the fp32 finite difference): "
                          "|gpu - cpu_fp64| <= 2 x |cpu_fp32 - cpu_fp64| on the same rows"}
        for m, mat in mats_now.items():
            err = float(np.abs(mat[:kq].cpu().numpy().astype(np.float64) - cpu64).max())
            ok = err <= (1e-4 * scale if m == "delta" else 2.0 * gap32)
            parity[m] = {"max_abs_diff_vs_cpu_fp64": err, "ok": bool(ok)}
            parity["ok"] = bool(parity["ok"] and ok)
        cpu["max_abs_diff_vs_gpu_rows"] = float(np.abs(cpu32 - full[:done].cpu().numpy().astype(np.float64)).max())

    if rank == 0:
        shard_note = ("single GPU" if not multi else
                      (("fp64" if a.mode == "delta" else "fp32") + " X*W1 sharded over ranks + all-gather" if baseline_sharded else
                       ("feature-difference fp64 product on every rank (one pass over X)" if (a.mode == "delta" and fp64_route == 1)
                        else "replicated on every rank")))
        hp = (h + 3) // 4 * 4
        coll = 0
        if multi:
            coll = world * per * a.n_test * 4 if probe_sharded else 0
            if baseline_sharded:
                coll += world * lt_dist.shard_bounds(n, rank, world)[2] * hp * (8 if a.mode == "delta" else 4)
        out = {
            "metric": "influence-matrix node-pairs/sec", "value": round(value, 1), "unit": "node-pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "value_note": "a step ends with influence_val on the HOST as float64 -- the region the reference times (attacker.py:213 -> 231, "
                          "SURVEY 8(d)); inputs (graph, features, weights, node lists) resident in HBM.  ONE library call per step "
                          "(lt_influence_rows_f64): half of the matrix's rows are zero-filled over PCIe by a few waves riding in the "
                          "product rows' and the pre-activation's launches and their probes' blocks send the touched positions only; the "
                          "other rows are widened whole by their blocks (round 6: 0.106 -> 0.094 ms; profiles/r06_host_landed_lab.txt)",
            "value_device": round(value_dev, 1), "ms_per_step_device": round(ms_per_step_dev, 4),
            "value_device_note": "the same K steps ending with the fp32 matrix in HBM (what `value` was up to round 5)",
            "host_matrix_equals_device_matrix": host_equals_device,
            "dtype": "f64" if a.mode == "delta" else "f32",
            "data": "synthetic",
            "config": {"workload": f"n_test={a.n_test} {a.workload}-shaped {'PL' if a.powerlaw else 'ER'} N={n} E={adj.nnz // 2} F={f} H={h} C={c} "
                                   f"influence=1e-4 (BASELINE configs[1])",
                       "graph": f"{'power-law' if a.powerlaw else 'Erdos-Renyi'}, nnz(A_hat)={nnz}, 2-layer GCN, FirstOrderGCN",
                       "features": "standardised 0/1 indicators, Bernoulli(0.006) (two values per column, as the reference's twitch loader "
                                   "produces: utils/load.py:53-59 + StandardScaler)",
                       "mode": a.mode, "probes_per_rank": n_probe_local, "baseline_XW1": shard_note,
                       "probe_sharding": probe_choice if multi else "one rank",
                       "fp64_product_route": {1: "feature rows as differences to a reference row (k_s1d_feature_rows)",
                                              2: "aggregate-first on the rows the probes reach (k_rows_tiled_xf64 + k_gemm_f64_rows)",
                                              0: "f64 matrix cores (k_gemm_f64acc_128)", -1: "not used"}[fp64_route],
                       "collective_bytes_per_step": coll,
                       "step": "baseline forward of the mode + all probes + norms" + (" + all-gather" if multi else "") +
                               " + the matrix as float64 on the host",
                       "host_path": ("lt_influence_rows_f64: the probes' blocks write their rows into pinned host memory (no export launch)"
                                     if (not multi and a.mode == "delta") else "lt_export_rows_f64 behind the last kernel")},
            "timing": {"blocks": len(block_h), "steps_per_block": a.steps, "reported": "median block",
                       "block_ms": [round(b * 1e3, 3) for b in block_h],
                       "device_block_ms": [round(b * 1e3, 3) for b in block_s],
                       "event_pair_overhead_us": event_pair_us,
                       "event_pair_note": "an EMPTY hipEventRecord pair on the kernels' stream reads this much: the per-class avg_launch_us "
                                          "figures (HIP events) sit that far above the rocprofv3 kernel durations in profiles/"},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "kernels": per_kernel,
            "kernels_note": f"dominant class ({dom_name}) timed by HIP events inside the device-resident timed region (`value_device`; every "
                            f"{PROFILE_EVERY}th step: an event pair costs ~5 us of stream time); the other rows from an "
                            f"instrumented repeat of the same {a.steps} steps ({round(elapsed_i / a.steps * 1e3, 4)} ms/step)",
            "kernels_host": per_kernel_host,
            "kernels_host_note": "the same classes over an instrumented repeat of the HOST-landed steps (`value`): the product rows' launch "
                                 "carries 35 % of the float64 matrix's zeros and the pre-activation's 15 % (a few waves each, over PCIe); "
                                 "item_stageB (k_delta_probe_finish) holds the transfer of the rest -- the link, not the kernel's "
                                 "arithmetic, is what it waits for",
        }
        if a.mode != "delta":
            out["parity_note"] = ("this mode is the reference's fp32 finite difference: its raw AUC can move by 1 / n_edges when a low-score "
                                  "edge quantises to 0 (DESIGN.md section 3); `delta` is the mode that meets north_star's 1e-4")
        if api is not None:
            out["api_wall"] = api
            out["value_host"] = api["value_host"]
        out.update(extras)
        if scaling is not None:
            out["scaling_workloads"] = scaling
        out["collectives"] = rccl_info()
        if os.environ.get("LT_BENCH_DUMP"):          # test hook: the matrix of the last timed step
            np.save(os.environ["LT_BENCH_DUMP"], full.cpu().numpy())
    if multi:
        dist.barrier()
        dist.destroy_process_group()
        flush_c_stdio()
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if rank == 0 and parity is not None and not parity["ok"]:
        sys.exit(3)


if __name__ == "__main__":
    main()
