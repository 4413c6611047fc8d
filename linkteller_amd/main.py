"""Command line of the reference (main.py:17-93, README.md:24-98) for the attack path:

    python -m linkteller_amd.main --mode vanilla-clean --dataset twitch/ES/RU --hidden 256 \
        --norm FirstOrderGCN --test --model-path model.pt --attack --attack-mode efficient \
        --sample-type unbalanced --n-test 500 [--influence-mode {full,sparse,delta}]

Every flag of the reference is accepted with its default; only ``--test`` (inference + attack on a
reference-trained ``state_dict``) is implemented -- training stays with the reference.
"""
from __future__ import annotations

import argparse
import logging
import random

import numpy as np
import torch

_NORMS = ["AugNormAdj", "FirstOrderGCN", "BingGeNormAdj", "NormAdj", "RWalk", "AugRWalk"]
_SAMPLE_TYPES = ["balanced", "unbalanced", "unbalanced-lo", "unbalanced-hi", "bfs", "balanced-full"]
_ATTACK_MODES = ["efficient", "naive", "baseline", "baseline-feat"]

# typed options: name -> (type, default[, choices]); argparse derives the reference's dest names
_OPTIONS = {
    "seed": (int, 42), "num-epochs": (int, 500), "lr": (float, 0.01), "weight_decay": (float, 5e-4),
    "hidden": (int, 16), "hidden1": (int, 16), "hidden2": (int, 16), "dropout": (float, 0.5),
    "dataset": (str, "cora"), "model-path": (str, ""), "mode": (str, "vanilla-clean"),
    "init-method": (str, "knn"), "cluster-method": (str, "hierarchical"), "scale": (str, "small"),
    "break-method": (str, "kmeans"), "norm": (str, "AugNormAdj", _NORMS),
    "sample-type": (str, "balanced", _SAMPLE_TYPES), "epsilon": (float, 0.1), "delta": (float, 1e-5),
    "influence": (float, 0.0001), "train-ratio": (float, 0.5), "patience": (int, 10),
    "n-clusters": (int, 10), "n-test": (int, 100), "n-layer": (int, 2), "break-ratio": (float, 1),
    "feature-size": (int, -1), "k": (float, 1), "noise-seed": (int, 42), "sample-seed": (int, 42),
    "cluster-seed": (int, 42), "knn": (int, -1), "noise-type": (str, "laplace"),
    "perturb-type": (str, "discrete", ["discrete", "continuous"]),
    "attack-mode": (str, "efficient", _ATTACK_MODES), "coeff": (float, 1), "degree": (int, 2),
    # additions (never renames): how lt_influence_rows evaluates a probe.  'delta' propagates the perturbation exactly
    # (scores / AUC / AP equal the reference evaluated in fp64); 'sparse' is the reference's fp32 finite difference
    # restricted to the rows a probe can change, bit-identical to 'full' (every probe a full forward); where ./data lives
    "influence-mode": (str, "delta", ["full", "sparse", "delta"]), "data-root": (str, "./data"),
}
_SWITCHES = ["no-cuda", "fastmode", "approx", "attack", "test", "break-down", "display", "same-size",
             "eval-degree", "trainable", "early", "fnormalize"]


def build_parser():
    p = argparse.ArgumentParser(description="LinkTeller attack path on MI355X")
    for name, spec in _OPTIONS.items():
        kw = dict(type=spec[0], default=spec[1])
        if len(spec) > 2:
            kw["choices"] = spec[2]
        p.add_argument(f"--{name}", **kw)
    for name in _SWITCHES:
        p.add_argument(f"--{name}", action="store_true", default=False)
    p.set_defaults(assign_seed=42)
    return p


def get_arguments(argv=None):
    return build_parser().parse_args(argv)


def init_distributed():
    """One process per GPU (``torchrun --nproc-per-node N -m linkteller_amd.main ...``): pin this rank's device
    BEFORE anything touches the GPU (Worker moves its tensors with ``.cuda()``) and join the process group, so
    that ``Attacker.influence_matrix`` shards the probes (linkteller_amd/dist.py) and only rank 0 writes the
    result file.  ``LT_DIST_BACKEND`` / ``LT_DIST_DEVICE`` are test hooks (gloo, all ranks on one device), and so is
    ``LT_FORCE_COLLECTIVES=1`` (a group and every collective even at world size 1: RCCL on a one-GPU box).
    Returns True when a group was created here (the caller destroys it)."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("LT_FORCE_COLLECTIVES") == "1"       # dist.force_collectives(): the N > 1 path at world size 1
    if (world <= 1 and not force) or dist.is_initialized():
        return False
    rank = int(os.environ.get("RANK", "0"))
    if "MASTER_PORT" not in os.environ:
        if world > 1:      # (a port invented per rank would differ on every rank: the rendezvous would hang until the store timeout)
            raise RuntimeError("WORLD_SIZE > 1 but MASTER_PORT is not set: launch the ranks with torchrun "
                               "(python -m torch.distributed.run --nproc-per-node N -m linkteller_amd.main ...)")
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    local = int(os.environ.get("LT_DIST_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("LT_DIST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return True


def main(argv=None):
    args = get_arguments(argv)
    owns_group = init_distributed()
    try:
        _run(args)
    except BaseException:
        # A rank that failed must NOT enter a barrier: its peers sit in the all-gather (or in the sharding policy's
        # collectives), the barrier would be a mismatched collective and the job would hang until the RCCL timeout with
        # the real exception hidden.  Re-raise at once; the launcher (torchrun) tears the other ranks down.
        raise
    else:
        if owns_group:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


def _run(args):
    print(str(args))
    logging.info(str(args))
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(args.seed)
    if not args.test:
        raise NotImplementedError("only --test (inference + attack on a saved state_dict) is implemented; "
                                  "train with the reference")
    from .trainer import GCNTrainer
    from .worker import Worker
    worker = Worker(args, dataset=args.dataset, mode=args.mode, data_root=args.data_root)
    trainer = GCNTrainer(args, worker=worker)
    trainer.init_model(model_path=args.model_path)
    trainer.test(args.eval_degree)


if __name__ == "__main__":
    main()
