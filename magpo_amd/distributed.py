"""Multi-GPU data parallelism: one process per GPU, envs sharded across ranks, ONE all-reduce of the flat
[guider grads | actor grads | loss scalars] buffer per minibatch (RCCL over xGMI; backend "nccl" on ROCm).

Mirrors the reference's two ``pmean`` pairs (rec_magpo.py:395-409): gradients and loss scalars are the
unweighted mean over groups; every group then applies the identical optimiser step, so parameters stay
replicated without a broadcast.  ~0.9 MB per message: latency-bound, so it is a single flat call.
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))  # before RCCL creates its communicator
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"))
    return rank, world, local


def make_grad_sync(world: int) -> Optional[Callable]:
    """Returns grad_sync(learner) -> scale for MagpoLearner.update, or None for a single group."""
    if world <= 1:
        return None

    staged = dist.get_backend() == "gloo"  # CPU rehearsal / tests: stage through host memory

    def grad_sync(learner) -> float:
        g = learner.grad_all
        if staged and g.is_cuda:
            h = g.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            g.copy_(h)
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return 1.0 / world

    return grad_sync


def shard_env_keys(total_keys: torch.Tensor, num_envs: int, rank: int) -> torch.Tensor:
    """Row-major (group, env) layout of the reset keys (rec_magpo.py:642-653): rank r owns rows
    1 + r*N .. 1 + (r+1)*N of split(key, G*N + 1)."""
    return total_keys[1 + rank * num_envs: 1 + (rank + 1) * num_envs]


def rank_world():
    """(rank, world size) of the initialised process group, (0, 1) for a single process."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def broadcast_object(obj, src: int = 0):
    """One small Python object from rank ``src`` to all ranks (identity for a single process)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]
