"""Data parallelism for the hot path: rays shard across ranks (one process per GPU), the model is replicated, and the
only exchange is ONE all-reduce(sum) of the flat "fields" gradient per step (RCCL over xGMI on MI355X; gloo in the
CPU tests).  This is the hook the reference disables by forcing world_size = 1 (umhs_pipeline.py:86,108-113)."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_rays(num_rays_global: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of a global ray batch owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(num_rays_global, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def rank_seed(seed: int, rank: int) -> int:
    """nerfstudio semantics: every rank draws its own rays from seed + rank."""
    return seed + rank


def allreduce_flat_grad(grad: torch.Tensor) -> float:
    """Sum the flat gradient over ranks in place; returns the factor (1/world) the optimizer applies to average it."""
    _, w = world()
    if w > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM)
    return 1.0 / w


def broadcast_params(flat: torch.Tensor, src: int = 0) -> None:
    if world()[1] > 1:
        dist.broadcast(flat, src=src)
