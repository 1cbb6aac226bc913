"""Time-step sharding across the GPUs of one node (one process per GPU, torch.distributed).

The LEC path shards embarrassingly: every time step is independent except dT/dt, which needs the
neighbouring time steps of T (thermodynamics.py:109-110 of the reference) -- a one-step halo that
each rank loads/generates itself, so the data path has no collective.  The only exchange is one
all_gather of the per-time-step results ([T_local, 16 + 21 L] fp64, a few KB per step) over
RCCL/xGMI (backend "nccl"; "gloo" in the CPU tests); budgets and residuals are then O(T) host work
on the gathered series (calc_budget_and_residual.py:32-56,131-154).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_steps: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced block [t0, t1) of rank `rank`: the first n_steps % world ranks get one extra step."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("need 0 <= rank < world")
    base, extra = divmod(n_steps, world)
    t0 = rank * base + min(rank, extra)
    return t0, t0 + base + (1 if rank < extra else 0)


def halo_range(t0: int, t1: int, n_steps: int) -> Tuple[int, int]:
    """Time steps a rank must hold to differentiate T in time over [t0, t1): one step either side."""
    return max(t0 - 1, 0), min(t1 + 1, n_steps)


def gather_timeseries(local: torch.Tensor, n_steps: int, group=None) -> torch.Tensor:
    """all_gather of per-time-step rows [T_local, n] into the full series [n_steps, n] on every rank.
    Shards may differ by one step; they are padded to the largest shard for the collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if local.shape[0] != n_steps:
            raise ValueError("single process: local series must be the whole series")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    t0, t1 = shard_range(n_steps, world, rank)
    if local.shape[0] != t1 - t0:
        raise ValueError(f"rank {rank}: expected {t1 - t0} local steps, got {local.shape[0]}")
    width = (n_steps + world - 1) // world
    pad = torch.zeros((width, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: t1 - t0] = local
    out = torch.empty((world * width, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    parts = []
    for r in range(world):
        a, b = shard_range(n_steps, world, r)
        parts.append(out[r * width: r * width + (b - a)])
    return torch.cat(parts, dim=0)
