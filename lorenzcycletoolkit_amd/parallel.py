"""Time-step sharding across the GPUs of one node (one process per GPU, torch.distributed).

The LEC path shards embarrassingly: every time step is independent except dT/dt, which needs the
neighbouring time steps of T (thermodynamics.py:109-110 of the reference) -- a one-step halo that
each rank loads/generates itself, so the data path has no collective.  The exchanges are one tiny all_reduce
of the NaN-level mask ([28, L] int32; it only matters for fields with below-ground NaNs) and one
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


def merge_dropmask(mask: torch.Tensor, group=None) -> None:
    """Element-wise max of every rank's any-time NaN-level mask, in place (a [28, nl] int32 all_reduce): a level
    that stays NaN at any time step of ANY shard is dropped from the pressure integrals of every time step, as
    the reference's dropna(dim=level) on the whole [time, level] array does (energy_contents.py:203-207)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    if dist.get_backend(group) == "gloo" and mask.is_cuda:      # CPU rehearsal: stage through host memory
        host = mask.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        mask.copy_(host)
    else:
        dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=group)


def compute_shard(engine, fields, time_s_global, n_steps: int, world: int, rank: int, box, *, with_q=True,
                  phi_scale=1.0, timing=None, group=None):
    """Runs the engine on this rank's contiguous block of time steps.

    ``fields``: dict with this rank's cubes (tair, u, v, omega, geopt) covering the HALO range
    ``halo_range(*shard_range(...))`` of the global series -- every rank holds its own steps plus one
    step either side, so dT/dt (np.gradient over the global time axis) needs no exchange.
    ``time_s_global``: seconds of all n_steps.  Returns the LECResult of the rank's own steps."""
    t0, t1 = shard_range(n_steps, world, rank)
    h0, h1 = halo_range(t0, t1, n_steps)
    if fields["tair"].shape[0] != h1 - h0:
        raise ValueError(f"rank {rank}: cube must hold time steps [{h0}, {h1}) (own steps plus halo)")
    return engine.compute(fields["tair"], fields["u"], fields["v"], fields["omega"], fields.get("geopt"), [box],
                          time_s=time_s_global[h0:h1] if with_q else None, t_begin=t0 - h0, t_count=t1 - t0,
                          with_q=with_q, phi_scale=phi_scale, timing=timing,
                          merge_dropmask=(lambda m: merge_dropmask(m, group)) if world > 1 else None)


def gather_result(res, n_steps: int, group=None):
    """One collective: [T_local, 16 + 21 nl] rows of every rank -> full series on every rank.
    With the gloo backend (CPU rehearsal) the rows are staged through host memory."""
    local = torch.cat([res.scalars, res.levels.reshape(res.levels.shape[0], -1)], dim=1)
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "gloo":
        return gather_timeseries(local.cpu(), n_steps, group).to(local.device)
    return gather_timeseries(local, n_steps, group)
