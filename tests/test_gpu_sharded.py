"""The N > 1 path on the GPU box's single card: two processes (gloo rendezvous, both on cuda:0) each
compute their contiguous block of time steps from their own halo'd cube, gather, and must reproduce
the single-process series bit for bit.  (RCCL needs one GPU per rank; the collective itself is also
covered on CPU in test_parallel_cpu.py.)"""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from tests.helpers import synthetic_domain

N_STEPS = 7


def _dom(nan_case=False):
    dom = synthetic_domain(N_STEPS, 5, 12, 128, seed=21)
    if nan_case:        # a top level that is NaN at ONE time step (of the first shard): dropped for EVERY step of every shard
        dom.v[1, 0, :, :] = np.nan
        dom.tair[5, 2, 4, 9] = np.nan      # an interior level in the last shard: repaired by interpolation there only
    return dom


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, nan_case, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    gpu = f"cuda:{rank}" if backend == "nccl" else "cuda:0"      # RCCL: one GPU per rank
    torch.cuda.set_device(gpu)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from lorenzcycletoolkit_amd.engine import LECEngine
        from lorenzcycletoolkit_amd.parallel import compute_shard, gather_result, halo_range, shard_range
        dom = _dom(nan_case)
        t0, t1 = shard_range(N_STEPS, world, rank)
        h0, h1 = halo_range(t0, t1, N_STEPS)
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a[h0:h1])).to(gpu)
        fields = {"tair": dev(dom.tair), "u": dev(dom.u), "v": dev(dom.v), "omega": dev(dom.omega), "geopt": dev(dom.geopt)}
        eng = LECEngine(dom.lat, dom.lon, dom.level, device=gpu)
        box = eng.box_from_limits(dom.lon[2], dom.lon[-3], dom.lat[1], dom.lat[-2])
        res = compute_shard(eng, fields, dom.time_s, N_STEPS, world, rank, box)
        full = gather_result(res, N_STEPS)
        np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.cpu().numpy())
    finally:
        dist.destroy_process_group()


def _single_process_series(nan_case):
    from lorenzcycletoolkit_amd.engine import LECEngine
    dom = _dom(nan_case)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")
    eng = LECEngine(dom.lat, dom.lon, dom.level, device="cuda:0")
    box = eng.box_from_limits(dom.lon[2], dom.lon[-3], dom.lat[1], dom.lat[-2])
    res = eng.compute(dev(dom.tair), dev(dom.u), dev(dom.v), dev(dom.omega), dev(dom.geopt), [box], time_s=dom.time_s)
    return torch.cat([res.scalars, res.levels.reshape(N_STEPS, -1)], dim=1).cpu().numpy()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: runs on a node with >= 2 GPUs")
@pytest.mark.parametrize("nan_case", [False, True])
def test_sharded_series_over_rccl_equals_single_process(tmp_path, nan_case):
    """The product's N > 1 path as shipped: backend "nccl" (= RCCL), one GPU per rank, device-side all_reduce of the int32
    NaN-level mask and all_gather_into_tensor of the device rows (parallel.py), bit for bit the single-process series."""
    world = min(torch.cuda.device_count(), 4)
    want = _single_process_series(nan_case)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), nan_case, "nccl"), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"full_{r}.npy")
        assert np.array_equal(got, want, equal_nan=True), f"rank {r}: RCCL-sharded series differs from the single-process one"


@pytest.mark.parametrize("world,nan_case", [(2, False), (3, False), (2, True), (3, True)])
def test_sharded_series_equals_single_process(tmp_path, world, nan_case):
    """nan_case: the any-time NaN-level mask must be merged across shards (one [28, L] all_reduce), otherwise
    only the shard that holds the NaN time step would drop the level."""
    from lorenzcycletoolkit_amd.engine import LECEngine
    dom = _dom(nan_case)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")
    eng = LECEngine(dom.lat, dom.lon, dom.level, device="cuda:0")
    box = eng.box_from_limits(dom.lon[2], dom.lon[-3], dom.lat[1], dom.lat[-2])
    res = eng.compute(dev(dom.tair), dev(dom.u), dev(dom.v), dev(dom.omega), dev(dom.geopt), [box], time_s=dom.time_s)
    want = torch.cat([res.scalars, res.levels.reshape(N_STEPS, -1)], dim=1).cpu().numpy()
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), nan_case), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"full_{r}.npy")
        assert got.shape == want.shape
        assert np.array_equal(got, want, equal_nan=True), f"rank {r}: sharded series differs from the single-process one"
    if nan_case:
        assert np.isfinite(want[:, :4]).all()      # the energy terms survive: the level was dropped / repaired, not propagated
