"""The N > 1 path on CPU: contiguous time-step shards, halo ranges, and the single collective
(all_gather of per-time-step rows) with world_size 2 and 3 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lorenzcycletoolkit_amd.parallel import (SeriesGatherer, gather_timeseries, halo_range, merge_dropmask, ranks_and_devices,
                                             record_checksums, shard_range, verify_gather)
from lorenzcycletoolkit_amd.tables import budgets_and_residuals


def test_shard_ranges_partition_the_series():
    for n in (1, 5, 64, 2048, 4097):
        for w in (1, 2, 3, 4, 8):
            r = [shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_halo_ranges():
    assert halo_range(0, 4, 10) == (0, 5)
    assert halo_range(4, 8, 10) == (3, 9)
    assert halo_range(8, 10, 10) == (7, 10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t0, t1 = shard_range(n_steps, world, rank)
        # the "results" of time step t: 3 columns that identify t unambiguously
        t = torch.arange(t0, t1, dtype=torch.float64)
        local = torch.stack([t, t * t, 1000.0 + t], dim=1)
        full = gather_timeseries(local, n_steps)
        np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
        # the NaN-level mask: every rank flags a different level; the merge is the element-wise max
        mask = torch.zeros((28, 6), dtype=torch.int32)
        mask[rank, rank % 6] = 1
        mask[27, 5] = rank + 1
        merge_dropmask(mask)
        np.save(os.path.join(out_dir, f"mask_{rank}.npy"), mask.numpy())
        # the allocation-free form the product uses: records written in place into the send buffer of a pipeline slot, the series
        # gathered to rank 0 only; three passes through two slots -- pass p carries p in its fourth column
        g = SeriesGatherer(n_steps, 4, "cpu", dst=0, slots=2)
        assert (g.t0, g.t1) == (t0, t1) and g.receives == (rank == 0)
        ptrs = [g.send(sl).data_ptr() for sl in range(2)]
        got = []
        for p in range(3):
            sl = p % 2
            if p >= 2:
                got.append(g.finish(sl))
                got[-1] = None if got[-1] is None else got[-1].clone()
            buf = g.send(sl)
            assert buf.data_ptr() == ptrs[sl] and buf.shape == (t1 - t0, 4)      # the same buffer every pass: nothing is allocated
            buf[:, :3] = local
            buf[:, 3] = float(p)
            g.start(sl)
        for p in (1, 2):
            r = g.finish(p % 2)
            got.append(None if r is None else r.clone())
        if rank == 0:
            np.save(os.path.join(out_dir, "passes.npy"), np.stack([x.numpy() for x in got]))
        else:
            assert all(x is None for x in got)
        # self-verification of a sharded run: every rank's block of the gathered series against the checksums of what that rank
        # sent.  The honest series passes; a series whose PEER blocks are permuted, stale (another pass) or zeroed does not --
        # rank 0's own block is intact in each of them, which is all the round-3 bench line looked at
        g3 = SeriesGatherer(n_steps, 3, "cpu", dst=0, slots=1)
        g3.send(0)[:] = local
        g3.start(0)
        s3 = g3.finish(0)
        verdicts = {"honest": verify_gather(s3, local, n_steps)}
        if rank == 0:
            w0 = shard_range(n_steps, world, 0)[1]
            swapped = s3.clone()
            swapped[w0:] = s3[w0:].flip(0)                   # the peers' rows in another order
            stale = s3.clone()
            stale[w0:, 2] += 1.0                             # the peers' rows of some other pass
            zero = s3.clone()
            zero[-1] = 0.0                                   # a row that never arrived
            own = s3.clone()
            own[0, 0] = -1.0                                 # and rank 0's own block
            tampered = {"swapped": swapped, "stale": stale, "zero": zero, "own": own, "short": s3[:-1]}
        for name in ("swapped", "stale", "zero", "own", "short"):
            verdicts[name] = verify_gather(tampered[name] if rank == 0 else None, local, n_steps)
        who = ranks_and_devices("cpu")
        assert who["ranks_seen"] == world and who["world_size"] == world and len(who["devices"]) == world and who["backend"] == "gloo"
        if rank == 0:
            import json
            json.dump({k: v for k, v in verdicts.items()}, open(os.path.join(out_dir, "verdicts.json"), "w"))
        else:
            assert all(v is None for v in verdicts.values())
        # a backend without gather: the gatherer switches to an all_gather on the first call, rank 0 still gets the series, the others None
        real_gather = dist.gather

        def no_gather(*a, **k):
            raise NotImplementedError("gather is not implemented by this backend")
        dist.gather = no_gather
        try:
            g2 = SeriesGatherer(n_steps, 3, "cpu", dst=0, slots=1)
            g2.send(0)[:] = local
            g2.start(0)
            s2 = g2.finish(0)
            assert g2.to_all
            if rank == 0:
                np.save(os.path.join(out_dir, "fallback.npy"), s2.numpy())
            else:
                assert s2 is None
        finally:
            dist.gather = real_gather
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_steps", [(2, 8), (2, 7), (3, 10)])
def test_gather_timeseries_gloo(tmp_path, world, n_steps):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_steps, str(tmp_path)), nprocs=world, join=True)
    t = np.arange(n_steps, dtype=np.float64)
    want = np.stack([t, t * t, 1000.0 + t], axis=1)
    want_mask = np.zeros((28, 6), dtype=np.int32)
    for r in range(world):
        want_mask[r, r % 6] = 1
    want_mask[27, 5] = world
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"full_{r}.npy"), want)
        assert np.array_equal(np.load(tmp_path / f"mask_{r}.npy"), want_mask)
    assert np.array_equal(np.load(tmp_path / "fallback.npy"), want)
    import json
    vd = json.load(open(tmp_path / "verdicts.json"))
    assert vd["honest"]["peer_blocks_ok"] is True and vd["honest"]["blocks_ok"] == [True] * world and vd["honest"]["steps_checked"] == n_steps
    for name in ("swapped", "stale", "zero"):
        assert vd[name]["peer_blocks_ok"] is False and vd[name]["blocks_ok"][0] is True and not all(vd[name]["blocks_ok"][1:]), name
    assert vd["own"]["blocks_ok"][0] is False and all(vd["own"]["blocks_ok"][1:])
    assert vd["short"]["peer_blocks_ok"] is False
    passes = np.load(tmp_path / "passes.npy")
    assert passes.shape == (3, n_steps, 4)
    for p in range(3):
        assert np.array_equal(passes[p, :, :3], want) and np.all(passes[p, :, 3] == p)


def test_series_gatherer_single_process():
    g = SeriesGatherer(5, 3, "cpu")
    assert not g.active and g.receives and (g.t0, g.t1) == (0, 5)
    g.send(0)[:] = 7.0
    g.start(0)
    s = g.finish(0)
    assert s.shape == (5, 3) and s.data_ptr() == g.send(0).data_ptr() and torch.all(s == 7.0)
    with pytest.raises(RuntimeError):
        g.finish(0)


def test_merge_dropmask_single_process_is_a_no_op():
    mask = torch.tensor([[0, 1], [2, 0]], dtype=torch.int32)
    merge_dropmask(mask)
    assert mask.tolist() == [[0, 1], [2, 0]]


def test_budgets_on_gathered_series_equal_single_process():
    """Budgets need the whole series (np.gradient over time): computed after the gather they equal
    the single-process result -- sharding changes nothing downstream."""
    rng = np.random.default_rng(0)
    n = 9
    s = {k: rng.standard_normal(n) for k in ["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe"]}
    t = np.arange(n) * 3600.0
    whole = budgets_and_residuals(s, t)
    parts = [shard_range(n, 2, r) for r in range(2)]
    glued = {k: np.concatenate([v[a:b] for a, b in parts]) for k, v in s.items()}
    again = budgets_and_residuals(glued, t)
    for k in whole:
        assert np.array_equal(whole[k], again[k])


def test_record_checksums_see_order_sign_of_zero_and_nan_payloads():
    a = torch.tensor([[1.0, 2.0, 3.0], [0.0, -0.0, float("nan")]], dtype=torch.float64)
    c = record_checksums(a)
    assert c.dtype == torch.int64 and c.shape == (2,)
    assert record_checksums(a[:, [1, 0, 2]])[0] != c[0]                    # swapped columns
    b = a.clone()
    b[1, 0] = -0.0
    assert record_checksums(b)[1] != c[1]                                   # the sign of a zero is a bit
    assert torch.equal(record_checksums(a.clone()), c)
    with pytest.raises(ValueError):
        record_checksums(a.float())
