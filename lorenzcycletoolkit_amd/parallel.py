"""Time-step sharding across the GPUs of one node (one process per GPU, torch.distributed).

The LEC path shards embarrassingly: every time step is independent except dT/dt, which needs the
neighbouring time steps of T (thermodynamics.py:109-110 of the reference) -- a one-step halo that
each rank loads/generates itself, so the data path has no collective.  The exchanges are one tiny all_reduce
of the NaN-level mask ([28, L] int32; it only matters for fields with below-ground NaNs) and one
gather of the per-time-step results ([T_local, 16 + 21 L] fp64, 6.3 KB per step at L = 37) over
RCCL/xGMI (backend "nccl"; "gloo" in the CPU tests and the one-GPU rehearsals); budgets and residuals are
then O(T) host work on the gathered series (calc_budget_and_residual.py:32-56,131-154).

The gather is built for the fabric it runs on.  xGMI is point-to-point (every GPU has its own link to every
other), so the series goes to the rank that writes the CSVs as ONE send per peer, all in flight at once, each
over its own link (``dist.gather`` = a grouped ncclSend / ncclRecv) -- not around a ring that would carry
world - 1 hops over single links; ``to_all=True`` asks for the all_gather where every rank needs the series.
``SeriesGatherer`` owns every buffer of that exchange (allocated once, two pipeline slots): ``lec_reduce``
writes its packed records straight into the send buffer, the collective is started asynchronously and only
waited for when its slot is reused, so a pass allocates nothing, repacks nothing and the collective of pass i
overlaps the kernels of pass i + 1.
"""
from __future__ import annotations

import time
from typing import Optional, Tuple

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


class ShardContext:
    """This process's place in a time-sharded run of the drop-in CLI / frameworks (one process per GPU): which steps of an
    n-step series it computes (``ranges``), on which device, and how the ranks talk (backend "nccl" = RCCL; "gloo" for
    rehearsals of the N > 1 path on one GPU).  ``args.shard`` carries it through the reference's call signatures."""

    def __init__(self, world: int, rank: int, device, backend: str, group=None):
        self.world, self.rank, self.device, self.backend, self.group = int(world), int(rank), torch.device(device), backend, group

    @property
    def root(self) -> bool:
        return self.rank == 0

    def ranges(self, n_steps: int) -> Tuple[int, int, int, int]:
        """(t0, t1, h0, h1): the rank's own steps [t0, t1) and the steps it must hold [h0, h1) (own + one-step T halo)."""
        if n_steps < self.world:
            raise ValueError(f"{n_steps} time steps cannot be sharded over {self.world} ranks: use at most one rank per time step")
        t0, t1 = shard_range(n_steps, self.world, self.rank)
        return (t0, t1) + halo_range(t0, t1, n_steps)

    def merge_dropmask(self, mask: torch.Tensor) -> None:
        merge_dropmask(mask, self.group, force=True)      # a ShardContext exists only with a process group: world 1 (rehearsal) still reduces

    def gather_rows(self, local: torch.Tensor, n_steps: int) -> Optional[torch.Tensor]:
        """[T_local, n] rows of every rank -> [n_steps, n] on rank 0 (None elsewhere): one gather."""
        g = SeriesGatherer(n_steps, local.shape[1], local.device, group=self.group, dst=0, slots=1, dtype=local.dtype, force=True)
        g.send(0).copy_(local)
        g.start(0)
        out = g.finish(0)
        return None if out is None else out.clone()

    def barrier(self) -> None:
        if self.backend == "nccl":
            dist.barrier(group=self.group, device_ids=[self.device.index])
        else:
            dist.barrier(group=self.group)


def shard_from_env() -> Optional[ShardContext]:
    """The shard context of a process started by ``python -m torch.distributed.run`` (or by ``lorenzcycletoolkit.py --gpus N``):
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, one GPU per rank (LOCAL_RANK), process group over
    LEC_DIST_BACKEND (default "nccl" = RCCL).  None when WORLD_SIZE is absent or 1: the ordinary one-process run --
    unless ``LEC_FORCE_SHARD=1``, which builds the world-1 context: process group, ``barrier(device_ids)``, the gather and the
    device-side mask all_reduce all run over the backend with ONE rank (the rehearsal of the N > 1 code path a one-GPU box allows)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("LEC_FORCE_SHARD", "0") != "1":
        return None
    if world <= 1 and "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("LEC_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError("lorenzcycletoolkit_amd needs an AMD GPU (PyTorch-ROCm): there is no CPU path")
    if backend == "nccl" and world > ndev:
        raise RuntimeError(f"{world} ranks but {ndev} GPU(s): RCCL needs one GPU per rank")
    device = torch.device("cuda", local % ndev)
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return ShardContext(world, rank, device, backend)


def launch_local_ranks(script: str, argv, n: int) -> int:
    """Starts ``n`` rank processes of ``script`` on this node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on
    127.0.0.1) and returns the job's exit code.  Nothing here initialises a GPU (``torch.cuda.device_count()`` does not), so the
    children are ordinary child processes of a GPU-free parent."""
    import os
    import socket
    import subprocess
    import sys
    backend = os.environ.get("LEC_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < n:
        print(f"--gpus {n} asked but this node shows {ndev} GPU(s); one rank per GPU is required", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for pr in list(pending):
                code = pr.poll()
                if code is None:
                    continue
                pending.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in pending:        # one rank failed: the others would wait in a collective for ever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def _dist_on(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


class SeriesGatherer:
    """Allocation-free gather of per-time-step records [T_local, ncol] into the series [n_steps, ncol].

    ``dst``: the rank that receives the series (default 0: it writes the CSVs); ``to_all=True``: every rank receives it.
    Shards may differ by one step; every rank sends ``width = ceil(n_steps / world)`` rows (the spare row stays zero).

        g = SeriesGatherer(n_steps, ncol, device)
        buf = g.send(slot)          # [T_local, ncol] view of the slot's send buffer: hand it to LECEngine.reduce(out=...)
        g.start(slot)               # asynchronous: the collective waits for the stream's work on `buf`, the host does not
        series = g.finish(slot)     # [n_steps, ncol] on the receiving rank(s) (a view of the slot's buffers), else None

    With one process (no process group, or world 1 and ``force=False``) ``finish`` returns the send buffer itself.
    Backend "gloo" with device tensors (the one-GPU rehearsal of the N > 1 path) stages through pinned host buffers that are
    allocated once here; ``profile`` (a dict) then receives the seconds spent per leg.
    """

    def __init__(self, n_steps: int, ncol: int, device, group=None, dst: int = 0, to_all: bool = False, slots: int = 2,
                 dtype=torch.float64, force: bool = False):
        self.n_steps, self.ncol, self.group, self.dst, self.to_all = int(n_steps), int(ncol), group, int(dst), bool(to_all)
        self.device = torch.device(device)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.t0, self.t1 = shard_range(self.n_steps, self.world, self.rank)
        self.width = (self.n_steps + self.world - 1) // self.world
        self.receives = (not self.active) or self.to_all or self.rank == self.dst
        self.slots = int(slots)
        kw = dict(dtype=dtype, device=self.device)
        self._send = [torch.zeros((self.width, self.ncol), **kw) for _ in range(self.slots)]
        self._work = [None] * self.slots
        self._pending = [False] * self.slots
        self.profile: Optional[dict] = None
        self.backend = dist.get_backend(group) if self.active else "none"
        self.staged = self.active and self.backend == "gloo" and self.device.type == "cuda"
        self.even = self.n_steps % self.world == 0
        if not self.active:
            return
        hostkw = dict(dtype=dtype, device="cpu", pin_memory=self.device.type == "cuda")
        if self.receives:
            self._recv = [torch.empty((self.world, self.width, self.ncol), **kw) for _ in range(self.slots)]
            if not self.even:
                idx = [r * self.width + i for r in range(self.world) for i in range(shard_range(self.n_steps, self.world, r)[1]
                                                                                   - shard_range(self.n_steps, self.world, r)[0])]
                self._idx = torch.tensor(idx, dtype=torch.int64, device=self.device)
                self._series = [torch.empty((self.n_steps, self.ncol), **kw) for _ in range(self.slots)]
        if self.staged:
            self._hsend = [torch.zeros((self.width, self.ncol), **hostkw) for _ in range(self.slots)]
            self._staged_ev = [torch.cuda.Event() for _ in range(self.slots)]
            if self.receives:
                self._hrecv = [torch.empty((self.world, self.width, self.ncol), **hostkw) for _ in range(self.slots)]

    def _switch_to_all_gather(self) -> None:
        kw = dict(dtype=self._send[0].dtype, device=self.device)
        self.to_all = True
        had = self.receives
        if not had:
            self._recv = [torch.empty((self.world, self.width, self.ncol), **kw) for _ in range(self.slots)]
            if self.staged:
                self._hrecv = [torch.empty((self.world, self.width, self.ncol), dtype=kw["dtype"], device="cpu", pin_memory=True) for _ in range(self.slots)]
        self._hands_out = had                      # the ranks that were to receive still do; the others drop what arrives
        self.receives = True

    # -- buffers ------------------------------------------------------------------------------------------------------------
    def send(self, slot: int = 0) -> torch.Tensor:
        """The slot's send buffer, this rank's own rows.  Waits (on the stream, not the host) for the collective that last read it."""
        self.wait(slot)
        return self._send[slot][: self.t1 - self.t0]

    def wait(self, slot: int = 0) -> None:
        w = self._work[slot]
        if w is not None:
            w.wait()               # nccl: the current stream waits for the collective; gloo: the host does
            self._work[slot] = None

    def _tick(self, key: str, t0: float) -> float:
        if self.profile is not None:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            now = time.perf_counter()
            self.profile[key] = self.profile.get(key, 0.0) + (now - t0)
            return now
        return t0

    # -- the exchange -------------------------------------------------------------------------------------------------------
    def start(self, slot: int = 0) -> None:
        """Launches the slot's collective.  The send buffer must not be written again before ``send(slot)`` / ``finish(slot)``."""
        self._pending[slot] = True
        if not self.active:
            return
        t = time.perf_counter() if self.profile is not None else 0.0
        src = self._send[slot]
        if self.staged:
            self._hsend[slot].copy_(src, non_blocking=True)
            self._staged_ev[slot].record(torch.cuda.current_stream(self.device))
            self._staged_ev[slot].synchronize()                     # gloo reads host memory: the copy must have landed
            t = self._tick("staging_d2h", t)
            src = self._hsend[slot]
        dstbuf = None
        if self.receives:
            dstbuf = self._hrecv[slot] if self.staged else self._recv[slot]
        if self.to_all:
            self._work[slot] = dist.all_gather_into_tensor(dstbuf.view(self.world * self.width, self.ncol), src, group=self.group, async_op=True)
        else:
            glist = list(dstbuf.unbind(0)) if self.rank == self.dst else None
            dst_global = dist.get_global_rank(self.group, self.dst) if self.group is not None else self.dst
            try:
                self._work[slot] = dist.gather(src, glist, dst=dst_global, group=self.group, async_op=True)
            except (NotImplementedError, RuntimeError) as e:
                # a backend without gather (every rank gets the same exception at the same call, before anything is enqueued):
                # from here on every rank receives the series through an all_gather; only `dst` hands it out
                if "gather" not in str(e).lower() and not isinstance(e, NotImplementedError):
                    raise
                self._switch_to_all_gather()
                self.start(slot)
                return
        if self.profile is not None:
            self._work[slot].wait()
            self._work[slot] = None
            self._tick("collective", t)

    def finish(self, slot: int = 0) -> Optional[torch.Tensor]:
        """Completes the slot's exchange; the series [n_steps, ncol] on the receiving rank(s) -- valid until the slot is used again."""
        if not self._pending[slot]:
            raise RuntimeError("finish() without start()")
        self._pending[slot] = False
        if not self.active:
            return self._send[slot][: self.n_steps]
        t = time.perf_counter() if self.profile is not None else 0.0
        self.wait(slot)
        t = self._tick("collective", t)
        if not self.receives or not getattr(self, "_hands_out", True):
            return None
        buf = self._recv[slot]
        if self.staged:
            buf.copy_(self._hrecv[slot], non_blocking=True)
            t = self._tick("staging_h2d", t)
        flat = buf.view(self.world * self.width, self.ncol)
        if self.even:
            return flat
        torch.index_select(flat, 0, self._idx, out=self._series[slot])
        self._tick("unpack", t)
        return self._series[slot]


def gather_timeseries(local: torch.Tensor, n_steps: int, group=None) -> torch.Tensor:
    """One-shot form: per-time-step rows [T_local, n] of every rank -> the full series [n_steps, n] on EVERY rank
    (``SeriesGatherer`` with ``to_all``; buffers made for this call).  CPU tensors over gloo, device tensors over nccl / gloo."""
    if not _dist_on(group):
        if local.shape[0] != n_steps:
            raise ValueError("single process: local series must be the whole series")
        return local
    g = SeriesGatherer(n_steps, local.shape[1], local.device, group=group, to_all=True, slots=1, dtype=local.dtype)
    if local.shape[0] != g.t1 - g.t0:
        raise ValueError(f"rank {g.rank}: expected {g.t1 - g.t0} local steps, got {local.shape[0]}")
    g.send(0).copy_(local)
    g.start(0)
    return g.finish(0).clone()


def merge_dropmask(mask: torch.Tensor, group=None, force: bool = False) -> None:
    """Element-wise max of every rank's any-time NaN-level mask, in place (a [28, nl] int32 all_reduce): a level
    that stays NaN at any time step of ANY shard is dropped from the pressure integrals of every time step, as
    the reference's dropna(dim=level) on the whole [time, level] array does (energy_contents.py:203-207).
    ``force``: reduce even in a world of one (rehearsals of the N > 1 path)."""
    if not (_dist_on(group) or (force and dist.is_available() and dist.is_initialized())):
        return
    if dist.get_backend(group) == "gloo" and mask.is_cuda:      # CPU rehearsal: stage through host memory
        host = mask.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        mask.copy_(host)
    else:
        dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=group)


def compute_shard(engine, fields, time_s_global, n_steps: int, world: int, rank: int, box, *, with_q=True,
                  phi_scale=1.0, timing=None, group=None, out=None):
    """Runs the engine on this rank's contiguous block of time steps.

    ``fields``: dict with this rank's cubes (tair, u, v, omega, geopt) covering the HALO range
    ``halo_range(*shard_range(...))`` of the global series -- every rank holds its own steps plus one
    step either side, so dT/dt (np.gradient over the global time axis) needs no exchange.
    ``time_s_global``: seconds of all n_steps.  ``out``: where the packed records go (``SeriesGatherer.send``).
    Returns the LECResult of the rank's own steps."""
    t0, t1 = shard_range(n_steps, world, rank)
    h0, h1 = halo_range(t0, t1, n_steps)
    if fields["tair"].shape[0] != h1 - h0:
        raise ValueError(f"rank {rank}: cube must hold time steps [{h0}, {h1}) (own steps plus halo)")
    return engine.compute(fields["tair"], fields["u"], fields["v"], fields["omega"], fields.get("geopt"), [box],
                          time_s=time_s_global[h0:h1] if with_q else None, t_begin=t0 - h0, t_count=t1 - t0,
                          with_q=with_q, phi_scale=phi_scale, timing=timing, out=out,
                          merge_dropmask=(lambda m: merge_dropmask(m, group)) if world > 1 else None)


def gather_result(res, n_steps: int, group=None):
    """One-shot form: the packed records [T_local, 16 + 21 nl] of every rank -> the full series on every rank."""
    local = res.packed if res.packed is not None else torch.cat([res.scalars, res.levels.reshape(res.levels.shape[0], -1)], dim=1)
    return gather_timeseries(local, n_steps, group)


# ---------------------------------------------------------------------------------------------------------------------------
# Self-verification of a sharded run (what a first N > 1 run on hardware nobody has watched must say about itself)
# ---------------------------------------------------------------------------------------------------------------------------
def record_checksums(rows: torch.Tensor) -> torch.Tensor:
    """[T, ncol] fp64 records -> [T] int64: the wrapping sum of every record's 64-bit patterns weighted by the odd numbers
    1, 3, 5 ... (so swapped columns change it; +0.0 / -0.0 and NaN payloads count as the bits they are)."""
    if rows.dtype != torch.float64 or rows.dim() != 2:
        raise ValueError("record_checksums wants [T, ncol] float64")
    bits = rows.contiguous().view(torch.int64)
    w = torch.arange(rows.shape[1], dtype=torch.int64, device=rows.device) * 2 + 1
    return (bits * w).sum(dim=1)


def _all_gather_small(t: torch.Tensor, group=None) -> torch.Tensor:
    """[n] tensor of every rank -> [world, n] on every rank (gloo with device tensors: through host memory)."""
    world = dist.get_world_size(group)
    staged = dist.get_backend(group) == "gloo" and t.is_cuda
    src = t.cpu() if staged else t.contiguous()
    out = torch.empty((world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out.view(-1), src.view(-1), group=group)
    return out.to(t.device) if staged else out


def verify_gather(series: Optional[torch.Tensor], local: torch.Tensor, n_steps: int, group=None, dst: int = 0) -> Optional[dict]:
    """Checks EVERY rank's block of a gathered series, not only the receiving rank's own (collective: call it on every rank).

    Each rank checksums the records it SENT (``local`` [T_local, ncol]); the checksums travel by an all_gather -- another
    collective than the gather that moved the records --; the receiving rank recomputes them from ``series`` [n_steps, ncol]
    block by block.  A permuted, stale, truncated or zero-filled peer block shows as ``blocks_ok[r] = False``.
    Returns {"peer_blocks_ok", "blocks_ok", "blocks_checked", "steps_checked"} on rank ``dst``, None elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    width = (n_steps + world - 1) // world
    own = torch.zeros((width,), dtype=torch.int64, device=local.device)
    t0, t1 = shard_range(n_steps, world, rank)
    if local.shape[0] != t1 - t0:
        raise ValueError(f"rank {rank}: expected {t1 - t0} local records, got {local.shape[0]}")
    own[: t1 - t0] = record_checksums(local)
    allsums = _all_gather_small(own, group)
    if rank != dst:
        return None
    ok = []
    if series is None or tuple(series.shape) != (n_steps, local.shape[1]):
        ok = [False] * world
    else:
        got = record_checksums(series)
        for r in range(world):
            a, b = shard_range(n_steps, world, r)
            ok.append(bool(torch.equal(got[a:b], allsums[r, : b - a].to(got.device))))
    return {"peer_blocks_ok": all(ok), "blocks_ok": ok, "blocks_checked": world, "steps_checked": n_steps}


def ranks_and_devices(device, group=None) -> dict:
    """How many ranks the backend really connected (an all_reduce of ones) and which GPU each one drives (host name, device
    index, PCI bus id, gathered): under "nccl" (= RCCL) the GPUs must be pairwise distinct -- two ranks on one card is a
    mis-launch that would still print a plausible line.  Collective: call it on every rank; every rank gets the dict."""
    import socket
    device = torch.device(device)
    backend = dist.get_backend(group)
    one = torch.ones((1,), dtype=torch.int64, device=device if backend != "gloo" else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
    ident = {"host": socket.gethostname(), "device_index": device.index, "pci": None, "name": None}
    if device.type == "cuda":
        pr = torch.cuda.get_device_properties(device)
        ident["name"] = pr.name
        bus = [getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        if bus[1] is not None:
            ident["pci"] = "%04x:%02x:%02x" % tuple(int(b or 0) for b in bus)
        uuid = getattr(pr, "uuid", None)
        if uuid is not None:
            ident["uuid"] = str(uuid)
    idents = [None] * dist.get_world_size(group)
    dist.all_gather_object(idents, ident, group=group)
    # two ranks drive one GPU only if EVERYTHING that names it agrees (a runtime that reports the same UUID for every card must not
    # make distinct device indices / bus ids look like one GPU)
    keys = [(d["host"], d["device_index"], d["pci"], d.get("uuid")) for d in idents]
    return {"ranks_seen": int(one.item()), "world_size": dist.get_world_size(group), "backend": backend, "devices": idents,
            "devices_distinct": len(set(keys)) == len(keys)}
