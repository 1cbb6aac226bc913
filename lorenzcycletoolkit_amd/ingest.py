"""Device ingest: file bytes -> GPU -> LEC terms, without a host pass over the data.

The reference decodes the whole file with xarray, then copies it four more times on the host
(longitude wrap + sorts, level filter, domain crop, unit conversion; src/utils/preprocessing.py:35-371,
src/utils/select_area.py:254-338, src/utils/box_data.py:297-310) before any term is computed.  Here the
memory-mapped file bytes of a chunk of time steps go over PCIe as they are (int16-packed ERA5 data is a
quarter of its fp64 size), ``lec_ingest`` decodes / sorts / crops / converts them in one gather pass on the
GPU, ``lec_rowstats`` turns the chunk into row records, the level half of ``lec_reduce`` condenses them at once
(12 KB per time step: the records live for one chunk), and its vertical half runs once over the whole series.  Chunks are double-buffered: the copy of chunk c+1 (copy stream) overlaps the kernels of
chunk c (compute stream).  Results are bit-identical to the resident path (``LECEngine.compute`` on the
host-prepared cubes): stage 1 is per row, and every chunk carries the one-step halo of T that dT/dt needs.

Both frameworks: one fixed box for the whole series, or one box per time step along a track (the analysis domain is then
the track-extent crop of slice_domain, select_area.py:297-313, and dT/dt is differentiated over the track-selected times on
the device, as lorenzcycletoolkit.py:184-186 does on the host).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from . import dataset as ds
from .engine import LECEngine, LECResult

_ROLE_KEYS = {"Air Temperature": "tair", "Eastward Wind Component": "u", "Northward Wind Component": "v",
              "Omega Velocity": "omega"}


IngestPlan, make_plan = ds.IngestPlan, ds.make_plan        # the index maps are built on the host side of the package


@dataclass
class StreamedDataset:
    """What ``lec_fixed`` / ``BoxData`` receive instead of a host-prepared LECDataset when the device ingest is used:
    the coordinates of the analysis domain now, the field data streamed from the file when the terms are computed."""
    raw: ds.RawDataset
    plan: IngestPlan
    chunk_steps: Optional[int] = None      # None: lec_streamed's choice (8; 12 when the chunks are inflated on the device)
    inflate: str = "auto"                   # where chunked NetCDF-4 variables are inflated (lec_streamed)

    def level_slice(self, role: str, level_pa: float, t_range=None) -> np.ndarray:
        """[time, lat, lon] of one role at one level, decoded on the host from the mapped file (the 850-hPa track diagnostics
        need three such slices: 3 / (5 * levels) of the data).  ``t_range``: only the processed steps [a, b)."""
        k = int(np.flatnonzero(self.plan.level == level_pa)[0])
        sub = ds.IngestPlan(self.plan.tsel, self.plan.kmap[k:k + 1], self.plan.jmap, self.plan.imap, self.plan.lat, self.plan.lon,
                            self.plan.level[k:k + 1], self.plan.time)
        return ds.gather_on_host(self.raw.variables[self.raw.names[role]], sub, t_range)[:, 0]

    lat = property(lambda self: self.plan.lat)
    lon = property(lambda self: self.plan.lon)
    level = property(lambda self: self.plan.level)
    time = property(lambda self: self.plan.time)
    time_s = property(lambda self: self.plan.time_s)
    names = property(lambda self: self.raw.names)


class StreamedRefusal(ValueError):
    """The streamed device path declined an input (a layout it does not read, a framework it does not serve ...): raised by
    ``refusals()`` around ``prepare_streamed`` / ``lec_streamed`` only, so that ``--ingest auto`` can fall back to the host preparation
    for exactly these and for nothing that goes wrong later in a run (engine, CSV writing).  ``__cause__`` is the original error."""


class refusals:
    """Context manager: a ValueError / NotImplementedError raised inside becomes a ``StreamedRefusal`` (same message, chained)."""

    def __enter__(self):
        return self

    def __exit__(self, etype, e, tb):
        if etype is not None and issubclass(etype, (ValueError, NotImplementedError)) and not issubclass(etype, StreamedRefusal):
            raise StreamedRefusal(f"{etype.__name__}: {e}") from e
        return False


def prepare_streamed(args, varlist: str = "inputs/namelist", app_logger=None, chunk_steps: Optional[int] = None, raw=None) -> StreamedDataset:
    """prepare_data (preprocessing.py:374-413) for the device ingest: validates the file against the namelist and builds
    the index maps; no field data is read here.  ``raw``: the file already opened (``prefers_device_ingest(..., keep_open=True)``: the
    chunk index of a long NetCDF-4 series is then parsed once, not twice)."""
    if getattr(args, "cdsapi", False):
        raise NotImplementedError("--cdsapi downloads need network access and are out of scope")
    if not (getattr(args, "fixed", False) or getattr(args, "track", False)):
        raise NotImplementedError("the device ingest serves the fixed (-f) and the track (-t) frameworks")
    df = ds.read_namelist(varlist, app_logger)
    if raw is None:
        raw = ds.open_raw(args.infile, df, mpas=bool(getattr(args, "mpas", False)), app_logger=app_logger)
    return StreamedDataset(raw, make_plan(raw, args, app_logger), chunk_steps, getattr(args, "inflate", None) or "auto")


AUTO_DEVICE_BYTES = 1 << 30      # --ingest auto: files from this size on are streamed to the device whatever their container


def prefers_device_ingest(args, varlist: str = "inputs/namelist", keep_open: bool = False, app_logger=None):
    """--ingest auto.  True (stream the file's bytes and prepare them on the GPU) for
      * a chunked NetCDF-4 file with at least one DEFLATED field variable, every field variable of which the device can take as it
        lies in the file (``H5Variable.chunk_streams``: fully written, filters within shuffle / deflate / fletcher32): the host path
        inflates on the host's threads, ten times slower than the GPU does (profiles/r04_notes.md section 6);
      * any file of ``AUTO_DEVICE_BYTES`` (1 GiB) or more that the streamed path can read (variables in (time, level, lat, lon) order,
        int8 / int16 / int32 / float32 / float64): the host preparation decodes, sorts and crops the whole data set with NumPy before a
        byte reaches the GPU, the device ingest moves the file's bytes at the link's rate.
    False (prepare on the host) for everything else: small files, other axis orders, a framework the streamed path does not serve.
    Never raises: a file that cannot be judged is left to the host path's messages.  Either way the output files are the same.
    ``keep_open``: return (decision, the opened RawDataset or None) -- on True the caller hands the data set to ``prepare_streamed``."""
    verdict, raw = _prefers_device_ingest(args, varlist, app_logger)
    if keep_open and verdict:
        return True, raw
    if raw is not None:
        raw.close()
    return (verdict, None) if keep_open else verdict


def _prefers_device_ingest(args, varlist, app_logger=None):
    if not (getattr(args, "fixed", False) or getattr(args, "track", False)) or getattr(args, "cdsapi", False):
        return False, None
    try:
        with open(args.infile, "rb") as fh:
            hdf5 = fh.read(8) == b"\x89HDF\r\n\x1a\n"
        big = os.path.getsize(args.infile) >= AUTO_DEVICE_BYTES
        if not (hdf5 or big):
            return False, None
        # (the logger: what open_raw reports -- the -m/--mpas drop of the `standard_height` dimension -- is said once, here, when the
        # opened file is handed on to prepare_streamed)
        raw = ds.open_raw(args.infile, ds.read_namelist(varlist), mpas=bool(getattr(args, "mpas", False)), app_logger=app_logger)
    except Exception:       # noqa: BLE001
        return False, None
    try:
        if big:
            return True, raw
        deflated = False
        for v in raw.variables.values():
            if not hasattr(v.data, "chunk_streams") or v.data.chunk_streams() is None:
                return False, raw
            deflated = deflated or any(fid == 1 for fid, _cd in getattr(v.data, "_filters", []))
        return deflated, raw
    except Exception:       # noqa: BLE001
        return False, raw


def _src_code(dtype: np.dtype) -> int:
    if dtype.kind == "i" and dtype.itemsize in (1, 2, 4):
        return {1: _lib.LEC_I8, 2: _lib.LEC_I16, 4: _lib.LEC_I32}[dtype.itemsize]
    if dtype.kind == "f" and dtype.itemsize == 4:
        return _lib.LEC_F32
    if dtype.kind == "f" and dtype.itemsize == 8:
        return _lib.LEC_F64
    raise ValueError(f"the device ingest reads int8, int16, int32, float32 and float64 variables, not {dtype}")


def _swapped(dtype: np.dtype) -> bool:
    return dtype.byteorder == (">" if sys.byteorder == "little" else "<")


_POOL = None


def _pool():
    """Threads for the page-cache -> pinned-memory copies (one core moves ~5-10 GB/s, PCIe 5 x16 takes ~50)."""
    global _POOL
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1)), thread_name_prefix="lec-stage")
    return _POOL


_PAGE = 4096


class RegisteredSpans:
    """Host memory registered with the HIP runtime for direct (staging-free) copies: a set of page-aligned, NON-OVERLAPPING blocks
    (``lec_host_register`` refuses a span that touches a registered one).  ``ensure(lo, hi, use)`` registers whatever part of
    [lo, hi) is not covered yet and stamps every block it touches with ``use`` (the chunk number); ``release(before)`` unregisters
    the blocks whose last use is older than ``before`` -- the caller has synchronised on those chunks' copies.  One instance serves
    every variable of a file (record variables interleave, so their spans touch)."""

    def __init__(self, lib):
        self.lib = lib
        self.blocks = []                # [start, end, last_use], sorted, disjoint
        self.registered_bytes = 0
        self.calls = 0
        self.seconds = 0.0

    def ensure(self, lo: int, hi: int, use: int) -> None:
        lo, hi = lo & ~(_PAGE - 1), (hi + _PAGE - 1) & ~(_PAGE - 1)
        at, new = lo, []
        for b in self.blocks:
            if b[1] <= lo or b[0] >= hi:
                continue
            b[2] = max(b[2], use)
            if b[0] > at:
                new.append([at, b[0], use])
            at = max(at, b[1])
        if at < hi:
            new.append([at, hi, use])
        t_reg = time.perf_counter()
        try:
            for b in new:
                _lib.check(self.lib.lec_host_register(C.c_void_p(b[0]), b[1] - b[0]), "lec_host_register")
                # tracked as soon as it is registered: if a LATER block of this call is refused (the mid-run refusal the pinned
                # fallback exists for), the blocks already done are released with the rest instead of staying pinned and untracked --
                # where they would make every later ensure() over those pages fail as overlapping
                self.blocks.append(b)
                self.registered_bytes += b[1] - b[0]
                self.calls += 1
        finally:
            self.seconds += time.perf_counter() - t_reg
            self.blocks.sort()

    def pieces(self, lo: int, hi: int):
        """[lo, hi) cut at the boundaries of the registered blocks: a copy must lie inside ONE registered allocation to go out as a
        direct DMA (the runtime treats a range that runs from one allocation into the next as unregistered memory)."""
        out = []
        for b in self.blocks:
            a, e = max(lo, b[0]), min(hi, b[1])
            if a < e:
                out.append((a, e))
        if sum(e - a for a, e in out) != hi - lo:
            raise RuntimeError("copy range is not wholly registered")
        return out

    def release(self, before: int) -> None:
        keep = []
        for b in self.blocks:
            if b[2] < before:
                _lib.check(self.lib.lec_host_unregister(C.c_void_p(b[0])), "lec_host_unregister")
            else:
                keep.append(b)
        self.blocks = keep

    def close(self) -> None:
        self.release(1 << 62)


class _Stager:
    """One variable's path to the GPU: a pinned host buffer and a raw device buffer per pipeline slot.

    Only what the analysis domain touches is staged: the file levels in ``levels`` (ascending file index, so that runs of
    neighbouring levels are single copies when the band is the whole latitude range) and the latitude band [j0, j1] of
    each -- whole longitude rows, which stay contiguous in the file.  A regional box out of a global file moves
    a fraction of the bytes (a 15-degree band of a 0.25-degree grid: 1/12)."""

    def __init__(self, var: ds.RawVariable, steps: int, device, levels: np.ndarray, j0: int, j1: int, slots: int = 2, pinned: bool = True):
        self.var = var
        self.levels = [int(k) for k in levels]
        self.j0, self.j1 = int(j0), int(j1)
        self.nx = int(var.data.shape[3])
        full = self.j0 == 0 and self.j1 == int(var.data.shape[2]) - 1
        self.runs = []                                   # (first staged level, first file level, count): contiguous in the file
        for n, k in enumerate(self.levels):
            if full and self.runs and self.runs[-1][1] + self.runs[-1][2] == k:
                self.runs[-1][2] += 1
            else:
                self.runs.append([n, k, 1])
        self.level_elems = (self.j1 - self.j0 + 1) * self.nx
        self.step_elems = len(self.levels) * self.level_elems
        self.itemsize = var.data.dtype.itemsize
        carrier = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[self.itemsize]      # bytes only; never interpreted
        # (direct copies from registered file memory need no pinned staging buffers)
        self.pinned = [torch.empty((steps, self.step_elems), dtype=carrier, pin_memory=True) for _ in range(slots)] if pinned else None
        self.raw_dev = [torch.empty((steps, self.step_elems), dtype=carrier, device=device) for _ in range(slots)]
        self._carrier_np = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[self.itemsize]

    def stage(self, slot: int, file_steps: np.ndarray, at: int):
        """File time steps -> pinned rows [at, at + len): the only host touch of the data (page cache -> pinned)."""
        dst = self.pinned[slot].numpy()
        if len(file_steps) == 0:
            return
        # one time step of a variable is contiguous in the file even when the time axis is the record dimension
        # (record variables are interleaved per record): never reshape across time, that would copy the variable
        as_bytes = lambda a: a.view(a.dtype.newbyteorder("=")).view(self._carrier_np)     # reinterpret, never convert
        jobs = []
        piece = max(1, (8 << 20) // self.itemsize)                  # ~8 MiB per copy job: memcpy releases the GIL (2 MiB jobs: the
                                                                    # workers queue for the GIL and staging falls from 125 to 41-75 GB/s)
        for r, ft in enumerate(file_steps):
            block = self.var.data[int(ft)]
            if not block.flags["C_CONTIGUOUS"]:
                raise ValueError("the device ingest needs each time step of a variable to be contiguous in the file")
            for n, k, cnt in self.runs:
                src = as_bytes(block[k: k + cnt, self.j0: self.j1 + 1].reshape(-1))
                out = dst[at + r, n * self.level_elems: (n + cnt) * self.level_elems]
                for a in range(0, cnt * self.level_elems, piece):
                    jobs.append((out[a: a + piece], src[a: a + piece]))
        if len(jobs) == 1:
            np.copyto(*jobs[0])
        else:
            list(_pool().map(lambda j: np.copyto(*j), jobs))

    def upload(self, slot: int, a: int, b: int):
        self.raw_dev[slot][a:b].copy_(self.pinned[slot][a:b], non_blocking=True)

    # -- staging-free path: the file's own memory, registered with the HIP runtime ------------------------------------------------
    def direct_ok(self) -> bool:
        """True when every time step of the variable is a C-contiguous block of ordinary (mapped-file or host) memory: a NumPy
        array, not a lazily inflated HDF5 variable."""
        d = self.var.data
        full = self.j0 == 0 and self.j1 == int(d.shape[2]) - 1      # a latitude band of a larger file: registering (= pinning, hence reading) whole
        return full and isinstance(d, np.ndarray) and d.ndim == 4 and d[0].flags["C_CONTIGUOUS"]      # levels would touch the bytes the band spares

    def step_address(self, ft: int) -> int:
        d = self.var.data
        return int(d.ctypes.data) + int(ft) * int(d.strides[0])

    def upload_direct(self, lib, spans: RegisteredSpans, slot: int, file_steps, at: int, use: int, stream) -> int:
        """File time steps -> raw device rows [at, at + len) of the slot, straight from the registered file memory: per step one
        linear copy per run of whole levels (cut where it runs from one registered block into the next)."""
        if len(file_steps) == 0:
            return 0
        d = self.var.data
        plane = int(d.shape[2]) * self.nx * self.itemsize
        step_bytes = int(d.shape[1]) * plane
        bases = [self.step_address(ft) for ft in file_steps]
        first = 0                                                       # one registration per run of neighbouring file steps (what of it is not
        for r in range(1, len(bases) + 1):                              # covered yet): steps a track skips are neither pinned nor read
            if r == len(bases) or int(file_steps[r]) != int(file_steps[r - 1]) + 1:
                spans.ensure(min(bases[first:r]), max(bases[first:r]) + step_bytes, use)
                first = r
        moved = 0
        for r, base in enumerate(bases):
            dst0 = self.raw_dev[slot][at + r].data_ptr()
            for n, k, cnt in self.runs:
                src, dst, nbytes = base + k * plane, dst0 + n * plane, cnt * plane
                for a, e in spans.pieces(src, src + nbytes):
                    _lib.check(lib.lec_copy_rows_async(C.c_void_p(dst + (a - src)), e - a, C.c_void_p(a), e - a, e - a, 1, stream),
                               "lec_copy_rows_async")
                moved += nbytes
        return moved


def chunk_copy_plan(addr: np.ndarray, size: np.ndarray, tail: int = 0, gap: int = 64 << 10):
    """Where the stored chunks at file offsets ``addr`` (``size`` + ``tail`` bytes each) go in a staging buffer, copied as few large
    spans as possible: RUNS of chunks that lie (almost) back to back in the file -- a whole variable written in one go is one run per
    call, a file written step by step (the variables interleaved) one run per step -- each run one span, gaps of up to ``gap`` bytes
    included (per-chunk copy jobs of a few 100 KB leave the thread pool waiting for the GIL: 11 GB/s instead of > 50).  A stream keeps
    its file alignment modulo 16 (``lec_inflate`` takes streams at any byte offset); repeated or overlapping chunks share their bytes.
    Returns (offset of every chunk in the buffer, the runs' file offsets, lengths and buffer offsets, bytes of buffer used)."""
    n = len(addr)
    order = np.argsort(addr, kind="stable")
    a_s, e_s = addr[order], (addr + size + tail)[order]
    reach = np.maximum.accumulate(e_s)                            # a run's end is the furthest byte seen so far
    starts = np.flatnonzero(np.concatenate([[True], a_s[1:] > reach[:-1] + gap]))
    run_lo = a_s[starts]
    run_len = np.concatenate([reach[starts[1:] - 1], reach[-1:]]) - run_lo
    lead = run_lo & 15
    padded = (lead + run_len + 15) & ~15
    base = np.concatenate([[0], np.cumsum(padded)[:-1]]) + lead   # where each run's first byte lands
    run_of = np.searchsorted(starts, np.arange(n), side="right") - 1
    src_off = np.empty(n, dtype=np.int64)
    src_off[order] = base[run_of] + (a_s - run_lo[run_of])
    return src_off, run_lo, run_len, base, int(padded.sum())


class _ChunkStager:
    """The path of a DEFLATED NetCDF-4 variable to the GPU: the compressed chunks cross the link as they lie in the file and
    ``lec_inflate`` (one wave per chunk) + ``lec_chunk_scatter`` (un-shuffle, chunk tiling) rebuild the raw sub-cube on the device --
    the same [steps][staged levels][band rows][nx] layout a ``_Stager`` uploads, so everything downstream is unchanged.  The host
    only copies compressed bytes (page cache -> pinned); the reference's netCDF4 / HDF5 stack inflates them on one thread.

    ``stage`` / ``upload`` have ``_Stager``'s signatures; the status words of every launch are fetched asynchronously and checked by
    ``check`` (before a slot is reused and at the end of the run): a corrupt chunk raises ``hdf5_lite.Hdf5Error`` naming it."""

    def __init__(self, var: ds.RawVariable, info: dict, steps: int, device, levels: np.ndarray, j0: int, j1: int, slots: int = 2,
                 spans: Optional["RegisteredSpans"] = None):
        """``spans``: the run's RegisteredSpans -- the compressed chunks are then copied to the GPU straight from the mapped FILE's pages
        (registered with the HIP runtime run by run, released two chunks later): no pinned staging buffers (pinning fresh memory
        costs ~0.12 s per GB: 1.5 s for the 12 GB a 96-step ERA5 batch wants) and no host copy."""
        self.var, self.info = var, info
        self.spans, self.use, self.runs = spans, 0, [None] * slots
        self.levels = [int(k) for k in levels]
        self.j0, self.j1 = int(j0), int(j1)
        shape = tuple(int(x) for x in var.data.shape)
        self.shape = shape
        self.nx = shape[3]
        self.itemsize = var.data.dtype.itemsize
        self.level_elems = (self.j1 - self.j0 + 1) * self.nx
        self.step_elems = len(self.levels) * self.level_elems
        self.chunk = tuple(int(c) for c in info["chunk"])
        ct, ck, cj, ci = self.chunk
        self.chunk_bytes = ct * ck * cj * ci * self.itemsize
        self.slot16 = (self.chunk_bytes + 15) & ~15
        # the chunk columns every step needs: levels x latitude band x all longitudes
        kk = sorted({k // ck for k in self.levels})
        jj = range(self.j0 // cj, self.j1 // cj + 1)
        ii = range(-(-shape[3] // ci))
        self.space = np.array([(k * ck, j * cj, i * ci) for k in kk for j in jj for i in ii], dtype=np.int64)
        self._tc = {}                                          # time-chunk -> (file offsets, stored sizes, stored-as-is flags) of its chunks
        n_all = -(-shape[0] // ct)
        n_tc = min(n_all, steps // ct + 2)                     # time-chunks that `steps` consecutive steps touch (stage() grows the buffers for more)
        # staging capacity: the `steps` largest time-chunks (sampled when the series is long), with room for the gaps of a
        # span copy (see stage) and the padding of the last stream
        sample = range(n_all) if n_all <= 64 else sorted(set(np.linspace(0, n_all - 1, 64).astype(int).tolist()))
        per_tc = sorted((int(((self._time_chunk(tc)[1] + 15) & ~15).sum()) for tc in sample), reverse=True)
        worst = int(1.35 * sum(per_tc[:n_tc]) * (1.0 if n_all <= 64 else 1.15)) + (4 << 20)
        # buffers are sized for consecutive steps (steps / ct + 2 time-chunks); a track that picks scattered steps out of chunks that
        # span several steps needs more -- stage() grows the slot then
        self.max_chunks = [min(n_all, steps // ct + 2) * len(self.space)] * slots
        dev = torch.device(device)
        self.device = dev
        carrier = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[self.itemsize]
        self.raw_dev = [torch.empty((steps, self.step_elems), dtype=carrier, device=dev) for _ in range(slots)]
        self.comp_pin = [None if spans is not None else torch.empty(worst, dtype=torch.uint8, pin_memory=True) for _ in range(slots)]
        self.comp_dev = [torch.empty(worst, dtype=torch.uint8, device=dev) for _ in range(slots)]
        self.inflated = [torch.empty(self.max_chunks[0] * self.slot16 + 16, dtype=torch.uint8, device=dev) for _ in range(slots)]
        # per chunk: the lec_inflate descriptor (4 int64) and the lec_chunk_scatter record (5 int64), one upload
        self.meta_pin = [torch.zeros(self.max_chunks[0] * 9, dtype=torch.int64, pin_memory=True) for _ in range(slots)]
        self.meta_dev = [torch.zeros(self.max_chunks[0] * 9, dtype=torch.int64, device=dev) for _ in range(slots)]
        self.status_dev = [torch.zeros((self.max_chunks[0], 4), dtype=torch.int32, device=dev) for _ in range(slots)]
        self.status_pin = [torch.zeros((self.max_chunks[0], 4), dtype=torch.int32, pin_memory=True) for _ in range(slots)]
        self.tmap_pin = [torch.zeros(steps * max(ct, 1) + 2 * ct + 8, dtype=torch.int32, pin_memory=True) for _ in range(slots)]
        self.tmap_dev = [torch.zeros_like(t, device=dev) for t in self.tmap_pin]
        kmap = np.full(shape[1], -1, dtype=np.int32)
        kmap[self.levels] = np.arange(len(self.levels), dtype=np.int32)
        self.kmap_dev = torch.as_tensor(kmap).to(dev)
        self.pending = [None] * slots                          # (n chunks, their origins) of the launch whose status is in flight
        self.staged = [None] * slots
        self.fetched = [torch.cuda.Event() for _ in range(slots)]
        self.compressed_bytes = 0
        # shuffled data repeat from one longitude row to the next, ci bytes apart within a byte plane: close enough for the short LDS ring
        self.short_window = bool(info["shuffle"]) and ci <= 3000
        self.view = np.frombuffer(info["map"], dtype=np.uint8)
        # its own stream: a launch lasts as long as ONE chunk takes one wave (tens of ms for a 0.5 MB chunk) however few chunks it
        # holds, so the five variables of a step must inflate side by side, not one after the other on the copy stream
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(slots)]      # per slot: consecutive batches overlap too (the tail
                                                                                   # of one batch leaves most of the GPU idle)

    def direct_ok(self) -> bool:
        return False

    def _time_chunk(self, tc: int):
        got = self._tc.get(tc)
        if got is None:
            table, t0 = self.info["table"], tc * self.chunk[0]
            if hasattr(table, "lookup"):                   # the index kept as arrays (hdf5_lite.ChunkTable): one vectorised lookup per time-chunk
                org = np.concatenate([np.full((len(self.space), 1), t0, dtype=np.int64), self.space], axis=1)
                a, b, c = table.lookup(org)
                got = self._tc[tc] = (a.astype(np.int64), b.astype(np.int64), c.astype(bool))
            else:
                rows = [table[(t0, int(k), int(j), int(i))] for k, j, i in self.space]
                got = self._tc[tc] = (np.array([r[0] for r in rows], dtype=np.int64), np.array([r[1] for r in rows], dtype=np.int64),
                                      np.array([r[2] for r in rows], dtype=bool))
        return got

    def stage(self, slot: int, file_steps: np.ndarray, at: int):
        """The compressed chunks that hold ``file_steps`` -> the slot's pinned buffer (thread pool), descriptors beside them."""
        self.check(slot)
        if len(file_steps) == 0:
            self.staged[slot] = None
            return
        ct = self.chunk[0]
        steps = [int(t) for t in file_steps]
        tcs = sorted({t // ct for t in steps})
        parts = [self._time_chunk(tc) for tc in tcs]
        addr, size, plain = (np.concatenate([p[i] for p in parts]) for i in range(3))
        m = len(self.space)
        n = m * len(tcs)
        if n > self.max_chunks[slot]:              # (the slot's last launch has completed: check() above)
            dev = self.device
            self.max_chunks[slot] = n
            self.inflated[slot] = torch.empty(n * self.slot16 + 16, dtype=torch.uint8, device=dev)
            self.meta_pin[slot] = torch.zeros(n * 9, dtype=torch.int64, pin_memory=True)
            self.meta_dev[slot] = torch.zeros(n * 9, dtype=torch.int64, device=dev)
            self.status_dev[slot] = torch.zeros((n, 4), dtype=torch.int32, device=dev)
            self.status_pin[slot] = torch.zeros((n, 4), dtype=torch.int32, pin_memory=True)
        origins = np.empty((n, 4), dtype=np.int64)
        origins[:, 0] = np.repeat(np.array(tcs, dtype=np.int64) * ct, m)
        origins[:, 1:] = np.tile(self.space, (len(tcs), 1))
        view = self.view
        tail = 4 if self.info.get("fletcher32") else 0           # the checksum bytes after each stream travel with it
        src_off, run_lo, run_len, base, need = chunk_copy_plan(addr, size, tail)
        if need + 2048 > self.comp_dev[slot].numel():           # larger than the sampled time-chunks (or many gaps): grow this slot's buffers
            grown = int(1.25 * need) + (4 << 20)                # (its last launch has completed: check() above)
            if self.spans is None:
                self.comp_pin[slot] = torch.empty(grown, dtype=torch.uint8, pin_memory=True)
            self.comp_dev[slot] = torch.empty(grown, dtype=torch.uint8, device=self.device)
        used = need
        piece = 8 << 20
        self.runs[slot] = (run_lo.tolist(), run_len.tolist(), base.tolist()) if self.spans is not None else None    # direct mode: upload() copies them
        jobs = []
        if self.spans is None:
            if self.comp_pin[slot] is None or self.comp_pin[slot].numel() < self.comp_dev[slot].numel():      # (a run that fell back from direct copies)
                self.comp_pin[slot] = torch.empty(self.comp_dev[slot].numel(), dtype=torch.uint8, pin_memory=True)
            comp = self.comp_pin[slot].numpy()
            jobs = [(comp[d + a: d + min(a + piece, ln)], view[lo + a: lo + min(a + piece, ln)])
                    for lo, ln, d in zip(run_lo.tolist(), run_len.tolist(), base.tolist()) for a in range(0, ln, piece)]
        meta = self.meta_pin[slot].numpy()
        desc, recs = meta[: 4 * n].reshape(n, 4), meta[4 * n: 9 * n].reshape(n, 5)
        slots16 = np.arange(n, dtype=np.int64) * self.slot16
        desc[:, 0], desc[:, 1], desc[:, 2], desc[:, 3] = src_off, np.where(plain, -size, size), slots16, self.chunk_bytes
        recs[:, 0], recs[:, 1:] = slots16, origins
        if len(jobs) == 1:
            np.copyto(*jobs[0])
        elif jobs:
            list(_pool().map(lambda j: np.copyto(*j), jobs))
        at_byte = used
        t_base = tcs[0] * ct
        n_tmap = (tcs[-1] + 1) * ct - t_base
        if n_tmap > self.tmap_pin[slot].numel():                 # a track that takes every n-th file step spans many more file steps than it stages
            self.tmap_pin[slot] = torch.zeros(n_tmap + 64, dtype=torch.int32, pin_memory=True)
            self.tmap_dev[slot] = torch.zeros(n_tmap + 64, dtype=torch.int32, device=self.device)
        tmap = self.tmap_pin[slot].numpy()
        tmap[:n_tmap] = -1
        for r, t in enumerate(steps):
            tmap[t - t_base] = at + r
        self.staged[slot] = (n, at_byte, t_base, n_tmap, origins)
        self.compressed_bytes += at_byte

    def _copy_from_file(self, lib, slot: int, stream) -> None:
        """The slot's runs of compressed chunks, file pages -> device, no host copy."""
        file0, dst0 = int(self.view.ctypes.data), self.comp_dev[slot].data_ptr()
        # registrations cost ~1 ms each: a latitude band out of a global file is thousands of 0.5-MB runs with the other bands'
        # chunks between them -- runs less than 4 MiB apart are registered as one span (the bytes between are pinned, not copied)
        order = sorted(range(len(self.runs[slot][0])), key=lambda q: self.runs[slot][0][q])
        glo = ghi = None
        for q in order:
            lo, hi = file0 + self.runs[slot][0][q], file0 + self.runs[slot][0][q] + self.runs[slot][1][q]
            if glo is not None and lo - ghi <= (4 << 20):
                ghi = max(ghi, hi)
                continue
            if glo is not None:
                self.spans.ensure(glo, ghi, self.use)
            glo, ghi = lo, hi
        if glo is not None:
            self.spans.ensure(glo, ghi, self.use)
        for lo, ln, d in zip(*self.runs[slot]):
            for a, e in self.spans.pieces(file0 + lo, file0 + lo + ln):
                _lib.check(lib.lec_copy_rows_async(C.c_void_p(dst0 + d + (a - file0 - lo)), e - a, C.c_void_p(a), e - a, e - a, 1, stream),
                           "lec_copy_rows_async")

    def upload(self, slot: int, a: int, b: int):
        """Enqueued on the current stream: compressed bytes + descriptors to the device, inflate, scatter into rows [a, b)."""
        st = self.staged[slot]
        if st is None:
            return
        n, nbytes, t_base, n_tmap, origins = st
        lib = _lib.load()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        used = nbytes + 1024
        if self.spans is not None:
            try:
                self._copy_from_file(lib, slot, stream)
            except _lib.LecLibraryError:
                # the runtime refused a registration in mid-run (a locked-memory limit, say): this variable goes on through pinned
                # staging buffers -- the spans registered so far stay with the run's RegisteredSpans and are released by it
                self.spans = None
        if self.spans is None:
            if self.comp_pin[slot] is None or self.comp_pin[slot].numel() < self.comp_dev[slot].numel():
                self.comp_pin[slot] = torch.empty(self.comp_dev[slot].numel(), dtype=torch.uint8, pin_memory=True)
            if self.runs[slot] is not None:              # staged for a direct copy: do the host copy now
                comp, view, piece = self.comp_pin[slot].numpy(), self.view, 8 << 20
                jobs = [(comp[d + q: d + min(q + piece, ln)], view[lo + q: lo + min(q + piece, ln)])
                        for lo, ln, d in zip(*self.runs[slot]) for q in range(0, ln, piece)]
                list(_pool().map(lambda j: np.copyto(*j), jobs))
            self.comp_dev[slot][:used].copy_(self.comp_pin[slot][:used], non_blocking=True)
        self.meta_dev[slot][: 9 * n].copy_(self.meta_pin[slot][: 9 * n], non_blocking=True)
        self.tmap_dev[slot][:n_tmap].copy_(self.tmap_pin[slot][:n_tmap], non_blocking=True)
        desc, recs = self.meta_dev[slot][: 4 * n], self.meta_dev[slot][4 * n: 9 * n]
        with torch.cuda.device(self.device):
            ia = _lib.InflateArgs(src_d=self.comp_dev[slot].data_ptr(), src_bytes=used, desc_d=desc.data_ptr(), n_streams=n,
                                  flags=int(bool(self.info.get("fletcher32"))) | (2 if self.short_window else 0),
                                  dst_d=self.inflated[slot].data_ptr(), status_d=self.status_dev[slot].data_ptr(), stream=stream,
                                  dst_bytes=self.inflated[slot].numel())
            _lib.check(lib.lec_inflate(C.byref(ia)), "lec_inflate")
            ct, ck, cj, ci = self.chunk
            sa = _lib.ChunkScatterArgs(src_d=self.inflated[slot].data_ptr(), chunk_d=recs.data_ptr(), n_chunks=n, elem_size=self.itemsize,
                                       shuffled=int(self.info["shuffle"]), ct=ct, ck=ck, cj=cj, ci=ci, t_base=t_base, n_tmap=n_tmap,
                                       n_kmap=self.shape[1], j0=self.j0, tmap_d=self.tmap_dev[slot].data_ptr(), kmap_d=self.kmap_dev.data_ptr(),
                                       nt=int(self.raw_dev[slot].shape[0]), nl=len(self.levels), ny=self.j1 - self.j0 + 1, nx=self.nx,
                                       out_d=self.raw_dev[slot].data_ptr(), stream=stream, src_bytes=self.inflated[slot].numel())
            _lib.check(lib.lec_chunk_scatter(C.byref(sa)), "lec_chunk_scatter")
        self.status_pin[slot][:n].copy_(self.status_dev[slot][:n], non_blocking=True)
        self.fetched[slot].record(torch.cuda.current_stream(self.device))
        self.pending[slot] = (n, origins)

    def check(self, slot: int):
        """Waits for the slot's last launch and raises if a chunk did not inflate."""
        p = self.pending[slot]
        if p is None:
            return
        self.pending[slot] = None
        self.fetched[slot].synchronize()
        n, origins = p
        st = self.status_pin[slot][:n, 0].numpy()
        bad = np.flatnonzero(st != 0)
        if bad.size:
            from .hdf5_lite import Hdf5Error
            c = int(bad[0])
            what = _lib.load().lec_inflate_status_text(int(st[c])).decode()
            raise Hdf5Error(f"deflated chunk at {tuple(int(x) for x in origins[c])} (time, level, lat, lon) did not inflate on the device: {what} "
                            f"({bad.size} of {n} chunks)")

    def finish(self):
        for slot in range(len(self.pending)):
            self.check(slot)


def _make_stager(var: ds.RawVariable, steps: int, device, levels, j0: int, j1: int, slots: int, pinned: bool, inflate: str, spans=None):
    """``inflate``: "device" -- deflated NetCDF-4 variables are inflated on the GPU (error if the variable is not one);
    "host" -- the pure-Python reader's thread pool inflates them; "auto" (default): device where the variable allows it."""
    if inflate not in ("auto", "host", "device"):
        raise ValueError("inflate must be 'auto', 'host' or 'device'")
    info = var.data.chunk_streams() if (inflate != "host" and hasattr(var.data, "chunk_streams")) else None
    if info is not None:
        return _ChunkStager(var, info, steps, device, levels, j0, j1, slots, spans=spans)
    if inflate == "device":
        raise ValueError("inflate='device' needs a fully written, deflated (optionally shuffled / checksummed) chunked NetCDF-4 variable")
    return _Stager(var, steps, device, levels, j0, j1, slots, pinned=pinned)


def _lec_code(dtype: np.dtype) -> int:
    return _lib.LEC_F64 if np.dtype(dtype) == np.float64 else _lib.LEC_F32


def storage_dtypes(rvars) -> tuple:
    """Per-variable decode dtype (the reference's xarray decode rules, dataset.decode_dtypes) and the one storage dtype of the
    cubes: the widest of them (widening is exact; the engine wants one dtype, and all arithmetic is fp64 anyway)."""
    dec = {r: ds.decode_dtypes(v.data.dtype, v.scale_factor, v.add_offset, v.fill_value)[1] for r, v in rvars.items()}
    return dec, np.result_type(*dec.values())


def _ingest_call(lib, v: ds.RawVariable, src_ptr: int, nt: int, geom, maps, unit: float, decode_dtype, out_dtype, out_ptr: int, stream,
                 step: Optional[torch.Tensor] = None, step_base: int = 0, nt_src: int = 0):
    """``step``: int32 [nt, 3] on the device -- per output step {source step, latitude offset, longitude offset} into the maps
    (``lec_ingest_args.step_d``: a box-packed series); ``step_base``: the source step ``src_ptr`` starts with, ``nt_src``: the steps it
    holds (with the maps' lengths: what bounds the table's entries on the device).  ``LEC_CHECK_TABLES=1`` (debug runs of the Python
    host): ``lec_check_maps`` scans the maps and the table before every call -- synchronous, so not the default."""
    nl_in, ny_in, nx_in, nl, ny, nx = geom
    kmap, jmap, imap = maps
    ga = _lib.IngestArgs(
        src_d=C.c_void_p(src_ptr), src_dtype=_src_code(v.data.dtype), swap_bytes=int(_swapped(v.data.dtype)),
        nt=nt, nl_in=nl_in, ny_in=ny_in, nx_in=nx_in, nl=nl, ny=ny, nx=nx,
        kmap_d=C.c_void_p(kmap.data_ptr()), jmap_d=C.c_void_p(jmap.data_ptr()), imap_d=C.c_void_p(imap.data_ptr()),
        has_packing=int(v.scale_factor is not None or v.add_offset is not None), has_fill=int(v.fill_value is not None),
        scale_factor=1.0 if v.scale_factor is None else v.scale_factor, add_offset=0.0 if v.add_offset is None else v.add_offset,
        fill_value=0.0 if v.fill_value is None else v.fill_value, unit_scale=float(unit),
        out_dtype=_lec_code(out_dtype), decode_dtype=_lec_code(decode_dtype),
        out_d=C.c_void_p(out_ptr), stream=C.c_void_p(stream.cuda_stream),
        step_d=None if step is None else C.c_void_p(step.data_ptr()), step_base=int(step_base), nt_src=int(nt_src),
        jmap_len=0 if step is None else int(jmap.numel()), imap_len=0 if step is None else int(imap.numel()))
    if os.environ.get("LEC_CHECK_TABLES", "0") == "1":
        status = torch.empty(4, dtype=torch.int32, device=kmap.device)
        _lib.check(lib.lec_check_maps(C.byref(ga), C.c_void_p(status.data_ptr())), "lec_check_maps")
    _lib.check(lib.lec_ingest(C.byref(ga)), "lec_ingest")


def device_cube(var: ds.RawVariable, plan: IngestPlan, device="cuda:0", unit: float = 1.0, out_dtype=None, inflate: str = "auto") -> torch.Tensor:
    """One variable of the analysis domain, [time, level, lat, lon] on the device, through the same staging + ``lec_ingest``
    path the streamed frameworks use (all time steps at once: for tests and small domains)."""
    lib = _lib.load()
    dev = torch.device(device)
    nt, nl, ny, nx = len(plan.tsel), plan.level.size, plan.lat.size, plan.lon.size
    decode = ds.decode_dtypes(var.data.dtype, var.scale_factor, var.add_offset, var.fill_value)[1]
    out_dtype = np.dtype(decode if out_dtype is None else out_dtype)
    j0, j1 = int(plan.jmap.min()), int(plan.jmap.max())
    file_levels = np.sort(plan.kmap)
    st = _make_stager(var, nt, dev, file_levels, j0, j1, 1, True, inflate)
    st.stage(0, plan.tsel, 0)
    with torch.cuda.device(dev):
        st.upload(0, 0, nt)
    up = lambda a: torch.as_tensor(a, dtype=torch.int32).to(dev)
    maps = (up(np.searchsorted(file_levels, plan.kmap)), up(plan.jmap - j0), up(plan.imap))
    out = torch.empty((nt, nl, ny, nx), dtype=torch.float64 if out_dtype == np.float64 else torch.float32, device=dev)
    with torch.cuda.device(dev):
        _ingest_call(lib, var, st.raw_dev[0].data_ptr(), nt, (nl, j1 - j0 + 1, int(var.data.shape[3]), nl, ny, nx), maps, unit,
                     decode, out_dtype, out.data_ptr(), torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    if hasattr(st, "finish"):
        st.finish()
    return out


def lec_streamed(raw: ds.RawDataset, plan: IngestPlan, variable_list_df, boxes_limits, *, per_step_boxes: bool = False,
                 device="cuda:0", chunk_steps: Optional[int] = None, with_q: bool = True, stats: Optional[dict] = None,
                 t_range=None, merge_dropmask=None, out=None, staging: str = "auto", inflate: str = "auto",
                 slots: Optional[int] = None, keep_level: Optional[float] = None, packed: Optional[bool] = None) -> LECResult:
    """All LEC terms for the whole series, streamed from the memory-mapped file.

    ``boxes_limits``: one (west, east, south, north) in degrees (fixed framework, as inputs/box_limits) or one per time step
    (``per_step_boxes``: the moving framework; dT/dt is differentiated over the plan's time axis on the device).
    ``chunk_steps`` time steps are resident per pipeline slot (``slots``: default two, three with the device inflate).  ``stats`` receives counters: bytes moved, chunks, dtype.
    ``staging``: "registered" -- the mapped file's own pages are registered with the HIP runtime chunk by chunk and copied from
    directly (no host copy, no staging threads: with eight ranks streaming at once the host's memory carries a third of the traffic);
    "staged" -- a thread pool copies page cache -> pinned buffers first; "auto" (default): registered where every variable is plain
    mapped memory and the runtime accepts the first span, else staged (lazily inflated NetCDF-4 variables, hosts that refuse).
    ``inflate``: where deflated NetCDF-4 chunks are inflated -- "device" (``lec_inflate``: the link carries the compressed bytes),
    "host" (the reader's thread pool), "auto" = device wherever the variable allows it (``_make_stager``).
    ``keep_level`` (Pa): the decoded u, v and geopotential slices of that level are kept on the device for the processed steps (the
    moving framework's 850-hPa diagnostics: no second pass over the file) -- ``stats["level_slices"]`` = {"u", "v", "geopt"} [steps, lat, lon].
    ``packed`` (default: the moving framework with Q): ``lec_ingest`` gathers, per time step, only that step's BOX out of the raw
    sub-cube (and T of the two neighbouring steps on it) -- the reference's per-step slice (box_data.py:297-310) done where the data
    are decoded anyway: a tenth of the decoded bytes of a track-extent crop, and stage 1 reads a dense block per step and level
    instead of row fragments (include/lec_hip.h "box-packed series"; the same records bit for bit).
    ``t_range`` = (t0, t1): a rank's share of a time-sharded run -- only those steps (and their one-step T halo) are staged, copied
    and computed, so N ranks move 1/N of the bytes each, over N host links; ``merge_dropmask`` / ``out``: see ``LECEngine.reduce``.
    """
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.LecLibraryError("the device ingest needs a GPU: there is no CPU path")
    t_enter = time.perf_counter()
    mem_enter = torch.cuda.memory_allocated(dev)
    engine = LECEngine(plan.lat, plan.lon, plan.level, device=dev)
    nt, nl, ny, nx = len(plan.tsel), plan.level.size, plan.lat.size, plan.lon.size
    boxes = [engine.box_from_limits(*lim) for lim in boxes_limits]
    if len(boxes) != (nt if per_step_boxes else 1):
        raise ValueError("boxes_limits: one box, or one per time step with per_step_boxes")
    if with_q and nt < 2:
        raise ValueError("dT/dt by finite differences needs at least 2 time steps")
    t0, t1 = (0, nt) if t_range is None else (int(t_range[0]), int(t_range[1]))
    if not (0 <= t0 < t1 <= nt):
        raise ValueError("t_range outside the series")
    geo_role = raw.geo_role
    roles = list(_ROLE_KEYS) + [geo_role]
    keys = {**_ROLE_KEYS, geo_role: "geopt"}
    rvars = {r: raw.variables[raw.names[r]] for r in roles}
    decode, common = storage_dtypes(rvars)
    out_dtype = torch.float64 if common == np.float64 else torch.float32
    on_gpu = inflate != "host" and any(getattr(v.data, "chunk_streams", lambda: None)() is not None for v in rvars.values())
    slots_given = slots is not None
    if slots is None:
        # pipeline slots: two keep a copy-bound pipeline full; a launch of the device inflate lasts as long as its slowest chunk
        # (tens of ms), so a third slot lets the tail of one batch overlap the next two -- where the buffers stay small (below)
        slots = 3 if on_gpu else 2
    if slots < 2:
        raise ValueError("slots must be >= 2")
    if chunk_steps is None:
        chunk_steps = 8
        if on_gpu:
            # A launch of the device inflate lasts as long as its slowest chunk however few chunks it holds, and the GPU takes ~4600
            # of them at a time: a batch should hold ~13000 (files with one chunk per level need more steps for that than files with
            # six), as far as 45 % of the free device memory allows for the three slots' buffers.
            jb0, jb1 = int(plan.jmap.min()), int(plan.jmap.max())
            per_step = 0
            for v in rvars.values():
                info = getattr(v.data, "chunk_streams", lambda: None)()
                if info is not None:
                    ct, ck, cj, ci = info["chunk"]
                    per_step += (len({int(k) // ck for k in plan.kmap}) * (jb1 // cj - jb0 // cj + 1) * -(-int(v.data.shape[3]) // ci)) / ct
            sub = nl * (jb1 - jb0 + 1) * int(rvars["Air Temperature"].data.shape[3])
            per_slot_step = sum(2 * sub * v.data.dtype.itemsize for v in rvars.values())      # compressed + inflated + raw sub-cube, per slot
            decoded_step = 5 * nl * ny * nx * (8 if common == np.float64 else 4)               # the decoded cubes: ONE set (below)
            avail = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)      # free + what torch caches
            by_memory = int(0.45 * avail / (slots * per_slot_step + decoded_step)) - 2
            chunk_steps = int(np.clip(-(-13000 // max(int(per_step), 1)), 8, max(8, min(32, by_memory))))
            # ... or a third of the series, whichever is smaller: a short series should run its three slots round, not size each of
            # them for most of the file (device buffers are paid for per byte: 10-17 ms per GB of hipMalloc, profiles/r05_notes.md)
            chunk_steps = min(chunk_steps, max(8, -(-(t1 - t0) // slots)))
            # Two slots instead of three where three would hold more than ~40 GB: measured on the 96-step global ERA5 box (chunks of
            # 21 steps), 62 GB and 1.39 s against 86 GB and 1.33 s -- while SMALLER chunks cost far more than they save (12 steps:
            # 1.72 s, 8 steps: 2.06 s) -- and on some boxes the driver makes a process wait ~27 ms per GB it allocates beyond the
            # first ~35 GB (profiles/r05_notes.md section 4: 1.4 s of `setup` for 86 GB, 0.1-0.6 s for 45 GB, none on other boxes).
            if not slots_given and slots * (min(chunk_steps, t1 - t0) + 2) * per_slot_step > 40e9:
                slots = 2
    chunk_steps = max(1, min(int(chunk_steps), t1 - t0))
    span = chunk_steps + 2                                                # own steps + the one-step halo of T either side
    # staged sub-cube of every file time step: the kept levels (already in output order) x the latitude band of the domain
    j0, j1 = int(plan.jmap.min()), int(plan.jmap.max())
    nl_in, ny_in, nx_in = nl, j1 - j0 + 1, int(rvars["Air Temperature"].data.shape[3])
    file_levels = np.sort(plan.kmap)
    if staging not in ("auto", "staged", "registered"):
        raise ValueError("staging must be 'auto', 'staged' or 'registered'")
    probe = {r: _Stager.__new__(_Stager) for r in roles}
    for r in roles:
        probe[r].var, probe[r].j0, probe[r].j1 = rvars[r], j0, j1
    direct = staging != "staged" and all(probe[r].direct_ok() for r in roles)
    spans = RegisteredSpans(lib) if direct else None
    if direct and staging == "auto":            # does this host's runtime register this memory at all?  (one page of the first variable)
        try:
            a0 = probe[roles[0]].step_address(int(plan.tsel[t0]))
            spans.ensure(a0, a0 + 1, -1)
            spans.release(0)
        except _lib.LecLibraryError:
            direct, spans = False, None
    # deflated variables: their compressed chunks go out straight from the file's own pages too (one RegisteredSpans for the run)
    chunked = [r for r in roles if inflate != "host" and getattr(rvars[r].data, "chunk_streams", lambda: None)() is not None]
    chunk_direct = bool(chunked) and staging != "staged"
    if chunk_direct and spans is None:
        try:
            spans = RegisteredSpans(lib)
            m0 = int(np.frombuffer(rvars[chunked[0]].data.chunk_streams()["map"], dtype=np.uint8).ctypes.data)
            spans.ensure(m0, m0 + 1, -1)
            spans.release(0)
        except (_lib.LecLibraryError, ValueError, TypeError):
            chunk_direct, spans = False, None
    if staging == "registered" and not direct and not (chunk_direct and len(chunked) == len(roles)):
        raise ValueError("staging='registered' needs every variable as plain (mapped) memory or as device-inflated chunks; variables that are "
                         "inflated on the host are staged")
    if not direct and not chunk_direct and chunked:
        # pinned staging buffers are paid for per byte (pinning): a short series should reuse its slots, not size them for one use each
        chunk_steps = max(1, min(chunk_steps, max(8, -(-(t1 - t0) // (2 * slots)))))
        span = chunk_steps + 2
    stagers = {r: _make_stager(rvars[r], span, dev, file_levels, j0, j1, slots, not direct, inflate, spans if chunk_direct else None) for r in roles}
    # ONE set of decoded cubes, not one per slot: lec_ingest writes them and lec_rowstats reads them on the same (compute) stream, chunk
    # after chunk in order, so the next chunk's decode cannot overtake this chunk's row pass; what the slots double-buffer is the RAW
    # side (uploads and the device inflate run ahead on their own streams).  Rounds 1-4 held `slots` sets: a third of the 114 GB a
    # global ERA5 box allocated for a 37-GB series.  (All five are `span` steps long: the kernels index every field with the same time
    # index, and only T's halo rows are ever written in the others.)
    # ... and they hold a SUB-CHUNK of `dec_steps` steps (about 1 GiB per field): a chunk's raw sub-cubes are decoded and row-passed
    # piece by piece (the raw side keeps its large chunks -- the device inflate wants ~13000 streams per batch --, the decoded side
    # does not need them: 8 steps of a 37 x 721 x 1440 grid are 213,000 rows per launch).  Same kernels on the same rows: same bits.
    bt, _ = engine._box_tables(boxes)            # the row count of the records is the tallest box of the WHOLE series, on every rank
    if packed is None:
        packed = bool(per_step_boxes and with_q)
    if packed and not (per_step_boxes and with_q):
        raise ValueError("packed: the moving framework (per_step_boxes) with Q")
    esize = 8 if common == np.float64 else 4
    up = lambda a: torch.as_tensor(a, dtype=torch.int32).to(dev)
    kmap_rel, jmap_rel, imap_rel = np.searchsorted(file_levels, plan.kmap), plan.jmap - j0, plan.imap   # maps into the staged sub-cube
    if packed:
        # slabs of the tallest x widest box per step; the maps are lengthened by a slab (their last entry repeated) so that a box
        # that ends at the domain's edge can be gathered with the slab's extents -- what lies beside a box is never read by stage 1
        nyp, nxp = bt.nyb_max, bt.nxb_max
        dec_steps = min(chunk_steps, max(4, (1 << 30) // (nl * nyp * nxp * esize)))
        cubes = {k: torch.empty((dec_steps, nl, nyp, nxp), dtype=out_dtype, device=dev) for k in list(keys.values()) + ["tm", "tp"]}
        dtdt = torch.empty((dec_steps, nl, nyp, nxp), dtype=torch.float64, device=dev) if common == np.float64 else None
        maps = (up(kmap_rel), up(np.concatenate([jmap_rel, np.repeat(jmap_rel[-1:], nyp)])), up(np.concatenate([imap_rel, np.repeat(imap_rel[-1:], nxp)])))
        # per output step {source step, where its box starts in the latitude / longitude maps}: for the fields, and for T of the two
        # neighbouring steps on the SAME box (the step itself where the series ends: its coefficient is 0).  Uploaded once, like the
        # d/dt coefficients: nothing inside the chunk loop makes the host wait for the GPU.
        org = np.array([(b[2], b[0]) for b in boxes], dtype=np.int64)
        steps_of = lambda shift: up(np.column_stack([np.clip(np.arange(nt) + shift, 0, nt - 1), org]))
        step_tab = {0: steps_of(0), -1: steps_of(-1), 1: steps_of(1)}
    else:
        dec_steps = min(chunk_steps, max(4, (1 << 30) // (nl * ny * nx * esize)))
        cubes = {keys[r]: torch.empty((dec_steps + 2, nl, ny, nx), dtype=out_dtype, device=dev) for r in roles}
        maps = (up(kmap_rel), up(jmap_rel), up(imap_rel))
    own_boxes = boxes[t0:t1] if per_step_boxes else boxes
    if per_step_boxes:
        own_boxes = engine.prepare_boxes(own_boxes, nyb_min=bt.nyb_max, packed=packed)
    else:
        fixed_box = engine.prepare_boxes(boxes, nyb_min=bt.nyb_max)
    # Row records live for one chunk only (6.8 MB per 37 x 721 time step: a month of hourly steps would be 5 GB, 30 k steps all of
    # HBM): every chunk's records go through the level half of stage 2 at once and leave 12 KB per step in `levraw`, which is what
    # the any-time NaN mask and the pressure integrals of the WHOLE series need at the end (energy_contents.py:190-208).  One buffer,
    # not one per slot: stage 1 and the level stage of consecutive chunks are on the same stream.
    keep, k_keep = None, None
    if keep_level is not None:
        k_keep = int(np.flatnonzero(plan.level == float(keep_level))[0])
        keep = {k: torch.empty((t1 - t0, ny, nx), dtype=out_dtype, device=dev) for k in ("u", "v", "geopt")}
    rows = torch.empty((dec_steps, nl, bt.nyb_max, _lib.LEC_NSTAT), dtype=torch.float64, device=dev)
    levraw = torch.empty((t1 - t0, nl, _lib.LEC_NLEVRAW), dtype=torch.float64, device=dev)
    time_s = plan.time_s
    phi_scale = ds.field_scale(variable_list_df, geo_role)

    t_setup = time.perf_counter()               # (tables, staging / device buffers allocated)
    mem_setup = torch.cuda.memory_allocated(dev) - mem_enter
    compute = torch.cuda.current_stream(dev)
    copier = torch.cuda.Stream(device=dev)
    copied = [[torch.cuda.Event() for _ in roles] for _ in range(slots)]      # the variable's upload of the slot has landed
    consumed = [torch.cuda.Event() for _ in range(slots)]      # the slot's raw buffers have been decoded
    used = [False] * slots
    moved, host_s = 0, 0.0
    # Nothing inside the chunk loop may make the host wait for the GPU (an upload from pageable memory does: it serialised staging
    # and copies in rounds 1-2, profiles/r03_notes.md): the d/dt coefficients of the whole time axis and the tables of every box go to
    # the device once, here; a chunk uses rows / views of them.
    tcoef_all = engine.time_coefs_device(time_s) if with_q else None
    # chunk schedule: a short first chunk fills the pipeline quickly (the first upload cannot start before its staging is done)
    first = max(1, min(chunk_steps, 1 if (t1 - t0) > chunk_steps else chunk_steps))
    bounds = [t0, min(t0 + first, t1)]
    while bounds[-1] < t1:
        bounds.append(min(bounds[-1] + chunk_steps, t1))
    n_chunks = len(bounds) - 1
    for c in range(n_chunks):
        slot = c % slots
        c0, c1 = bounds[c], bounds[c + 1]
        h0, h1 = (max(c0 - 1, 0), min(c1 + 1, nt)) if with_q else (c0, c1)
        if used[slot]:
            for ev in copied[slot]:             # the pinned buffers of this slot may be overwritten now: EVERY variable's upload has landed
                ev.synchronize()                # (they ride on different streams when deflated and plain variables mix: ADVICE r3)
            if spans is not None:
                spans.release(c - slots + 1)    # ... and the file spans only chunks up to c - slots used are no longer being read
        # only T carries the halo; the other fields start at their own first step (rows [c0 - h0, c1 - h0) of the slot)
        span_of = lambda r: (0, h1 - h0) if r == "Air Temperature" else (c0 - h0, c1 - h0)
        copier_waited = False                   # has the shared copy stream waited for this slot's last decode yet?
        for n, r in enumerate(roles):
            # variable by variable: stage (thread pool), enqueue its upload, decode it -- the copy engine starts after a fifth of the
            # chunk's staging, and the next variable is staged while this one is on the link
            a, b = span_of(r)
            t_host = time.perf_counter()
            if not direct:
                stagers[r].stage(slot, plan.tsel[h0 + a: h0 + b], a)
            host_s += time.perf_counter() - t_host
            up = stagers[r].streams[slot] if hasattr(stagers[r], "streams") else copier   # a deflated variable uploads and inflates on its own stream
            stagers[r].use = c
            with torch.cuda.stream(up):
                if used[slot] and (up is not copier or not copier_waited):
                    up.wait_event(consumed[slot])       # the raw device buffers of this slot have been decoded
                    copier_waited = copier_waited or up is copier
                if direct:
                    t_host = time.perf_counter()
                    moved += stagers[r].upload_direct(lib, spans, slot, plan.tsel[h0 + a: h0 + b], a, c, C.c_void_p(copier.cuda_stream))
                    host_s += time.perf_counter() - t_host
                else:
                    stagers[r].upload(slot, a, b)
                    if not isinstance(stagers[r], _ChunkStager):
                        moved += (b - a) * stagers[r].step_elems * stagers[r].itemsize
                copied[slot][n].record(up)
            compute.wait_event(copied[slot][n])
        # decode + row pass, sub-chunk by sub-chunk: T with the one-step halo of the SUB-chunk (rows of the raw slot count from h0)
        for s0 in (() if packed else range(c0, c1, dec_steps)):
            s1 = min(s0 + dec_steps, c1)
            g0, g1 = (max(s0 - 1, 0), min(s1 + 1, nt)) if with_q else (s0, s1)
            with torch.cuda.device(dev):
                for r in roles:
                    a, b = (g0 - h0, g1 - h0) if r == "Air Temperature" else (s0 - h0, s1 - h0)
                    unit = 1.0 if r == geo_role else ds.field_scale(variable_list_df, r)
                    _ingest_call(lib, rvars[r], stagers[r].raw_dev[slot][a].data_ptr(), b - a, (nl_in, ny_in, nx_in, nl, ny, nx), maps, unit,
                                 decode[r], common, cubes[keys[r]][a - (g0 - h0)].data_ptr(), compute)
            f = {k: t[: g1 - g0] for k, t in cubes.items()}
            if keep is not None:                # (u, v, geopotential start at their own first step: rows [s0 - g0, s1 - g0) of the cubes)
                for k in keep:
                    keep[k][s0 - t0: s1 - t0].copy_(cubes[k][s0 - g0: s1 - g0, k_keep])
            part = own_boxes.part(s0 - t0, s1 - t0) if per_step_boxes else fixed_box
            engine.rowstats(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], part,
                            tcoef=tcoef_all[g0:g1] if with_q else None, t_begin=s0 - g0, t_count=s1 - s0, with_q=with_q,
                            rows_out=rows[: s1 - s0], per_step_boxes=per_step_boxes)
            engine.level_stage(rows[: s1 - s0], part, levraw[s0 - t0: s1 - t0], phi_scale=phi_scale)
        for s0 in (range(c0, c1, dec_steps) if packed else ()):
            # the box-packed series: every output step holds that step's box alone (lec_ingest enters the maps at the box's south-west
            # corner, step by step: `step_tab`), T also from the two neighbouring steps' raw slices
            s1 = min(s0 + dec_steps, c1)
            geom = (nl_in, ny_in, nx_in, nl, nyp, nxp)
            n = s1 - s0
            with torch.cuda.device(dev):
                for r in roles:                 # ONE gather per plane and sub-chunk: every output step's box through lec_ingest's per-step origins
                    unit = 1.0 if r == geo_role else ds.field_scale(variable_list_df, r)
                    for key, shift in [(keys[r], 0)] + ([("tm", -1), ("tp", 1)] if r == "Air Temperature" else []):
                        _ingest_call(lib, rvars[r], stagers[r].raw_dev[slot][0].data_ptr(), n, geom, maps, unit, decode[r], common,
                                     cubes[key][0].data_ptr(), compute, step=step_tab[shift][s0:s1], step_base=h0,
                                     nt_src=int(stagers[r].raw_dev[slot].shape[0]))
                if keep is not None:            # the diagnostics' level of u, v, Phi over the whole crop: one gather of that level per field
                    kmap1 = maps[0][k_keep: k_keep + 1]
                    for r, k in (("Eastward Wind Component", "u"), ("Northward Wind Component", "v"), (geo_role, "geopt")):
                        unit = 1.0 if r == geo_role else ds.field_scale(variable_list_df, r)
                        _ingest_call(lib, rvars[r], stagers[r].raw_dev[slot][s0 - h0].data_ptr(), n, (nl_in, ny_in, nx_in, 1, ny, nx),
                                     (kmap1, maps[1], maps[2]), unit, decode[r], common, keep[k][s0 - t0].data_ptr(), compute)
            f = {k: t[:n] for k, t in cubes.items()}
            part = own_boxes.part(s0 - t0, s1 - t0)
            if dtdt is not None:
                engine.time_stencil(f["tm"], f["tair"], f["tp"], tcoef_all[s0:s1], out=dtdt[:n])
                tkw = dict(dTdt=dtdt[:n])
            else:
                tkw = dict(tm=f["tm"], tp=f["tp"], tcoef=tcoef_all[s0:s1])
            engine.rowstats(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], part, t_begin=0, t_count=n, with_q=True,
                            rows_out=rows[:n], per_step_boxes=True, **tkw)
            engine.level_stage(rows[:n], part, levraw[s0 - t0: s1 - t0], phi_scale=phi_scale)
        consumed[slot].record(compute)
        used[slot] = True
    t_loop = time.perf_counter()                # (every chunk enqueued)
    res = engine.vertical_stage(levraw, own_boxes if per_step_boxes else fixed_box, drop_any_time=not per_step_boxes,
                                merge_dropmask=merge_dropmask, out=out)
    on_device = [r for r in roles if isinstance(stagers[r], _ChunkStager)]
    for r in on_device:
        stagers[r].finish()                     # every chunk inflated (raises otherwise)
        moved += stagers[r].compressed_bytes
    if spans is not None:
        copier.synchronize()                    # every copy has read its span (the deflated variables' streams: finish() above)
        reg_stats = dict(registered_bytes=spans.registered_bytes, register_calls=spans.calls)
        spans.close()
    if stats is not None:
        stats.update(staging="registered" if (direct or chunk_direct) else "staged", **(reg_stats if spans is not None else {}))
        if keep is not None:
            stats["level_slices"] = keep
        stats.update(inflate="device" if on_device else ("host" if any(hasattr(v.data, "chunk_streams") for v in rvars.values()) else "none"))
        stats.update(row_record_bytes=rows.numel() * 8, levraw_bytes=levraw.numel() * 8, device_buffer_bytes=int(mem_setup))
        torch.cuda.synchronize(dev)
        stats.update(seconds=dict(setup=t_setup - t_enter, chunk_loop_host=t_loop - t_setup, drain=time.perf_counter() - t_loop,
                                  registering=(spans.seconds if spans is not None else 0.0)))
        stats.update(bytes_moved=moved, host_staging_seconds=host_s, chunks=n_chunks, chunk_steps=chunk_steps, storage=str(out_dtype).replace("torch.", ""),
                     decode={keys[r]: str(d) for r, d in decode.items()}, box=tuple(int(x) for x in boxes[0]), domain=(nt, nl, ny, nx))
    return res


def lec_fixed_streamed(raw: ds.RawDataset, plan: IngestPlan, variable_list_df, box_limits, **kw) -> LECResult:
    """The fixed framework on the streamed path: one box for the whole series (see lec_streamed)."""
    return lec_streamed(raw, plan, variable_list_df, [box_limits], per_step_boxes=False, **kw)
