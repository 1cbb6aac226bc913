"""LECEngine: device-resident Lorenz-Energy-Cycle computation through the C-ABI HIP library.

Holds the (time, level, lat, lon) fields as PyTorch-ROCm tensors, builds the small coefficient
tables on the host, and issues ``lec_rowstats`` + ``lec_reduce`` on the current HIP stream.
It is the replacement for ``BoxData`` + the four analysis classes of the reference
(box_data.py:58-105, energy_contents.py, conversion_terms.py, boundary_terms.py,
generation_and_dissipation_terms.py).  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib, tables
from .constants import LEVEL_TERMS, SCALAR_TERMS


@dataclass
class LECResult:
    """Per-time-step outputs (device tensors) of one engine call.  ``scalars`` and ``levels`` are views of ``packed``: one record of
    16 + 21 nl doubles per time step, written by ``lec_reduce`` in place -- the unit the time-sharded gather moves (parallel.py)."""
    scalars: torch.Tensor      # [t_count, 16] fp64, columns = constants.SCALAR_TERMS
    levels: torch.Tensor       # [t_count, 21, nl] fp64, tables = constants.LEVEL_TERMS
    nanflag: torch.Tensor      # [t_count] int32: NaN level values repaired/dropped by _handle_nans
    rows: Optional[torch.Tensor] = None   # [t_count, nl, nyb_max, 32] row records (kept on request)
    packed: Optional[torch.Tensor] = None  # [t_count, 16 + 21 nl] fp64

    def scalars_dict(self) -> Dict[str, np.ndarray]:
        s = self.scalars.cpu().numpy()
        return {name: s[:, i].copy() for i, name in enumerate(SCALAR_TERMS)}

    def levels_dict(self) -> Dict[str, np.ndarray]:
        lv = self.levels.cpu().numpy()
        return {name: lv[:, i, :].copy() for i, name in enumerate(LEVEL_TERMS)}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_KERNELS = {"auto": _lib.KERNEL_AUTO, "two_sweep": _lib.KERNEL_TWO_SWEEP, "row_sweep": _lib.KERNEL_ROW_SWEEP,
            "row_block": _lib.KERNEL_ROW_BLOCK, "box_tile": _lib.KERNEL_BOX_TILE, "box_plane": _lib.KERNEL_BOX_PLANE}
_ORDERS = {"auto": _lib.ORDER_AUTO, "memory": _lib.ORDER_MEMORY, "xcd_lat": _lib.ORDER_XCD_LAT, "xcd_tiled": _lib.ORDER_XCD_TILED}


def make_tuning(tuning: Optional[dict]) -> "_lib.Tuning":
    """``lec_tuning`` from a dict with any of: kernel ("auto" | "two_sweep" | "row_sweep" | "row_block" | "box_tile" | "box_plane"),
    block_shape (e.g. 212), order ("auto" | "memory" | "xcd_lat" | "xcd_tiled"), tile_t, tile_j, f32_vec.
    None / {} = the library's defaults (what is measured and shipped); the rest is for cross-checks and A/B runs."""
    t = _lib.Tuning()
    if not tuning:
        return t
    unknown = set(tuning) - {"kernel", "block_shape", "order", "tile_t", "tile_j", "f32_vec"}
    if unknown:
        raise ValueError(f"unknown tuning keys {sorted(unknown)}")
    k, o = tuning.get("kernel", "auto"), tuning.get("order", "auto")
    if k not in _KERNELS or o not in _ORDERS:
        raise ValueError(f"tuning: kernel must be one of {sorted(_KERNELS)}, order one of {sorted(_ORDERS)}")
    t.kernel, t.order = _KERNELS[k], _ORDERS[o]
    t.block_shape = int(tuning.get("block_shape", 0))
    t.tile_t, t.tile_j = int(tuning.get("tile_t", 0)), int(tuning.get("tile_j", 0))
    t.f32_vec = int(tuning.get("f32_vec", 0))
    return t


class PreparedBoxes:
    """Boxes whose host tables are built and uploaded (``LECEngine.prepare_boxes``): a series of thousands of per-time-step boxes
    is normalised, keyed and looked up once instead of at every call."""

    def __init__(self, boxes, bt, dev):
        self.boxes, self.bt, self.dev = boxes, bt, dev

    def __len__(self):
        return len(self.boxes)

    def part(self, a: int, b: int) -> "PreparedBoxes":
        """Boxes [a, b) of a series of per-time-step boxes (a chunk of the series): views of the uploaded tables, no host work and no
        copy -- every table is indexed by box along its first axis."""
        import dataclasses
        if not (0 <= a < b <= len(self.boxes)):
            raise ValueError("part: need 0 <= a < b <= number of boxes")
        bt = dataclasses.replace(self.bt, box=self.bt.box[a:b], boxtab=self.bt.boxtab[a:b], wlon=self.bt.wlon[a:b], glon=self.bt.glon[a:b],
                                 lattab=self.bt.lattab[a:b], boxtab2=self.bt.boxtab2[a:b], lattab2=self.bt.lattab2[a:b])
        return PreparedBoxes(self.boxes[a:b], bt, {k: v[a:b] for k, v in self.dev.items()})


class LECEngine:
    """One engine per (grid, device).

    Parameters
    ----------
    lat_deg, lon_deg : 1-D arrays, degrees, ascending (lat S->N, lon W->E in (-180, 180])
    level_pa         : 1-D array, Pa, ascending
    device           : torch device of the field tensors (``cuda:N`` on ROCm)
    """

    MAX_STEPS_PER_LAUNCH = 32768       # time steps per lec_rowstats launch (longer calls are cut into parts by `rowstats`)

    def __init__(self, lat_deg, lon_deg, level_pa, device="cuda"):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LecLibraryError("LECEngine needs a GPU device: the HIP path is the only path")
        self.lat = np.asarray(lat_deg, dtype=np.float64)
        self.lon = np.asarray(lon_deg, dtype=np.float64)
        self.level = np.asarray(level_pa, dtype=np.float64)
        if np.any(np.diff(self.lat) <= 0) or np.any(np.diff(self.lon) <= 0):
            raise ValueError("lat and lon must be strictly ascending (preprocessing.py:358-362 sorts them)")
        levtab, levtab2 = tables.level_tables(self.level)
        self._levtab = self._up(levtab)
        self._levtab2 = self._up(levtab2)
        self._box_cache = {}
        self._work = {}        # stage-2 workspaces by shape: am, levraw, dropmask (true scratch: never handed out)
        self._tcoef = None     # (time axis, its d/dt coefficients on the device) of the last call

    # -- helpers ---------------------------------------------------------------------------
    def _up(self, a: np.ndarray, dtype=torch.float64) -> torch.Tensor:
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def box_from_limits(self, west, east, south, north):
        """Nearest-grid-point inclusive box, as BoxData._set_domain_limits (box_data.py:115-131)."""
        return tables.box_indices(self.lat, self.lon, west, east, south, north)

    def prepare_boxes(self, boxes, nyb_min: int = 0, packed: bool = False) -> PreparedBoxes:
        """Index quadruples (iw, ie, js, jn) -> PreparedBoxes for ``rowstats`` / ``reduce`` / ``compute``.  ``nyb_min``: the row
        count of the record buffer the boxes will be used with (chunks of a series share the tallest box's).  ``packed``: the boxes
        of a BOX-PACKED series (``pack_boxes``; include/lec_hip.h): stage 1 then addresses every step's box at the origin of its
        slab, while every table is built from the true grid boxes as always."""
        boxes = [tuple(int(x) for x in b) for b in boxes]
        bt, dev = self._box_tables(boxes, nyb_min)
        if packed:
            dev = dict(dev, box_data=self._up(np.array([(0, b[1] - b[0], 0, b[3] - b[2]) for b in boxes], dtype=np.int32), torch.int32))
        return PreparedBoxes(boxes, bt, dev)

    def pack_boxes(self, cube: torch.Tensor, boxes, shift: int = 0, ny: Optional[int] = None, nx: Optional[int] = None) -> torch.Tensor:
        """[nt, nl, ny_grid, nx_grid] -> the box-packed layout [len(boxes), nl, ny, nx] (default: the tallest / widest box): step t
        holds box t of ``cube[t + shift]`` (clamped to the cube: the step itself where a time neighbour does not exist) at its
        origin.  ONE launch of ``lec_ingest`` -- the gather the streamed moving framework packs its series with (``ingest.lec_streamed``:
        there the source is the file's raw bytes, here a cube in HBM; ``lec_ingest_args.step_d`` = per step {source step, where the box
        starts}) -- for tests, ``bench.py --moving`` and cubes a caller already holds.  What a slab holds beside a box lower / narrower
        than the slab is whatever the lengthened index maps point at (the grid's last row / column): stage 1 never reads it."""
        b = np.array([tuple(int(x) for x in q) for q in (boxes.boxes if isinstance(boxes, PreparedBoxes) else boxes)], dtype=np.int64)
        nt, nl, ny_in, nx_in = (int(x) for x in cube.shape)
        if len(b) != nt:
            raise ValueError("pack_boxes: one box per time step of the cube")
        if cube.dtype not in (torch.float64, torch.float32) or not cube.is_contiguous() or cube.device.type != "cuda":
            raise ValueError("pack_boxes: a contiguous float64 / float32 cube on the GPU")
        nxb, nyb = b[:, 1] - b[:, 0] + 1, b[:, 3] - b[:, 2] + 1
        ny, nx = int(ny or nyb.max()), int(nx or nxb.max())
        if ny < nyb.max() or nx < nxb.max() or ny > ny_in or nx > nx_in:
            raise ValueError("pack_boxes: slabs must hold the tallest / widest box and fit the grid")
        dev = cube.device
        key = (nl, ny_in, nx_in, ny, nx, str(dev))
        maps = self._pack_maps.get(key) if hasattr(self, "_pack_maps") else None
        if maps is None:            # identity maps, lengthened by a slab (their last entry repeated): a box at the grid's edge is gathered with the slab's extents
            up = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).to(dev)
            maps = (up(np.arange(nl)), up(np.minimum(np.arange(ny_in + ny), ny_in - 1)), up(np.minimum(np.arange(nx_in + nx), nx_in - 1)))
            self._pack_maps = {key: maps}
        step = torch.as_tensor(np.column_stack([np.clip(np.arange(nt) + shift, 0, nt - 1), b[:, 2], b[:, 0]]).astype(np.int32)).to(dev)
        out = torch.empty((nt, nl, ny, nx), dtype=cube.dtype, device=dev)
        code = _lib.LEC_F64 if cube.dtype == torch.float64 else _lib.LEC_F32
        for a0 in range(0, nt, 8192):            # (a call addresses its output rows with a 31-bit grid index)
            a1 = min(nt, a0 + 8192)
            ga = _lib.IngestArgs(
                src_d=_ptr(cube), src_dtype=code, swap_bytes=0, nt=a1 - a0, nl_in=nl, ny_in=ny_in, nx_in=nx_in, nl=nl, ny=ny, nx=nx,
                kmap_d=_ptr(maps[0]), jmap_d=_ptr(maps[1]), imap_d=_ptr(maps[2]), has_packing=0, has_fill=0, scale_factor=1.0, add_offset=0.0,
                fill_value=0.0, unit_scale=1.0, out_dtype=code, decode_dtype=code, out_d=_ptr(out[a0:a1]),
                stream=C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), step_d=_ptr(step[a0:a1]), step_base=0, nt_src=nt,
                jmap_len=int(maps[1].numel()), imap_len=int(maps[2].numel()))
            with torch.cuda.device(dev):
                _lib.check(self.lib.lec_ingest(C.byref(ga)), "lec_ingest")
        return out

    def pack_boxes_by_indexing(self, cube: torch.Tensor, boxes, shift: int = 0, ny: Optional[int] = None, nx: Optional[int] = None) -> torch.Tensor:
        """``pack_boxes`` as one torch advanced-indexing gather, zeros beside the boxes: the independent form the tests hold the
        ``lec_ingest`` gather to (inside the boxes: the same values)."""
        b = np.array([tuple(int(x) for x in q) for q in (boxes.boxes if isinstance(boxes, PreparedBoxes) else boxes)], dtype=np.int64)
        nt = cube.shape[0]
        if len(b) != nt:
            raise ValueError("pack_boxes: one box per time step of the cube")
        nxb, nyb = b[:, 1] - b[:, 0] + 1, b[:, 3] - b[:, 2] + 1
        ny, nx = int(ny or nyb.max()), int(nx or nxb.max())
        dev = cube.device
        tt = torch.as_tensor(np.clip(np.arange(nt) + shift, 0, nt - 1), device=dev)[:, None, None, None]
        kk = torch.arange(cube.shape[1], device=dev)[None, :, None, None]
        jj = torch.arange(ny, device=dev)[None, None, :, None]
        ii = torch.arange(nx, device=dev)[None, None, None, :]
        up = lambda a: torch.as_tensor(a, device=dev)[:, None, None, None]
        inside = (jj < up(nyb)) & (ii < up(nxb))
        rows = torch.minimum(up(b[:, 2]) + jj, up(b[:, 3]))
        cols = torch.minimum(up(b[:, 0]) + ii, up(b[:, 1]))
        return torch.where(inside, cube[tt, kk, rows, cols], torch.zeros((), dtype=cube.dtype, device=dev)).contiguous()

    def time_stencil(self, tm: torch.Tensor, t: torch.Tensor, tp: torch.Tensor, tcoef: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dT/dt of a box-packed series as an fp64 cube (``lec_dtdt``): tcoef[s] . (tm[s], t[s], tp[s]) with the bits of stage 1's own
        per-point evaluation.  ``tcoef``: fp64 [steps, 3] on the device, the rows of the steps the cubes hold."""
        if tm.shape != t.shape or tp.shape != t.shape or tm.dtype != t.dtype or tp.dtype != t.dtype or not (tm.is_contiguous() and t.is_contiguous() and tp.is_contiguous()):
            raise ValueError("time_stencil: three contiguous cubes of one shape and dtype")
        n = int(t.shape[0])
        if tcoef.shape != (n, 3) or tcoef.dtype != torch.float64 or not tcoef.is_contiguous() or tcoef.device != t.device:
            raise ValueError("time_stencil: tcoef must be a contiguous fp64 [steps, 3] tensor on the cubes' device")
        if out is None:
            out = torch.empty(t.shape, dtype=torch.float64, device=t.device)
        elif out.shape != t.shape or out.dtype != torch.float64 or not out.is_contiguous() or out.device != t.device:
            raise ValueError("time_stencil: out must be a contiguous fp64 tensor of the cubes' shape")
        for a in range(0, n, 65535):
            b = min(n, a + 65535)
            args = _lib.DtdtArgs(tm_d=_ptr(tm[a:b]), t_d=_ptr(t[a:b]), tp_d=_ptr(tp[a:b]), dtype=_lib.LEC_F64 if t.dtype == torch.float64 else _lib.LEC_F32,
                                 n_steps=b - a, step_elems=int(t[0].numel()), tcoef_d=_ptr(tcoef[a:b]), out_d=_ptr(out[a:b]),
                                 stream=C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream))
            with torch.cuda.device(t.device):
                _lib.check(self.lib.lec_dtdt(C.byref(args)), "lec_dtdt")
        return out

    def pack_series(self, tair, u, v, omega, geopt, boxes, tcoef: torch.Tensor, ny: Optional[int] = None, nx: Optional[int] = None,
                    timing: Optional[dict] = None) -> dict:
        """The box-packed form of a moving series held as whole cubes (tests, ``bench.py --moving``, host-prepared tracks): the keyword
        arguments of ``rowstats`` -- the five fields packed per step (``pack_boxes``) and dT/dt as the series' own data: an fp64 cube
        for fp64 storage (``time_stencil``: one operand fewer per point), T of the two time neighbours on the step's box for fp32
        storage (the same bytes as a cube would be, and no rounding of dT/dt to the storage dtype).  ``tcoef``: rows of the cubes' steps.
        ``timing``: a dict that receives HIP events recorded on the current stream -- "pack" = (start, end) around the gathers (the
        per-step slice, box_data.py:297-310), "dtdt" = (start, end) around ``lec_dtdt`` (fp64 storage only): what the PRODUCER of a
        packed series spends, which ``bench.py --moving`` reports beside the consumer's rate."""
        def ev():
            if timing is None:
                return None
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e

        with torch.cuda.device(tair.device):
            e0 = ev()
            pk = {k: self.pack_boxes(c, boxes, ny=ny, nx=nx) for k, c in (("tair", tair), ("u", u), ("v", v), ("omega", omega), ("geopt", geopt)) if c is not None}
            tm, tp = self.pack_boxes(tair, boxes, shift=-1, ny=ny, nx=nx), self.pack_boxes(tair, boxes, shift=+1, ny=ny, nx=nx)
            e1 = ev()
            if tair.dtype == torch.float64:
                pk["dTdt"] = self.time_stencil(tm, pk["tair"], tp, tcoef)
            else:
                pk["tm"], pk["tp"] = tm, tp
            e2 = ev()
        if timing is not None:
            timing["pack"] = (e0, e1)
            if "dTdt" in pk:
                timing["dtdt"] = (e1, e2)
        pk.setdefault("geopt", None)
        return pk

    def _resolve_boxes(self, boxes, nyb_min: int = 0):
        if isinstance(boxes, PreparedBoxes):
            if boxes.bt.nyb_max < nyb_min:
                raise ValueError("PreparedBoxes were built for a lower record buffer: pass nyb_min to prepare_boxes")
            return boxes.boxes, boxes.bt, boxes.dev
        boxes = [tuple(int(x) for x in b) for b in (boxes if isinstance(boxes[0], (tuple, list, np.ndarray)) else [boxes])]
        bt, dev = self._box_tables(boxes, nyb_min)
        return boxes, bt, dev

    def _box_tables(self, boxes, nyb_min: int = 0):
        key = (int(nyb_min),) + tuple(int(v) for b in boxes for v in b)
        hit = self._box_cache.get(key)
        if hit is not None:
            return hit
        bt = tables.build_box_tables(self.lat, self.lon, boxes, nyb_min=nyb_min)
        dev = {
            "box": self._up(bt.box, torch.int32), "boxtab": self._up(bt.boxtab), "wlon": self._up(bt.wlon),
            "glon": self._up(bt.glon), "lattab": self._up(bt.lattab), "boxtab2": self._up(bt.boxtab2),
            "lattab2": self._up(bt.lattab2),
        }
        if len(self._box_cache) > 8:
            self._box_cache.clear()
        self._box_cache[key] = (bt, dev)
        return bt, dev

    def time_coefs_device(self, time_s) -> torch.Tensor:
        """np.gradient's coefficients over a whole series' time axis, on the device ([n, 3]).  Rows [h0, h1) serve any cube that holds
        the steps [h0, h1) of the series and processes only steps whose neighbours it holds (own steps + one-step halo): there the
        coefficients of the sub-axis equal the series' (they differ only at the halo steps, which are not processed)."""
        return self._up(tables.time_coefs(np.asarray(time_s, dtype=np.float64)))

    # -- the hot path ----------------------------------------------------------------------
    def compute(self, tair: torch.Tensor, u: torch.Tensor, v: torch.Tensor, omega: torch.Tensor,
                geopt: Optional[torch.Tensor], boxes: Sequence[Sequence[int]], *,
                time_s=None, dTdt: Optional[torch.Tensor] = None, t_begin: int = 0,
                t_count: Optional[int] = None, with_q: bool = True, phi_scale: float = 1.0,
                keep_rows: bool = False, timing: Optional[list] = None,
                drop_any_time: Optional[bool] = None,
                merge_dropmask: Optional[Callable[[torch.Tensor], None]] = None,
                tuning: Optional[dict] = None, per_step_boxes: Optional[bool] = None,
                out: Optional[torch.Tensor] = None, tm: Optional[torch.Tensor] = None, tp: Optional[torch.Tensor] = None,
                tcoef: Optional[torch.Tensor] = None) -> LECResult:
        """All LEC terms for time steps [t_begin, t_begin + t_count) of the cubes: ``rowstats`` then ``reduce``.
        (``tm`` / ``tp`` / ``tcoef``: a box-packed series, see ``rowstats``.)

        ``boxes``: one (iw, ie, js, jn) quadruple (fixed framework) or one per processed time step
        (moving framework).  ``per_step_boxes``: True = the moving framework's semantics (default when more than one
        box is given); say so explicitly for a ONE-step shard or chunk of a moving series, which would otherwise read
        as a fixed box (same numbers to rounding, but another kernel family: not the same bits).  ``time_s`` (seconds, length nt) gives dT/dt by np.gradient over the
        cube's time axis unless a ``dTdt`` cube is supplied (moving framework).
        ``drop_any_time``: _handle_nans' dropna semantics -- True drops a level that stays NaN at any processed
        time step from every time step's integrals (what the fixed framework's [time, level] arrays do); default:
        True for one fixed box, False for per-time-step boxes (the moving framework builds one BoxData per step).
        ``merge_dropmask``: see ``reduce`` (time-sharded runs).
        ``timing``: a list that receives one (start, end) pair of HIP events recorded on the launch
        stream around the stage-1 kernel (bench.py's roofline figure).
        """
        rows = self.rowstats(tair, u, v, omega, geopt, boxes, time_s=time_s, dTdt=dTdt, t_begin=t_begin, t_count=t_count,
                             with_q=with_q, timing=timing, tuning=tuning, per_step_boxes=per_step_boxes, tm=tm, tp=tp, tcoef=tcoef)
        if drop_any_time is None and per_step_boxes:
            drop_any_time = False
        return self.reduce(rows, boxes, phi_scale=phi_scale, drop_any_time=drop_any_time, merge_dropmask=merge_dropmask,
                           keep_rows=keep_rows, out=out)

    def rowstats(self, tair: torch.Tensor, u: torch.Tensor, v: torch.Tensor, omega: torch.Tensor,
                 geopt: Optional[torch.Tensor], boxes: Sequence[Sequence[int]], *,
                 time_s=None, dTdt: Optional[torch.Tensor] = None, t_begin: int = 0,
                 t_count: Optional[int] = None, with_q: bool = True, timing: Optional[list] = None,
                 rows_out: Optional[torch.Tensor] = None, tuning: Optional[dict] = None,
                 per_step_boxes: Optional[bool] = None, tcoef: Optional[torch.Tensor] = None,
                 tm: Optional[torch.Tensor] = None, tp: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Stage 1 (``lec_rowstats``): row records [t_count, nl, nyb_max, 32] of time steps [t_begin, t_begin + t_count).
        ``tm`` / ``tp``: a BOX-PACKED series (``pack_boxes``; ``boxes`` prepared with ``packed=True``): the cubes hold every step's
        box at the origin of its slab, ``tm`` / ``tp`` T of the previous / next step on that box.  Same records, bit for bit.
        ``tuning``: see ``make_tuning`` (kernel family / order / tile shape; default = the library's choice).
        ``tcoef``: the d/dt coefficients of the cube's time steps already on the device (fp64 [nt, 3], e.g. rows [h0, h1) of
        ``time_coefs_device`` of the whole series) instead of ``time_s`` -- a chunk loop then uploads nothing per call (an upload from
        pageable memory makes the host wait for the stream, which serialises a copy / compute pipeline)."""
        if tair.dim() != 4:
            raise ValueError("fields must be [time, level, lat, lon]")
        nt, nl, ny, nx = tair.shape
        packed = tm is not None or tp is not None or (isinstance(boxes, PreparedBoxes) and "box_data" in boxes.dev)
        if packed:
            if not isinstance(boxes, PreparedBoxes) or "box_data" not in boxes.dev or (with_q and (dTdt is None) == (tm is None or tp is None)):
                raise ValueError("a box-packed series: boxes from prepare_boxes(..., packed=True) and, with_q, either tm and tp or a dTdt cube")
            if nl != self.level.size or ny < boxes.bt.nyb_max or nx < boxes.bt.nxb_max or ny > self.lat.size or nx > self.lon.size:
                raise ValueError(f"packed cubes {tuple(tair.shape)}: need {self.level.size} levels and slabs that hold the tallest / widest box")
        elif (nl, ny, nx) != (self.level.size, self.lat.size, self.lon.size):
            raise ValueError(f"field shape {tuple(tair.shape)} does not match the engine grid "
                             f"({self.level.size} levels, {self.lat.size} lats, {self.lon.size} lons)")
        if tair.dtype not in (torch.float64, torch.float32):
            raise ValueError("fields must be float64 or float32")
        cubes = [tair, u, v, omega] + ([geopt] if geopt is not None else []) + ([dTdt] if dTdt is not None else []) + ([tm, tp] if tm is not None else [])
        for c in cubes:
            if c.shape != tair.shape or c.dtype != tair.dtype or c.device != tair.device or not c.is_contiguous():
                raise ValueError("all field cubes must share shape, dtype, device and be contiguous")
        if tair.device.type != "cuda":
            raise _lib.LecLibraryError("fields must live on the GPU: there is no CPU path")
        if t_count is None:
            t_count = nt - t_begin
        # rows_out: a slice of a longer series' record buffer (chunked processing): its row count is the tallest box of the whole series
        boxes, bt, dev = self._resolve_boxes(boxes, nyb_min=0 if rows_out is None else int(rows_out.shape[2]))
        if per_step_boxes is None:
            per_step_boxes = len(boxes) != 1
        if len(boxes) != (t_count if per_step_boxes else 1):
            raise ValueError("boxes: give one box, or one per processed time step")

        if tcoef is not None:
            if tcoef.shape != (nt, 3) or tcoef.dtype != torch.float64 or tcoef.device != tair.device or not tcoef.is_contiguous():
                raise ValueError("tcoef must be a contiguous fp64 [nt, 3] tensor on the fields' device")
            if nt < 2 and tm is None:           # (a box-packed series brings its time neighbours along: one step is a series)
                raise ValueError("dT/dt by finite differences needs at least 2 time steps")
        elif with_q and dTdt is None:
            if time_s is None:
                raise ValueError("with_q needs time_s (seconds) or a dTdt cube")
            time_s = np.asarray(time_s, dtype=np.float64)
            if time_s.size != nt:
                raise ValueError("time_s must have one entry per time step of the cube")
            if nt < 2:
                raise ValueError("dT/dt by finite differences needs at least 2 time steps")
            # cached: an upload from pageable memory makes the host wait for the stream at every call
            if self._tcoef is None or self._tcoef[0].shape != time_s.shape or not np.array_equal(self._tcoef[0], time_s):
                self._tcoef = (time_s.copy(), self._up(tables.time_coefs(time_s)))
            tcoef = self._tcoef[1]

        f64 = dict(dtype=torch.float64, device=tair.device)
        if t_count > self.MAX_STEPS_PER_LAUNCH:
            # a stage-1 launch addresses its time steps with 16-bit grid coordinates in some kernel families: longer series go out in
            # parts (same kernels, same bits: results do not depend on how a series is cut)
            rows = rows_out if rows_out is not None else torch.empty((t_count, nl, bt.nyb_max, _lib.LEC_NSTAT), **f64)
            whole = PreparedBoxes(boxes, bt, dev)
            for a in range(0, t_count, self.MAX_STEPS_PER_LAUNCH):
                b = min(a + self.MAX_STEPS_PER_LAUNCH, t_count)
                self.rowstats(tair, u, v, omega, geopt, whole.part(a, b) if per_step_boxes else whole, dTdt=dTdt, t_begin=t_begin + a,
                              t_count=b - a, with_q=with_q, timing=timing, rows_out=rows[a:b], tuning=tuning, per_step_boxes=per_step_boxes,
                              tcoef=tcoef, tm=tm, tp=tp)
            return rows
        if rows_out is None:
            rows = torch.empty((t_count, nl, bt.nyb_max, _lib.LEC_NSTAT), **f64)
        else:       # a slice of a longer series' record buffer (chunked ingest): stage 2 runs once over all of it
            rows = rows_out
            if rows.shape != (t_count, nl, bt.nyb_max, _lib.LEC_NSTAT) or rows.dtype != torch.float64 or not rows.is_contiguous():
                raise ValueError("rows_out must be a contiguous fp64 [t_count, nl, nyb_max, 32] tensor")
        stream = C.c_void_p(torch.cuda.current_stream(tair.device).cuda_stream)
        ra = _lib.RowstatsArgs(
            tair_d=_ptr(tair), u_d=_ptr(u), v_d=_ptr(v), omega_d=_ptr(omega), geopt_d=_ptr(geopt), dTdt_d=_ptr(dTdt),
            dtype=_lib.LEC_F64 if tair.dtype == torch.float64 else _lib.LEC_F32, with_q=int(bool(with_q)),
            nt=nt, nl=nl, ny=ny, nx=nx, t_begin=t_begin, t_count=t_count,
            n_box=len(boxes), nxb_max=bt.nxb_max, nyb_max=bt.nyb_max, lon_uniform=int(bt.lon_uniform),
            box_per_step=int(bool(per_step_boxes)), reserved0=0, box_d=_ptr(dev["box_data"] if packed else dev["box"]), boxtab_d=_ptr(dev["boxtab"]), wlon_d=_ptr(dev["wlon"]), glon_d=_ptr(dev["glon"]),
            lattab_d=_ptr(dev["lattab"]), levtab_d=_ptr(self._levtab), tcoef_d=_ptr(tcoef),
            rows_d=_ptr(rows), stream=stream, tuning=make_tuning(tuning), tm_d=_ptr(tm), tp_d=_ptr(tp))
        with torch.cuda.device(tair.device):
            if timing is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            _lib.check(self.lib.lec_rowstats(C.byref(ra)), "lec_rowstats")
            if timing is not None:
                ev1.record()
                timing.append((ev0, ev1))
        return rows

    @staticmethod
    def packed_width(nl: int) -> int:
        """Doubles per time step of a packed result record: 16 scalars + 21 level tables."""
        return _lib.LEC_NSCALAR + _lib.LEC_NLEVTAB * int(nl)

    def reduce(self, rows: torch.Tensor, boxes: Sequence[Sequence[int]], *, phi_scale: float = 1.0,
               drop_any_time: Optional[bool] = None, merge_dropmask: Optional[Callable[[torch.Tensor], None]] = None,
               keep_rows: bool = False, out: Optional[torch.Tensor] = None,
               nanflag_out: Optional[torch.Tensor] = None) -> LECResult:
        """Stage 2 on row records [t_count, nl, nyb_max, 32] (``lec_reduce``).

        ``out``: a caller-owned fp64 [t_count, 16 + 21 nl] tensor (rows may be strided, columns contiguous) that receives the packed
        records -- e.g. the send buffer of the time-sharded gather (``parallel.SeriesGatherer``), so a pass allocates nothing and
        repacks nothing; ``nanflag_out``: int32 [t_count].  Default: fresh tensors.

        ``merge_dropmask``: for a series processed in shards -- called with this shard's any-time NaN-level mask
        (int32 [28, nl], non-zero = drop) and must merge it in place with the other shards' masks (element-wise
        max, e.g. an all_reduce); the merged mask then applies to every shard, as xarray's dropna(dim=level) on the
        whole [time, level] array does in the reference (energy_contents.py:203-207)."""
        t_count, nl = int(rows.shape[0]), int(rows.shape[1])
        boxes, bt, dev = self._resolve_boxes(boxes, nyb_min=int(rows.shape[2]))
        if len(boxes) not in (1, t_count):
            raise ValueError("boxes: give one box, or one per processed time step")
        if rows.shape != (t_count, self.level.size, bt.nyb_max, _lib.LEC_NSTAT) or rows.dtype != torch.float64 or not rows.is_contiguous():
            raise ValueError("rows must be a contiguous fp64 [t_count, nl, nyb_max, 32] tensor")
        am, levraw, dropmask_ws = self._workspace(t_count, nl, rows.device)
        res = self._stage2(_lib.STAGE_BOTH, rows, levraw, am, dropmask_ws, boxes, bt, dev, phi_scale, drop_any_time, merge_dropmask, out, nanflag_out)
        res.rows = rows if keep_rows else None
        return res

    # -- the two halves of stage 2, for series whose row records are not held whole (ingest.lec_streamed) ---------------------------
    def level_stage(self, rows: torch.Tensor, boxes, levraw_out: torch.Tensor, *, phi_scale: float = 1.0) -> None:
        """rows [t_count, nl, nyb_max, 32] -> ``levraw_out`` [t_count, nl, 40] (``lec_reduce`` with stage LEC_STAGE_LEVELS): the level x
        latitude half of stage 2 for a chunk of a series.  6.8 MB of row records per 37 x 721 time step become 12 KB, so a streamed
        series keeps ONE levraw buffer for all its steps and recycles the chunk's row records."""
        t_count, nl = int(rows.shape[0]), int(rows.shape[1])
        boxes, bt, dev = self._resolve_boxes(boxes, nyb_min=int(rows.shape[2]))
        if len(boxes) not in (1, t_count):
            raise ValueError("boxes: give one box, or one per processed time step")
        if rows.shape != (t_count, self.level.size, bt.nyb_max, _lib.LEC_NSTAT) or rows.dtype != torch.float64 or not rows.is_contiguous():
            raise ValueError("rows must be a contiguous fp64 [t_count, nl, nyb_max, 32] tensor")
        if (levraw_out.shape != (t_count, nl, _lib.LEC_NLEVRAW) or levraw_out.dtype != torch.float64 or levraw_out.device != rows.device
                or not levraw_out.is_contiguous()):
            raise ValueError("levraw_out must be a contiguous fp64 [t_count, nl, 40] tensor on the rows' device")
        am = self._workspace(t_count, nl, rows.device, am_only=True)
        self._stage2(_lib.STAGE_LEVELS, rows, levraw_out, am, None, boxes, bt, dev, phi_scale, False, None, None, None)

    def vertical_stage(self, levraw: torch.Tensor, boxes, *, drop_any_time: Optional[bool] = None,
                       merge_dropmask: Optional[Callable[[torch.Tensor], None]] = None, out: Optional[torch.Tensor] = None,
                       nanflag_out: Optional[torch.Tensor] = None) -> LECResult:
        """levraw [t_count, nl, 40] of a whole series (or shard) -> the packed records (``lec_reduce`` with stage LEC_STAGE_VERTICAL):
        _handle_nans with the any-time mask over ALL the steps (``drop_any_time`` / ``merge_dropmask`` as in ``reduce``), the pressure
        integrals and the boundary assembly.  ``boxes``: the series' box(es), for the per-box constants."""
        t_count, nl = int(levraw.shape[0]), int(levraw.shape[1])
        boxes, bt, dev = self._resolve_boxes(boxes)
        if len(boxes) not in (1, t_count):
            raise ValueError("boxes: give one box, or one per processed time step")
        if levraw.shape != (t_count, self.level.size, _lib.LEC_NLEVRAW) or levraw.dtype != torch.float64 or not levraw.is_contiguous():
            raise ValueError("levraw must be a contiguous fp64 [t_count, nl, 40] tensor")
        dropmask_ws = torch.empty((_lib.LEC_NLEVFUN, nl), dtype=torch.int32, device=levraw.device)
        return self._stage2(_lib.STAGE_VERTICAL, None, levraw, None, dropmask_ws, boxes, bt, dev, 1.0, drop_any_time, merge_dropmask, out, nanflag_out)

    def _workspace(self, t_count: int, nl: int, device, am_only: bool = False):
        # the workspaces are scratch of one call only (stream-ordered: the next call on the stream may reuse them); keyed by the
        # stream too: two calls of one shape on different streams must not share scratch (or the dropmask)
        f64 = dict(dtype=torch.float64, device=device)
        wkey = (t_count, nl, str(device), int(torch.cuda.current_stream(device).cuda_stream))
        if wkey not in self._work:
            if len(self._work) > 4:
                self._work.clear()
            self._work[wkey] = (torch.empty((t_count, nl, 8), **f64), torch.empty((t_count, nl, _lib.LEC_NLEVRAW), **f64),
                                torch.empty((_lib.LEC_NLEVFUN, nl), dtype=torch.int32, device=device))
        return self._work[wkey][0] if am_only else self._work[wkey]

    def _stage2(self, stage, rows, levraw, am, dropmask_ws, boxes, bt, dev, phi_scale, drop_any_time, merge_dropmask, out, nanflag_out) -> Optional[LECResult]:
        t_count, nl = int(levraw.shape[0]), int(levraw.shape[1])
        device = levraw.device
        f64 = dict(dtype=torch.float64, device=device)
        width = self.packed_width(nl)
        scalars = levels = nanflag = None
        if stage != _lib.STAGE_LEVELS:
            if out is None:
                out = torch.empty((t_count, width), **f64)
            elif (out.shape != (t_count, width) or out.dtype != torch.float64 or out.device != device or out.stride(1) != 1
                  or out.stride(0) < width or out.data_ptr() % 8):
                raise ValueError(f"out must be an fp64 [t_count, {width}] tensor on the rows' device with contiguous columns")
            scalars = out[:, :_lib.LEC_NSCALAR]
            levels = out[:, _lib.LEC_NSCALAR:].unflatten(1, (_lib.LEC_NLEVTAB, nl))
            if nanflag_out is None:
                nanflag = torch.empty((t_count,), dtype=torch.int32, device=device)
            else:
                nanflag = nanflag_out
                if nanflag.shape != (t_count,) or nanflag.dtype != torch.int32 or nanflag.device != device or not nanflag.is_contiguous():
                    raise ValueError("nanflag_out must be a contiguous int32 [t_count] tensor on the rows' device")
        if drop_any_time is None:
            drop_any_time = len(boxes) == 1
        dropmask = dropmask_ws if (drop_any_time and stage != _lib.STAGE_LEVELS) else None
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        mode = 0 if dropmask is None else (2 if merge_dropmask is not None else 1)
        rd = _lib.ReduceArgs(
            rows_d=_ptr(rows), t_count=t_count, nl=nl, n_box=len(boxes), nyb_max=bt.nyb_max,
            box_d=_ptr(dev["box"]), boxtab2_d=_ptr(dev["boxtab2"]), lattab2_d=_ptr(dev["lattab2"]),
            levtab2_d=_ptr(self._levtab2), phi_scale=float(phi_scale),
            drop_any_time=mode, stage=stage, dropmask_d=_ptr(dropmask),
            am_d=_ptr(am), levraw_d=_ptr(levraw), scalars_d=_ptr(scalars), levels_d=_ptr(levels),
            nanflag_d=_ptr(nanflag), stream=stream, scalars_stride=0 if out is None else int(out.stride(0)),
            levels_stride=0 if out is None else int(out.stride(0)))
        with torch.cuda.device(device):
            if mode == 2:
                _lib.check(self.lib.lec_dropmask(C.byref(rd)), "lec_dropmask")
                merge_dropmask(dropmask)
                if stage == _lib.STAGE_BOTH:
                    rd.stage = _lib.STAGE_VERTICAL          # lec_dropmask has filled levraw: the level half need not run again
            _lib.check(self.lib.lec_reduce(C.byref(rd)), "lec_reduce")
        if stage == _lib.STAGE_LEVELS:
            return None
        return LECResult(scalars=scalars, levels=levels, nanflag=nanflag, rows=None, packed=out)
