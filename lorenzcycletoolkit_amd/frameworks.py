"""Drop-in counterparts of the reference's L4/L3 call surface, backed by the HIP engine.

* ``lec_fixed`` / ``lec_moving``: same arguments, results tree and CSV schema as
  src/frameworks/lec_fixed_framework.py:30-303 and lec_moving_framework.py:546-745.
* ``BoxData`` + ``EnergyContents`` / ``ConversionTerms`` / ``BoundaryTerms`` /
  ``GenerationDissipationTerms`` with the reference's ``calc_*`` methods (box_data.py:78-90,
  lec_fixed_framework.py:216-271): BoxData runs the two HIP stages once for the whole cube; the
  ``calc_*`` methods hand out the finished series and append the per-level CSV rows.

Where the reference loops over time steps in Python (lec_moving_framework.py:639) this module
issues ONE engine call with one box per time step.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Optional

import numpy as np
import pandas as pd
import torch

from . import dataset as ds
from . import phases
from .constants import LEVEL_TERMS
from .engine import LECEngine, LECResult
from .tables import budgets_and_residuals

FIXED_COLUMNS = ["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "Gz", "Ge"]
MOVING_COLUMNS = ["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "BΦZ", "BΦE", "Gz", "Ge"]


def _device(args=None):
    shard = getattr(args, "shard", None)
    dev = shard.device if shard is not None else (getattr(args, "device", None) or os.environ.get("LEC_DEVICE", "cuda:0"))
    if not torch.cuda.is_available():
        raise RuntimeError("lorenzcycletoolkit_amd needs an AMD GPU (PyTorch-ROCm): there is no CPU path")
    return torch.device(dev)


class BoxData:
    """Counterpart of box_data.py:58-105.  ``data`` is an LECDataset (time, level, lat, lon).

    Fixed framework: one box for all time steps, dT/dt by finite differences over the cube's time axis.
    Moving framework: pass ``boxes_limits`` = one (west, east, south, north) per time step (the four
    scalar limits are then ignored); ``dTdt`` may be None (differentiated on the device, as
    lorenzcycletoolkit.py:184-186 does over the same time axis) or a host cube.

    Time-sharded runs (``args.shard``, a parallel.ShardContext: one process per GPU): every rank computes its contiguous block of
    time steps from the steps it holds (own + one-step T halo; ``data.t_held``), the any-time NaN-level mask is merged across the
    ranks, and ONE gather brings the packed per-step records to rank 0, whose BoxData then looks exactly like the one-process one;
    on the other ranks ``result`` / ``scalars`` / ``levels`` are None.
    """

    def __init__(self, data: ds.LECDataset, variable_list_df: pd.DataFrame, western_limit=None, eastern_limit=None,
                 southern_limit=None, northern_limit=None, args=None, results_subdirectory: str = ".",
                 results_subdirectory_vertical_levels: str = ".", dTdt: Optional[np.ndarray] = None,
                 boxes_limits=None):
        if args is not None and not getattr(args, "residuals", True):
            # the reference looks up "Friction Velocity" here and fails (box_data.py:190-195; SURVEY B-8)
            raise KeyError("Friction Velocity")
        self.args = args
        self.results_subdirectory = results_subdirectory
        self.results_subdirectory_vertical_levels = results_subdirectory_vertical_levels
        self.LonIndexer = variable_list_df.loc["Longitude"]["Variable"]
        self.LatIndexer = variable_list_df.loc["Latitude"]["Variable"]
        self.TimeName = variable_list_df.loc["Time"]["Variable"]
        self.VerticalCoordIndexer = variable_list_df.loc["Vertical Level"]["Variable"]
        self.PressureData = data.level
        self.time = data.time
        self._row_labels = {}
        dev = _device(args)
        self.engine = LECEngine(data.lat, data.lon, data.level, device=dev)
        limits = boxes_limits if boxes_limits is not None else [(western_limit, eastern_limit, southern_limit, northern_limit)]
        self.per_step_boxes = boxes_limits is not None
        self.boxes = [self.engine.box_from_limits(*lim) for lim in limits]
        iw, ie, js, jn = self.boxes[0]
        self.western_limit, self.eastern_limit = float(data.lon[iw]), float(data.lon[ie])
        self.southern_limit, self.northern_limit = float(data.lat[js]), float(data.lat[jn])

        n_steps, nl = len(data.time), len(data.level)
        shard = self.shard = getattr(args, "shard", None)
        t0, t1, h0, h1 = (0, n_steps, 0, n_steps) if shard is None else shard.ranges(n_steps)
        self.t_own, self.t_held = (t0, t1), (h0, h1)
        width = LECEngine.packed_width(nl)
        out, merge, gat = None, None, None
        if shard is not None:
            # one extra column carries the NaN counter of the step, so that ONE gather moves everything rank 0 needs
            from .parallel import SeriesGatherer
            gat = SeriesGatherer(n_steps, width + 1, dev, group=shard.group, dst=0, slots=1, force=True)
            out = gat.send(0)[:, :width]
            merge = shard.merge_dropmask if not self.per_step_boxes else None

        from . import ingest
        from .ingest import StreamedDataset
        if isinstance(data, StreamedDataset):
            # device ingest: the file bytes are streamed, decoded, sorted and cropped on the GPU (ingest.py); in the moving
            # framework dT/dt is differentiated on the device over the (track-selected) time axis
            if dTdt is not None:
                raise NotImplementedError("the device ingest differentiates T in time itself; do not pass a dTdt cube")
            self.ingest_stats = {}
            # the moving framework's 850-hPa diagnostics take their three slices from the cubes this pass decodes (no second read of the
            # file); only where the namelist's units leave the values as they are in the file, so that both paths see the same numbers
            plain_units = all(ds.field_scale(variable_list_df, r) == 1.0 for r in ("Eastward Wind Component", "Northward Wind Component"))
            keep = 85000.0 if (boxes_limits is not None and plain_units and 85000.0 in data.level) else None
            with ingest.refusals():       # (what --ingest auto may fall back from: an input the streamed path declines, nothing later)
                self.result: LECResult = ingest.lec_streamed(data.raw, data.plan, variable_list_df, limits, per_step_boxes=boxes_limits is not None,
                                                             device=dev, chunk_steps=data.chunk_steps, stats=self.ingest_stats, inflate=data.inflate,
                                                             t_range=None if shard is None else (t0, t1), merge_dropmask=merge, out=out, keep_level=keep)
            self.level_slices = self.ingest_stats.pop("level_slices", None)
        else:
            self.result = self._compute_resident(data, variable_list_df, dev, dTdt, merge, out)
        if shard is not None:
            gat.send(0)[:, width] = self.result.nanflag.to(torch.float64)
            gat.start(0)
            series = gat.finish(0)
            torch.cuda.synchronize(dev)
            if series is None:                   # not the rank that writes the results
                self.result = self.scalars = self.levels = self.nanflag = None
                return
            self.result = LECResult(scalars=series[:, :16], levels=series[:, 16:width].unflatten(1, (21, nl)),
                                    nanflag=series[:, width].to(torch.int32), packed=series[:, :width])
        torch.cuda.synchronize(dev)
        self.scalars = self.result.scalars_dict()
        self.levels = self.result.levels_dict()
        self.nanflag = self.result.nanflag.cpu().numpy()

    def row_labels(self, method: str):
        """The time-stamp column of the per-level tables (formatted once per series, not once per table)."""
        got = self._row_labels.get(method)
        if got is None:
            got = self._row_labels[method] = time_labels(self.time, method)
        return got

    def _compute_resident(self, data: ds.LECDataset, variable_list_df: pd.DataFrame, dev, dTdt, merge=None, out=None) -> LECResult:
        """Host-prepared cubes, uploaded whole (a time-sharded rank: the steps it holds)."""
        (t0, t1), (h0, h1) = self.t_own, self.t_held
        n_steps = len(data.time)
        held = getattr(data, "t_held", None) or (0, n_steps)
        if self.shard is not None and tuple(held) == (0, n_steps) and (h0, h1) != (0, n_steps):
            data = data.held_steps((h0, h1))         # a whole data set handed to a sharded BoxData: keep this rank's steps only
            held = (h0, h1)
        if tuple(held) != (h0, h1):
            raise ValueError(f"data set holds time steps {held}, this rank needs {(h0, h1)}")
        geo_role = "Geopotential" if "Geopotential" in variable_list_df.index else "Geopotential Height"
        roles = ["Air Temperature", "Eastward Wind Component", "Northward Wind Component", "Omega Velocity", geo_role]
        arrays = []
        for role in roles:
            a = data.variables[str(variable_list_df.loc[role]["Variable"])]
            if a.dtype.kind != "f":                     # an unpacked integer variable: xarray would compute with it as it is
                a = a.astype(np.float32 if a.dtype.itemsize <= 2 else np.float64)
            scale = ds.field_scale(variable_list_df, role)
            if role != geo_role and scale != 1.0:
                a = a * a.dtype.type(scale)
            arrays.append(a)
        # a file may mix dtypes (float32 T, u, v with an int16-packed z that decodes to float64): the engine wants one storage
        # dtype, the widest of them -- widening is exact, and all arithmetic is fp64 anyway (xarray would promote pairwise)
        common = np.result_type(*[a.dtype for a in arrays])
        phi_scale = ds.field_scale(variable_list_df, geo_role)
        if self.per_step_boxes and dTdt is None:
            return self._compute_resident_packed(arrays, common, data, dev, phi_scale, merge, out)
        cubes = [torch.as_tensor(np.ascontiguousarray(a, dtype=common)).to(dev) for a in arrays]
        dTdt_dev = None
        if dTdt is not None:
            dTdt = np.asarray(dTdt)
            if dTdt.shape[0] == n_steps and (h0, h1) != (0, n_steps):
                dTdt = dTdt[h0:h1]
            dTdt_dev = torch.as_tensor(np.ascontiguousarray(dTdt, dtype=common)).to(dev)
        boxes = self.boxes[t0:t1] if self.per_step_boxes else self.boxes
        if self.per_step_boxes and self.shard is not None:
            # the records of every rank have the row count of the tallest box of the WHOLE series (as the one-process run's)
            nyb = max(b[3] - b[2] + 1 for b in self.boxes)
            boxes = self.engine.prepare_boxes(boxes, nyb_min=nyb)
        return self.engine.compute(cubes[0], cubes[1], cubes[2], cubes[3], cubes[4], boxes,
                                   time_s=data.time_s[h0:h1] if dTdt is None else None, dTdt=dTdt_dev, phi_scale=phi_scale,
                                   t_begin=t0 - h0, t_count=t1 - t0, per_step_boxes=self.per_step_boxes,
                                   drop_any_time=not self.per_step_boxes, merge_dropmask=merge, out=out)

    def _compute_resident_packed(self, arrays, common, data, dev, phi_scale, merge, out) -> LECResult:
        """The moving framework on host-prepared data: the reference slices every step's box out of the crop (box_data.py:297-310);
        here the host does that slice BEFORE the upload -- each step's box to the origin of its slab, T also from the two neighbouring
        steps -- so a tenth of the crop crosses the link and stage 1 gets the box-packed series the streamed path hands it too
        (include/lec_hip.h; fp64 storage: dT/dt as a cube, `lec_dtdt`; fp32: the two neighbours).  Same records as the crop, bit for bit."""
        (t0, t1), (h0, h1) = self.t_own, self.t_held
        boxes = self.boxes[t0:t1]
        nyb = max(b[3] - b[2] + 1 for b in self.boxes)       # (the records of every rank have the row count of the tallest box of the WHOLE series)
        nxb = max(b[1] - b[0] + 1 for b in self.boxes)
        nl = arrays[0].shape[1]

        def pack(a, shift=0):
            p = np.zeros((t1 - t0, nl, nyb, nxb), dtype=common)
            for i, (iw, ie, js, jn) in enumerate(boxes):
                ts = min(max(t0 + i + shift, h0), h1 - 1) - h0       # the step itself where the series has no neighbour (coefficient 0)
                p[i, :, : jn - js + 1, : ie - iw + 1] = a[ts, :, js: jn + 1, iw: ie + 1]
            return torch.as_tensor(p).to(dev)

        f = [pack(a) for a in arrays]
        tm, tp = pack(arrays[0], -1), pack(arrays[0], +1)
        tcoef = self.engine.time_coefs_device(data.time_s[h0:h1])[t0 - h0: t1 - h0].contiguous()
        pb = self.engine.prepare_boxes(boxes, nyb_min=nyb, packed=True)
        kw = dict(dTdt=self.engine.time_stencil(tm, f[0], tp, tcoef)) if common == np.float64 else dict(tm=tm, tp=tp, tcoef=tcoef)
        return self.engine.compute(f[0], f[1], f[2], f[3], f[4], pb, phi_scale=phi_scale, t_begin=0, t_count=t1 - t0, per_step_boxes=True,
                                   drop_any_time=False, merge_dropmask=merge, out=out, **kw)


def time_labels(times, method: str):
    """The first field of every row of a per-level table, as the reference's pandas call prints it, as one byte string of
    fixed-width labels: the moving framework formats its index itself ("%Y-%m-%d %H:%M:%S", conversion_terms.py:299-300); the fixed
    framework leaves a DatetimeIndex to pandas, which prints dates alone when every stamp is a midnight -- taken from pandas itself
    (an index without columns through to_csv) so that the rule is pandas' own."""
    idx = pd.DatetimeIndex(times)
    if method == "fixed":
        labels = pd.DataFrame(index=idx).to_csv(header=None).splitlines()
    else:
        labels = list(idx.strftime("%Y-%m-%d %H:%M:%S"))
    width = len(labels[0]) if labels else 0
    if any(len(x) != width for x in labels) or any(("," in x or '"' in x) for x in labels):
        return None, 0                           # (stamps pandas prints at mixed widths: the caller falls back to pandas itself)
    return "".join(labels).encode("ascii"), width


def format_level_table(table, labels) -> bytes:
    """`table` [time, level] (float64) as the lines DataFrame.to_csv(header=None) writes for it (lec_format_csv_rows)."""
    import ctypes as C
    from . import _lib
    blob, width = labels
    a = np.ascontiguousarray(table, dtype=np.float64)
    if a.ndim != 2:
        raise ValueError("a per-level table is [time, level]")
    rows, cols = a.shape
    if blob is None or rows * width != len(blob):
        raise ValueError("one fixed-width label per row")
    lib = _lib.load()
    cap = rows * (width + 27 * cols + 1)
    out = C.create_string_buffer(max(cap, 1))
    n = lib.lec_format_csv_rows(a.ctypes.data, rows, cols, cols, blob, width, C.addressof(out), cap)
    if n < 0:
        _lib.check(1, "lec_format_csv_rows")
    return out.raw[:n]


class _Terms:
    """Shared plumbing of the four analysis classes: hand out a finished series and append its
    per-level CSV rows exactly where the reference's calc_* would (``_save_vertical_levels``,
    e.g. conversion_terms.py:287-308)."""

    def __init__(self, box_obj: BoxData, method: str, app_logger=None):
        if method not in ("fixed", "moving"):
            raise ValueError("method must be 'fixed' or 'moving'")
        self.box_obj, self.method, self.app_logger = box_obj, method, app_logger

    def _save_vertical_levels(self, name: str):
        """One table appended to its CSV file: byte for byte what the reference's pandas call writes (conversion_terms.py:287-308:
        DataFrame.to_csv(mode="a", header=None) -- float cells as repr(float), NaN as an empty field), formatted for the whole
        series in one call of the library (lec_format_csv_rows) instead of one Python object per cell."""
        b = self.box_obj
        path = f"{b.results_subdirectory_vertical_levels}/{name}_{b.VerticalCoordIndexer}.csv"
        table = b.levels[name]
        if self.method == "fixed" and name in ("Cz_1", "Ce_1"):        # level-only term: written transposed (conversion_terms.py:293-294)
            df = pd.DataFrame({b.VerticalCoordIndexer: b.PressureData, name: table[0]}).T
            df.to_csv(path, mode="a", header=None)
            return
        labels = b.row_labels(self.method)
        if labels[0] is None:                    # time stamps pandas prints at mixed widths (sub-second steps): pandas writes the table
            idx = pd.DatetimeIndex(b.time) if self.method == "fixed" else pd.DatetimeIndex(b.time).strftime("%Y-%m-%d %H:%M:%S")
            pd.DataFrame(table, index=idx, columns=b.PressureData).to_csv(path, mode="a", header=None)
            return
        with open(path, "ab") as f:
            f.write(format_level_table(table, labels))

    def _series(self, name: str, tables):
        for t in tables:
            self._save_vertical_levels(t)
        s = self.box_obj.scalars[name]
        return s


class EnergyContents(_Terms):
    """energy_contents.py:99-165"""
    def calc_az(self): return self._series("Az", ["Az"])
    def calc_ae(self): return self._series("Ae", ["Ae"])
    def calc_kz(self): return self._series("Kz", ["Kz"])
    def calc_ke(self): return self._series("Ke", ["Ke"])


class ConversionTerms(_Terms):
    """conversion_terms.py:103-245"""
    def calc_ca(self): return self._series("Ca", ["Ca_1", "Ca_2", "Ca"])
    def calc_ce(self): return self._series("Ce", ["Ce_1", "Ce_2", "Ce"])
    def calc_cz(self): return self._series("Cz", ["Cz_1", "Cz_2", "Cz"])
    def calc_ck(self): return self._series("Ck", ["Ck_1", "Ck_2", "Ck_3", "Ck_4", "Ck_5", "Ck"])


class BoundaryTerms(_Terms):
    """boundary_terms.py:125-418"""
    def calc_baz(self): return self._series("BAz", [])
    def calc_bae(self): return self._series("BAe", [])
    def calc_bkz(self): return self._series("BKz", [])
    def calc_bke(self): return self._series("BKe", [])
    def calc_boz(self): return self._series("BΦZ", [])
    def calc_boe(self): return self._series("BΦE", [])


class GenerationDissipationTerms(_Terms):
    """generation_and_dissipation_terms.py:122-188"""
    def calc_gz(self): return self._series("Gz", ["Gz"])
    def calc_ge(self): return self._series("Ge", ["Ge"])

    def calc_dz(self):
        raise NotImplementedError("Dz needs friction velocities; the reference marks it unfinished "
                                  "(generation_and_dissipation_terms.py:158) -- run with -r")

    calc_de = calc_dz


def _create_level_csvs(directory, time_name, vert_name, level_pa):
    """lec_fixed_framework.py:172-197 / lec_moving_framework.py:583-611"""
    for term in LEVEL_TERMS:
        columns = [time_name] + [float(i) for i in level_pa]
        pd.DataFrame(columns=columns).to_csv(Path(directory, f"{term}_{vert_name}.csv"), index=None)


def _log_ingest(box_obj, app_logger):
    st = getattr(box_obj, "ingest_stats", None)
    if st:
        app_logger.info("Device ingest: %.1f MB over the link in %d chunk(s) of %d step(s), staging %s, inflate %s, storage %s, device buffers %.2f GB" % (
            st["bytes_moved"] / 1e6, st["chunks"], st["chunk_steps"], st["staging"], st.get("inflate", "none"), st["storage"],
            st.get("device_buffer_bytes", 0) / 1e9))
        if "seconds" in st:
            app_logger.info("Device ingest seconds: " + ", ".join(f"{k} {v:.3f}" for k, v in st["seconds"].items()) +
                            (f"; {st['register_calls']} registrations, {st['registered_bytes'] / 1e9:.2f} GB" if "register_calls" in st else ""))


def _compute_all(box_obj, method, app_logger):
    _log_ingest(box_obj, app_logger)
    ec = EnergyContents(box_obj, method, app_logger)
    out = {"Az": ec.calc_az(), "Ae": ec.calc_ae(), "Kz": ec.calc_kz(), "Ke": ec.calc_ke()}
    ct = ConversionTerms(box_obj, method, app_logger)
    out.update({"Cz": ct.calc_cz(), "Ca": ct.calc_ca(), "Ck": ct.calc_ck(), "Ce": ct.calc_ce()})
    bt = BoundaryTerms(box_obj, method, app_logger)
    out.update({"BAz": bt.calc_baz(), "BAe": bt.calc_bae(), "BKz": bt.calc_bkz(), "BKe": bt.calc_bke(),
                "BΦZ": bt.calc_boz(), "BΦE": bt.calc_boe()})
    gd = GenerationDissipationTerms(box_obj, method, app_logger)
    out.update({"Gz": gd.calc_gz(), "Ge": gd.calc_ge()})
    return out


def lec_fixed(data: ds.LECDataset, variable_list_df: pd.DataFrame, results_subdirectory: str,
              results_subdirectory_vertical_levels: str, app_logger, args):
    """Eulerian framework, lec_fixed_framework.py:30-303 (plots are out of scope)."""
    app_logger.info("Computing energetics using fixed framework (MI355X HIP engine)...")
    min_lon, max_lon, min_lat, max_lat = ds.read_box_limits(args.box_limits)
    time_name = variable_list_df.loc["Time"]["Variable"]
    vert_name = variable_list_df.loc["Vertical Level"]["Variable"]
    app_logger.info(f"Bounding box: lon=[{min_lon}, {max_lon}], lat=[{min_lat}, {max_lat}]")
    shard = getattr(args, "shard", None)
    root = shard is None or shard.root
    if root:
        _create_level_csvs(results_subdirectory_vertical_levels, time_name, vert_name, data.level)
    try:
        box_obj = BoxData(data, variable_list_df, min_lon, max_lon, min_lat, max_lat, args, results_subdirectory,
                          results_subdirectory_vertical_levels)
    except Exception:
        app_logger.exception("An exception occurred while creating BoxData object")
        raise
    phases.mark("ingest_compute_gather")
    if box_obj.result is None:              # time-sharded run: rank 0 holds the gathered series and writes every file
        return None
    if int(box_obj.nanflag.sum()):
        app_logger.warning("NaN level values were interpolated/dropped per time step (_handle_nans semantics)")
    terms = _compute_all(box_obj, "fixed", app_logger)
    app_logger.info("Computed energy, conversion, boundary and generation terms")
    df = pd.DataFrame(index=pd.DatetimeIndex(data.time))
    for col in FIXED_COLUMNS:                       # BΦZ / BΦE are computed, then dropped (:252-253,:287-290)
        df[col] = terms[col]
    full = budgets_and_residuals({c: df[c].values for c in df.columns}, data.time_s, residuals=True)
    for col in full:
        if col not in df.columns:
            df[col] = full[col]
    if getattr(args, "outname", None):
        results_filename = args.outname
    else:
        results_filename = os.path.basename(args.infile).split(".nc")[0] + "_fixed_results"
    results_file = Path(results_subdirectory, f"{results_filename}.csv")
    df.to_csv(results_file)
    phases.mark("csv_writes")
    app_logger.info(f"Results saved to {results_file}")
    if getattr(args, "plots", False):
        app_logger.warning("-p/--plots: plotting is out of scope of the MI355X engine; the CSVs feed the reference's plot scripts unchanged")
    return df


def get_limits(track: pd.DataFrame, t):
    """get_limits (lec_moving_framework.py:199-266), track branch: nearest track row in time,
    default 15 x 15 degree box unless the track has width/length columns."""
    row = track.iloc[int(np.argmin(np.abs(track.index - t)))]
    clat, clon = float(row["Lat"]), float(row["Lon"])
    width, length = row.get("width", 15), row.get("length", 15)
    return {"datestr": pd.to_datetime(t).strftime("%Y-%m-%d-%H%M"), "central_lat": clat, "central_lon": clon,
            "length": length, "width": width, "min_lon": clon - width / 2, "max_lon": clon + width / 2,
            "min_lat": clat - length / 2, "max_lat": clat + length / 2}


def lec_moving(data: ds.LECDataset, variable_list_df: pd.DataFrame, dTdt, results_subdirectory: str,
               figures_directory: str, results_subdirectory_vertical_levels: str, app_logger, args):
    """Semi-Lagrangian framework, lec_moving_framework.py:546-745.  ``dTdt`` may be None: the engine then
    differentiates T over the dataset's (track-selected) time axis on the device, which is what
    run_lec_analysis computes (lorenzcycletoolkit.py:184-186)."""
    app_logger.info("Computing energetics using moving framework (MI355X HIP engine)...")
    if not getattr(args, "track", False):
        raise NotImplementedError("only -t/--track is supported; -c/--choose needs an interactive map")
    time_name = variable_list_df.loc["Time"]["Variable"]
    vert_name = variable_list_df.loc["Vertical Level"]["Variable"]
    shard = getattr(args, "shard", None)
    root = shard is None or shard.root
    if root:
        _create_level_csvs(results_subdirectory_vertical_levels, time_name, vert_name, data.level)
    times = pd.DatetimeIndex(data.time)
    track = ds.read_track(args.trackfile, app_logger)
    # handle_track_file (lec_moving_framework.py:58-160)
    if track.index[0] < times.min() or track.index[-1] > times.max():
        raise ValueError("Track time limits do not match with data time limits.")
    for name, coord in (("Lon", data.lon), ("Lat", data.lat)):
        word = "longitude" if name == "Lon" else "latitude"
        if track[name].max() > coord.max():
            raise ValueError(f"Track file {word} max limit ({track[name].max():.2f}) exceeds data max {word} limit ({float(coord.max()):.2f}).")
        if track[name].min() < coord.min():
            raise ValueError(f"Track file {word} min limit ({track[name].min():.2f}) is below data min {word} limit ({float(coord.min()):.2f}).")
    if 85000.0 not in data.level:
        raise KeyError(85000)                                   # lec_moving_framework.py:653-657 selects 85000 Pa exactly
    limits = [get_limits(track, t) for t in times]
    boxes = [(l["min_lon"], l["max_lon"], l["min_lat"], l["max_lat"]) for l in limits]
    box_obj = BoxData(data, variable_list_df, args=args, results_subdirectory=results_subdirectory,
                      results_subdirectory_vertical_levels=results_subdirectory_vertical_levels, dTdt=dTdt,
                      boxes_limits=boxes)
    phases.mark("ingest_compute_gather")
    # 850-hPa diagnostics of every box (lec_moving_framework.py:650-709); a time-sharded rank does its own steps, rank 0 gets them all
    from .diagnostics import track_diagnostics
    form = getattr(args, "vorticity_form", None) or "metpy_no_crs"
    positions = track_diagnostics(data, variable_list_df, limits, track, use_track_zeta=bool(getattr(args, "zeta", False)),
                                  device=_device(args), shard=shard, formulation=form, slices=getattr(box_obj, "level_slices", None))
    phases.mark("track_diagnostics")
    if box_obj.result is None:              # time-sharded run: rank 0 holds the gathered series and writes every file
        return None
    terms = _compute_all(box_obj, "moving", app_logger)
    df = pd.DataFrame({c: terms[c] for c in MOVING_COLUMNS}, index=times, dtype=float)
    full = budgets_and_residuals({c: df[c].values for c in df.columns}, data.time_s,
                                 residuals=bool(getattr(args, "residuals", False)))
    for col in full:
        if col not in df.columns:
            df[col] = full[col]
    method = "track"
    infile_name = os.path.basename(args.infile).split(".nc")[0]
    results_file = os.path.join(results_subdirectory, f"{infile_name}_{method}_results.csv")
    df.to_csv(results_file)
    app_logger.info(f"Results saved to {results_file}")
    # 850-hPa diagnostics of every box (lec_moving_framework.py:650-709); parity unpinned, see diagnostics.py
    app_logger.info(f"850 hPa track diagnostics (min_max_zeta_850, min_hgt_850, max_wind_850) on the GPU (lec_track_diag), vorticity formulation "
                    f"'{form}' (" + ("plain dv/dx - du/dy on great-circle grid distances, a = 6370997 m: MetPy 1.6.2 for DataArrays without a CRS, as the "
                                     "reference passes them" if form == "metpy_no_crs" else "spherical: dv/dx - du/dy + u tan(phi) / Re") +
                    "); NOT pinned against MetPy itself (--vorticity-form selects the other formulation)")
    out_track = pd.DataFrame([{**l, **p} for l, p in zip(limits, positions)])
    out_track = out_track.rename(columns={"datestr": "time", "central_lat": "Lat", "central_lon": "Lon"})
    out_track.to_csv(os.path.join(results_subdirectory, f"{infile_name}_{method}_trackfile"), index=False, sep=";")
    phases.mark("csv_writes")
    return results_file, df
