"""Input side of the drop-in: namelist / box_limits / track parsing, a NetCDF-3 reader and the
preprocessing semantics of the reference's ``prepare_data`` (src/utils/preprocessing.py:149-413,
src/utils/select_area.py:254-338, src/utils/validation.py), without xarray.

Only the plumbing the LEC path needs is reproduced; the arrays stay NumPy on the host until the
framework moves them to the GPU.  Classic NetCDF-3 (CDF-1/2/5) files are read through scipy, NetCDF-4 / HDF5
files through the pure-Python reader in hdf5_lite.py (the image has no HDF5 library).
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import pandas as pd

from .constants import G

REQUIRED_ROLES = ["Air Temperature", "Omega Velocity", "Eastward Wind Component", "Northward Wind Component",
                  "Longitude", "Latitude", "Time", "Vertical Level"]

# factor to SI for the units the preset namelists use (pint did this in the reference: box_data.py:297-310)
_UNIT_SCALE = {
    "K": 1.0, "kelvin": 1.0, "m/s": 1.0, "m s-1": 1.0, "m s**-1": 1.0, "Pa/s": 1.0, "Pa s-1": 1.0, "Pa s**-1": 1.0,
    "hPa/s": 100.0, "m**2/s**2": 1.0, "m2/s2": 1.0, "m**2 s**-2": 1.0, "meter ** 2 / second ** 2": 1.0, "m": 1.0, "gpm": 1.0,
    "dam": 10.0, "meter": 1.0, "kelvin": 1.0, "meter / second": 1.0, "pascal / second": 1.0,
}
_LEVEL_SCALE = {"pa": 1.0, "hpa": 100.0, "mb": 100.0, "mbar": 100.0, "millibar": 100.0, "millibars": 100.0}


@dataclass
class LECDataset:
    """The role xarray.Dataset plays in the reference: variables on [time, level, lat, lon] plus
    their coordinates, addressed by the names the namelist gives them."""
    variables: Dict[str, np.ndarray]
    lat: np.ndarray
    lon: np.ndarray
    level: np.ndarray                 # Pa, ascending, after process_data
    time: np.ndarray                  # datetime64[ns]
    names: Dict[str, str] = field(default_factory=dict)   # role -> variable / coordinate name
    level_units: Optional[str] = None
    t_held: Optional[tuple] = None    # time-sharded run: the variables hold only the steps [h0, h1) of `time` (own steps + T halo)

    @property
    def time_s(self) -> np.ndarray:
        """Seconds since the first time step (xarray's datetime_to_numeric with datetime_unit='s')."""
        return (self.time - self.time.min()) / np.timedelta64(1, "s")

    def isel(self, t=None, j=None, i=None) -> "LECDataset":
        t = slice(None) if t is None else t
        j = slice(None) if j is None else j
        i = slice(None) if i is None else i
        if self.t_held is not None and not (isinstance(t, slice) and t == slice(None)):
            raise ValueError("a time-sharded data set cannot be re-sliced in time")
        v = {k: np.ascontiguousarray(a[t][:, :, j][:, :, :, i]) for k, a in self.variables.items()}
        return LECDataset(v, self.lat[j], self.lon[i], self.level, self.time[t], dict(self.names), self.level_units, self.t_held)

    def held_steps(self, t_held) -> "LECDataset":
        """The same data set holding only the time steps [h0, h1) of its (unchanged) time axis: a rank's share of a sharded run."""
        h0, h1 = t_held
        v = {k: np.ascontiguousarray(a[h0:h1]) for k, a in self.variables.items()}
        return LECDataset(v, self.lat, self.lon, self.level, self.time, dict(self.names), self.level_units, (int(h0), int(h1)))


# --------------------------------------------------------------------------------------------
# the three ';'-separated input files
# --------------------------------------------------------------------------------------------
def read_namelist(path: str, app_logger=None) -> pd.DataFrame:
    """validate_namelist_file (validation.py:167-244): ';'-separated, rows keyed by role."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"Namelist file not found: {path}")
    try:
        df = pd.read_csv(path, sep=";", index_col=0, header=0)
    except pd.errors.EmptyDataError:
        raise ValueError(f"Namelist file is empty: {path}")
    if "Variable" not in df.columns:
        raise ValueError(f"Namelist file missing 'Variable' column: {path}")
    missing = [r for r in REQUIRED_ROLES if r not in df.index]
    if "Geopotential" not in df.index and "Geopotential Height" not in df.index:
        missing.append("Geopotential or Geopotential Height")
    if missing:
        raise ValueError(f"Namelist file missing required entries: {missing}")
    return df


def read_box_limits(path: str):
    """lec_fixed's box_limits handling (lec_fixed_framework.py:59-154)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"Box limits file not found: {os.path.abspath(path)}. "
                                f"Create one or use --box_limits to specify path.")
    df = pd.read_csv(path, header=None, delimiter=";", index_col=0)
    missing = [f for f in ("min_lon", "max_lon", "min_lat", "max_lat") if f not in df.index]
    if missing:
        raise ValueError(f"Box limits file missing required fields: {missing}. Found: {list(df.index)}")
    w, e = float(df.loc["min_lon"].iloc[0]), float(df.loc["max_lon"].iloc[0])
    s, n = float(df.loc["min_lat"].iloc[0]), float(df.loc["max_lat"].iloc[0])
    if w > e:
        raise ValueError(f"Invalid box_limits: min_lon ({w}) > max_lon ({e}). Check {path}")
    if s > n:
        raise ValueError(f"Invalid box_limits: min_lat ({s}) > max_lat ({n}). Check {path}")
    return w, e, s, n


def read_track(path: str, app_logger=None) -> pd.DataFrame:
    """validate_track_file + the read in process_data (validation.py:28-164, preprocessing.py:171-182):
    ';' (or ',') separated, columns time;Lat;Lon[;length;width][...], time as YYYY-MM-DD-HHMM."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"Track file not found: {path}")
    with open(path) as f:
        first, second = f.readline().strip(), f.readline().strip()
    if ";" in first:
        delim = ";"
    elif "," in first:
        delim = ","
        if app_logger:
            app_logger.warning("Track file uses ',' as delimiter instead of the standard ';'")
    else:
        raise ValueError(f"Invalid track file format. Header should contain ';' or ',' separators.\nFound: {first}")
    cols = [c.strip() for c in first.split(delim)]
    missing = [c for c in ("time", "Lat", "Lon") if c not in cols]
    if missing:
        raise ValueError(f"Track file missing required columns: {missing}\nExpected: ['time', 'Lat', 'Lon']\nFound: {cols}")
    if second:
        d = second.split(delim)[0].strip()
        if not re.match(r"^\d{4}-\d{2}-\d{2}-\d{4}$", d):
            raise ValueError(f"Invalid date format in track file: '{d}'\nExpected: YYYY-MM-DD-HHMM (e.g., 2005-08-08-0000)")
    track = pd.read_csv(path, delimiter=delim)
    track["time"] = pd.to_datetime(track["time"], format="%Y-%m-%d-%H%M")
    return track.set_index("time")


# --------------------------------------------------------------------------------------------
# NetCDF-3 reader
# --------------------------------------------------------------------------------------------
def _decode_time(values: np.ndarray, units: str) -> np.ndarray:
    m = re.match(r"\s*(\w+)\s+since\s+(.+)", units)
    if not m:
        raise ValueError(f"cannot decode time units '{units}'")
    unit, origin = m.group(1).lower(), m.group(2).strip()
    origin = pd.Timestamp(re.sub(r"\s*(UTC|Z)$", "", origin))
    per = {"seconds": "s", "second": "s", "minutes": "m", "minute": "m", "hours": "h", "hour": "h", "days": "D", "day": "D"}[unit]
    secs = {"s": 1.0, "m": 60.0, "h": 3600.0, "D": 86400.0}[per]
    # go through integer seconds so that 1800-based hour counts do not overflow 64-bit nanoseconds early
    off = np.round(np.asarray(values, dtype=np.float64) * secs).astype("int64")
    return (np.datetime64(origin.to_datetime64(), "s") + off.astype("timedelta64[s]")).astype("datetime64[ns]")


FIELD_ROLES = ("Air Temperature", "Omega Velocity", "Eastward Wind Component", "Northward Wind Component")


class _NcVar:
    """One variable of either container format: ``data`` (ndarray, memory map, or a lazy HDF5 variable that is indexed
    by time step), ``dimensions`` and attributes."""

    def __init__(self, data, dimensions, attrs):
        self.data, self.dimensions, self._attrs = data, tuple(dimensions), attrs

    def attr(self, name):
        a = self._attrs(name)
        if isinstance(a, bytes):
            a = a.decode()
        if isinstance(a, np.ndarray) and a.size == 1:
            a = a.reshape(()).item()
        return a

    def values(self) -> np.ndarray:
        """The whole variable in native byte order."""
        a = self.data.read() if hasattr(self.data, "read") else np.array(self.data)
        return a.astype(a.dtype.newbyteorder("="))


class _Container:
    """Classic NetCDF (scipy) or NetCDF-4 / HDF5 (hdf5_lite) behind one face."""

    def __init__(self, path: str, mmap: bool):
        if not os.path.exists(path):
            raise FileNotFoundError(f"Input file not found: {path}")
        with open(path, "rb") as f:
            magic = f.read(8)
        self.mapped = mmap
        if magic[:3] == b"CDF":
            from scipy.io import netcdf_file
            self._nc = netcdf_file(path, mmap=mmap)
            self.kind = "netcdf3"
            self.variables = {n: _NcVar(v.data, v.dimensions, (lambda name, v=v: getattr(v, name, None)))
                              for n, v in self._nc.variables.items()}
        elif magic == b"\x89HDF\r\n\x1a\n":
            from .hdf5_lite import H5File
            self._nc = H5File(path)
            self.kind = "hdf5"
            self.variables = {n: _NcVar(v, v.dims, (lambda name, v=v: v.attrs.get(name)))
                              for n, v in self._nc.variables.items()}
        else:
            raise ValueError(f"{path}: neither a classic NetCDF file nor an HDF5 (NetCDF-4) file (magic {magic[:4]!r})")

    def close(self):
        self.variables = {}
        if self.kind == "netcdf3" and self.mapped:
            _close_mapped(self._nc)
        else:
            self._nc.close()


def _close_mapped(nc):
    """Closes a memory-mapped netcdf_file; scipy warns when views of the map are still alive (the map goes with them)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        try:
            nc.close()
        except Exception:
            pass


MPAS_DROPPED_DIM = "standard_height"


def _drop_mpas_dims(nc, path: str, app_logger=None) -> None:
    """-m / --mpas: the reference drops the MPAS-BR post-processor's ``standard_height`` dimension and with it every variable on it
    (``data.drop_dims("standard_height")``, preprocessing.py:367-368), so the analysis reads isobaric variables only; a namelist
    that names one of the dropped variables then fails with a KeyError, as it does there."""
    gone = [n for n, v in nc.variables.items() if MPAS_DROPPED_DIM in tuple(v.dimensions)]
    for n in gone:
        del nc.variables[n]
    if app_logger is not None:
        app_logger.info(f"--mpas: dropped the '{MPAS_DROPPED_DIM}' dimension of {path}" +
                        (f" and the variables on it: {', '.join(gone)}" if gone else " (no variable uses it)"))


def _open_nc(path: str, variable_list_df: pd.DataFrame, mmap: bool, mpas: bool = False, app_logger=None):
    """Opens a data file and checks it against the namelist (get_data, preprocessing.py:35-146;
    validate_variable_match / validate_required_coordinates, validation.py:247-356)."""
    nc = _Container(path, mmap)
    if mpas:
        _drop_mpas_dims(nc, path, app_logger)
    var = lambda role: str(variable_list_df.loc[role]["Variable"])
    names = {r: var(r) for r in REQUIRED_ROLES}
    geo_role = "Geopotential" if "Geopotential" in variable_list_df.index else "Geopotential Height"
    names[geo_role] = var(geo_role)
    missing = [f"{r} -> {n}" for r, n in names.items() if n not in nc.variables]
    if missing:
        have = list(nc.variables)
        nc.close()
        raise KeyError(f"namelist variables not found in {path}: {missing}; file has {have}")
    native = lambda n: nc.variables[n].values()
    lat, lon, lev = native(names["Latitude"]), native(names["Longitude"]), native(names["Vertical Level"])
    time = _decode_time(native(names["Time"]), nc.variables[names["Time"]].attr("units"))
    level_units = nc.variables[names["Vertical Level"]].attr("units")
    want = (names["Time"], names["Vertical Level"], names["Latitude"], names["Longitude"])
    return nc, names, geo_role, lat, lon, lev, time, level_units, want


def _packing(v):
    """CF packing attributes of a variable: (scale_factor, add_offset, fill value) with None for absent ones.
    ``_FillValue`` and ``missing_value`` both mark missing data (xarray's CFMaskCoder masks either); files that give two
    different values are refused rather than half-handled."""
    scale, offset = v.attr("scale_factor"), v.attr("add_offset")
    fill, miss = v.attr("_FillValue"), v.attr("missing_value")
    if fill is not None and miss is not None and float(fill) != float(miss) and not (np.isnan(float(fill)) or np.isnan(float(miss))):
        raise ValueError("variable has different _FillValue and missing_value attributes; give it one fill value")
    if fill is None or (isinstance(fill, float) and np.isnan(fill)):
        fill = miss
    if fill is not None and np.isnan(float(fill)):
        fill = None                                  # a NaN fill value masks nothing that is not NaN already
    return (None if scale is None else float(scale), None if offset is None else float(offset),
            None if fill is None else float(fill))


def decode_dtypes(src_dtype, scale, offset, fill):
    """(dtype after masking, dtype after unpacking) of a file variable, as the reference's pinned xarray 2024.2.0 decodes it
    (requirements.txt:88; the data reach BoxData in exactly this dtype and the reference computes in it):

    * CFMaskCoder.decode -> dtypes.maybe_promote: integers become float32 (itemsize <= 2) or float64 when a fill value is
      present, floats keep their dtype;
    * then CFScaleOffsetCoder.decode -> _choose_float_dtype(masked dtype, has add_offset): float32 (and float16) stay
      float32; integers of <= 2 bytes WITHOUT an add_offset become float32; everything else float64.

    So int16 data with scale_factor + add_offset + _FillValue (ERA5) are float32 in the reference; without a fill value they
    are float64; a scale_factor alone gives float32."""
    d = np.dtype(src_dtype).newbyteorder("=")
    if fill is not None and d.kind in "iu":
        d = np.dtype(np.float32 if d.itemsize <= 2 else np.float64)
    masked = d
    if scale is not None or offset is not None:
        if d.kind == "f" and d.itemsize <= 4:
            d = np.dtype(np.float32)
        elif d.kind in "iu" and d.itemsize <= 2 and offset is None:
            d = np.dtype(np.float32)
        else:
            d = np.dtype(np.float64)
    elif d.kind in "iu":
        d = np.dtype(np.float32 if d.itemsize <= 2 else np.float64)      # the engine needs floating-point cubes
    return masked, d


def decode_values(raw: np.ndarray, scale, offset, fill) -> np.ndarray:
    """Native-byte-order file values -> decoded values, in the dtype and with the roundings of the reference's decode
    (see decode_dtypes): mask first, then ``data = data.astype(dtype); data *= scale_factor; data += add_offset`` with the
    attributes as float64 scalars (NumPy >= 2: a float32 array times a float64 scalar is computed in float64 and rounded
    back to float32 by the in-place store, once per operation)."""
    masked, out = decode_dtypes(raw.dtype, scale, offset, fill)
    a = raw
    if fill is not None:
        a = np.where(raw == fill, np.nan, raw.astype(masked)) if masked.kind == "f" else raw
    if scale is not None or offset is not None:
        # every operation in float64, rounded to the decode dtype once per operation -- written out, so that it does not depend on
        # the NumPy version's promotion rules (NumPy 1.x would compute `float32_array *= float64_scalar` in float32) and agrees with
        # lec_ingest's (float)((double)v * scale), (float)((double)v + offset) bit for bit
        a = a.astype(out, copy=True)
        if scale is not None:
            a = (a.astype(np.float64) * np.float64(scale)).astype(out)
        if offset is not None:
            a = (a.astype(np.float64) + np.float64(offset)).astype(out)
    return np.asarray(a, dtype=out)


def open_dataset(path: str, variable_list_df: pd.DataFrame, mpas: bool = False, app_logger=None) -> LECDataset:
    """Host-side decode of the whole file with the semantics of ``xr.open_dataset`` (decode_values)."""
    nc, names, geo_role, lat, lon, lev, time, level_units, want = _open_nc(path, variable_list_df, mmap=False, mpas=mpas, app_logger=app_logger)
    variables = {}
    for role in FIELD_ROLES + (geo_role,):
        v = nc.variables[names[role]]
        raw = v.values()
        scale, offset, fill = _packing(v)
        a = decode_values(raw, scale, offset, fill)
        if set(v.dimensions) != set(want):
            raise ValueError(f"{names[role]} has dimensions {v.dimensions}, expected {want}")
        variables[names[role]] = np.transpose(a, [v.dimensions.index(d) for d in want])
    nc.close()
    return LECDataset(variables, lat, lon, lev, time, names, level_units)


@dataclass
class RawVariable:
    """One file variable as stored: memory-mapped, file byte order, possibly CF-packed."""
    data: np.ndarray                  # [time, level, lat, lon] in FILE order of each axis
    scale_factor: Optional[float]
    add_offset: Optional[float]
    fill_value: Optional[float]


@dataclass
class RawDataset:
    """The undecoded file: what the device ingest (ingest.py) streams to the GPU."""
    variables: Dict[str, RawVariable]
    lat: np.ndarray
    lon: np.ndarray
    level: np.ndarray
    time: np.ndarray
    names: Dict[str, str]
    level_units: Optional[str]
    geo_role: str
    _nc: object = None                # keeps the memory map alive

    def close(self):
        if self._nc is not None:
            self.variables.clear()
            self._nc.close()
            self._nc = None


def open_raw(path: str, variable_list_df: pd.DataFrame, mpas: bool = False, app_logger=None) -> RawDataset:
    """Like open_dataset, but nothing is decoded or copied: the variables stay memory-mapped file bytes."""
    nc, names, geo_role, lat, lon, lev, time, level_units, want = _open_nc(path, variable_list_df, mmap=True, mpas=mpas, app_logger=app_logger)
    variables = {}
    for role in FIELD_ROLES + (geo_role,):
        v = nc.variables[names[role]]
        if tuple(v.dimensions) != want:
            nc.close()
            raise ValueError(f"{names[role]} has dimensions {v.dimensions}; the device ingest needs {want} order")
        if v.data.dtype.kind not in "if" or (v.data.dtype.kind, v.data.dtype.itemsize) not in (("i", 1), ("i", 2), ("i", 4), ("f", 4), ("f", 8)):
            nc.close()
            raise ValueError(f"{names[role]}: the device ingest reads int8, int16, int32, float32 and float64 variables, not {v.data.dtype}")
        scale, offset, fill = _packing(v)
        variables[names[role]] = RawVariable(v.data, scale, offset, fill)
    return RawDataset(variables, lat, lon, lev, time, names, level_units, geo_role, nc)


# --------------------------------------------------------------------------------------------
# process_data / slice_domain
# --------------------------------------------------------------------------------------------
@dataclass
class ProcessIndex:
    """What process_data does to the axes, as index maps (sorted axis -> file axis)."""
    tpos: Optional[np.ndarray]        # selected time steps (track times) or None for all
    ik: np.ndarray                    # kept levels, ascending in Pa
    ij: np.ndarray                    # latitudes S -> N
    io: np.ndarray                    # longitudes W -> E after the wrap to -180..180
    lat: np.ndarray
    lon: np.ndarray
    level: np.ndarray                 # Pa
    time: np.ndarray


def process_index(lat, lon, lev, time, level_units, names, args, app_logger=None) -> ProcessIndex:
    """process_data (preprocessing.py:149-371) on the coordinates only: track-time selection, 0..360 -> -180..180
    longitudes, level -> Pa, sort lon / level / lat ascending, drop levels above 10 hPa."""
    tpos = None
    if getattr(args, "track", False):
        track = read_track(args.trackfile, app_logger)
        data_dt = int((time[1] - time[0]) / np.timedelta64(1, "h"))
        track_dt = int((track.index[1] - track.index[0]) / np.timedelta64(1, "h"))
        if data_dt > track_dt:
            raise ValueError(f"Data time step ({data_dt}h) is higher than track time step ({track_dt}h). "
                             "Cannot select track timesteps that don't exist in data. "
                             "Please resample the track or re-download data with higher temporal resolution.")
        if track.index[0] < time[0]:
            raise ValueError(f"Track initial timestamp ({track.index[0]}) is earlier than data initial timestamp "
                             f"({time[0]}). Please adjust the track file.")
        if track.index[-1] > time[-1]:
            raise ValueError(f"Track final timestamp ({track.index[-1]}) is later than data final timestamp "
                             f"({time[-1]}). Please adjust the track file or re-download the data.")
        tpos = pd.Index(time).get_indexer(track.index.values)
        if np.any(tpos < 0):
            raise KeyError(f"track times not found in the data: {list(track.index[tpos < 0])}")
        time = time[tpos]
    if lon.min() < -180 or lon.max() > 180:
        lon = (lon + 180) % 360 - 180                                   # tools.py:76-92
    key = (level_units or "hPa").strip().lower()
    if level_units is None and app_logger:
        app_logger.warning(f"Vertical level coordinate has no units attribute. Assuming hPa (hectopascals).")
    if key not in _LEVEL_SCALE:
        raise ValueError(f"Cannot convert vertical level units to Pa. Check if '{names.get('Vertical Level')}' "
                         "has valid pressure units.")
    lev = lev.astype(np.float64) * _LEVEL_SCALE[key]
    io, ik, ij = np.argsort(lon, kind="stable"), np.argsort(lev, kind="stable"), np.argsort(lat, kind="stable")
    keep = lev[ik] >= 1000.0                                            # preprocessing.py:364-365
    return ProcessIndex(tpos, ik[keep], ij, io, lat[ij], lon[io], lev[ik][keep], time)


def process_data(data: LECDataset, args, variable_list_df: pd.DataFrame, app_logger=None) -> LECDataset:
    """process_data (preprocessing.py:149-371): track-time selection, 0..360 -> -180..180 longitudes,
    level -> Pa, sort lon / level / lat ascending, drop levels above 10 hPa."""
    px = process_index(data.lat, data.lon, data.level, data.time, data.level_units, data.names, args, app_logger)
    v = data.variables if px.tpos is None else {k: a[px.tpos] for k, a in data.variables.items()}
    v = {k: np.ascontiguousarray(a[:, px.ik][:, :, px.ij][:, :, :, px.io]) for k, a in v.items()}
    return LECDataset(v, px.lat, px.lon, px.level, px.time, dict(data.names), "Pa")


def domain_slices(lat: np.ndarray, lon: np.ndarray, args):
    """slice_domain (select_area.py:254-338) on sorted coordinates: (lat slice, lon slice).  Fixed -> nearest-point
    crop from the hard-coded inputs/box_limits; track -> label slice of the track extent +- (half the largest
    box + one grid step)."""
    from .tables import nearest_index
    if getattr(args, "fixed", False):
        w, e, s, n = read_box_limits("inputs/box_limits")
        iw, ie = nearest_index(lon, w), nearest_index(lon, e)
        js, jn = nearest_index(lat, s), nearest_index(lat, n)
        return slice(js, jn + 1), slice(iw, ie + 1)
    if getattr(args, "track", False):
        dx, dy = lon[1] - lon[0], lat[1] - lat[0]
        track = read_track(args.trackfile or "inputs/track")
        if "width" in track.columns:
            mw, ml = track["width"].max(), track["length"].max()
        else:
            mw, ml = 15, 15
        w, e = track["Lon"].min() - mw / 2 - dx, track["Lon"].max() + mw / 2 + dx
        s, n = track["Lat"].min() - ml / 2 - dy, track["Lat"].max() + ml / 2 + dy
        ii = np.flatnonzero((lon >= w) & (lon <= e))
        jj = np.flatnonzero((lat >= s) & (lat <= n))
        if ii.size < 2 or jj.size < 2:
            raise ValueError("track extent selects fewer than 2 grid points of the data")
        return slice(jj[0], jj[-1] + 1), slice(ii[0], ii[-1] + 1)
    raise NotImplementedError("the interactive -c/--choose domain selection needs a GUI and is out of scope")


def slice_domain(data: LECDataset, args, variable_list_df: pd.DataFrame) -> LECDataset:
    """slice_domain (select_area.py:254-338), see domain_slices."""
    js, is_ = domain_slices(data.lat, data.lon, args)
    return data.isel(j=js, i=is_)


def field_scale(variable_list_df: pd.DataFrame, role: str) -> float:
    """Factor from the namelist's units to SI for one role (the reference converts with pint,
    box_data.py:297-310; geopotential height is additionally multiplied by g, box_data.py:233-241)."""
    units = str(variable_list_df.loc[role]["Units"]).strip()
    if units not in _UNIT_SCALE:
        raise ValueError(f"Unit error in {role}: unsupported units '{units}'")
    return _UNIT_SCALE[units] * (G if role == "Geopotential Height" else 1.0)


@dataclass
class IngestPlan:
    """Index maps from the analysis domain (sorted, cropped) to file positions, plus the domain's coordinates."""
    tsel: np.ndarray        # file time index of every processed time step
    kmap: np.ndarray        # int32 [nl]
    jmap: np.ndarray        # int32 [ny]
    imap: np.ndarray        # int32 [nx]
    lat: np.ndarray
    lon: np.ndarray
    level: np.ndarray       # Pa
    time: np.ndarray        # datetime64[ns]

    @property
    def time_s(self) -> np.ndarray:
        return (self.time - self.time.min()) / np.timedelta64(1, "s")


def make_plan(raw: "RawDataset", args, app_logger=None) -> IngestPlan:
    """process_data + slice_domain (preprocessing.py:149-371, select_area.py:254-338) as index maps."""
    px = process_index(raw.lat, raw.lon, raw.level, raw.time, raw.level_units, raw.names, args, app_logger)
    js, is_ = domain_slices(px.lat, px.lon, args)
    tsel = np.arange(raw.time.size) if px.tpos is None else np.asarray(px.tpos)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    return IngestPlan(tsel, i32(px.ik), i32(px.ij[js]), i32(px.io[is_]), px.lat[js], px.lon[is_], px.level, px.time)



def gather_on_host(var: RawVariable, plan: IngestPlan, t_range=None) -> np.ndarray:
    """The analysis-domain cube of one raw variable, decoded -- what lec_ingest does on the GPU, one time step at a time
    on the host, so that only the selected time steps and the cropped domain are ever read from a large file.
    ``t_range``: only the processed time steps [a, b) (a rank's share of a time-sharded run)."""
    out = None
    tsel = plan.tsel if t_range is None else plan.tsel[t_range[0]: t_range[1]]
    for n, ft in enumerate(tsel):
        if hasattr(var.data, "read_step"):          # a lazily inflated NetCDF-4 variable: only the chunks of the wanted levels
            raw = var.data.read_step(int(ft), plan.kmap)[:, plan.jmap][:, :, plan.imap]
        else:
            raw = np.asarray(var.data[int(ft)])[plan.kmap][:, plan.jmap][:, :, plan.imap]
        raw = raw.astype(raw.dtype.newbyteorder("="))
        a = decode_values(raw, var.scale_factor, var.add_offset, var.fill_value)
        if out is None:
            out = np.empty((len(tsel),) + a.shape, dtype=a.dtype)
        out[n] = a
    return out


def prepare_data(args, varlist: str = "inputs/namelist", app_logger=None) -> LECDataset:
    """prepare_data (preprocessing.py:374-413): open, process, crop.  The reference relies on xarray's lazy indexing to touch
    only what the analysis needs; here the index maps are built from the coordinates first (make_plan) and the variables are
    gathered time step by time step.  Files whose variables are not in (time, level, lat, lon) order take the whole-file
    path (open_dataset + process_data + slice_domain), which gives the same arrays."""
    if getattr(args, "cdsapi", False):
        raise NotImplementedError("--cdsapi downloads need network access and are out of scope")
    variable_list_df = read_namelist(varlist, app_logger)
    shard = getattr(args, "shard", None)          # time-sharded run (parallel.ShardContext): this rank decodes its own steps + halo only
    mpas = bool(getattr(args, "mpas", False))
    try:
        raw = open_raw(args.infile, variable_list_df, mpas=mpas, app_logger=app_logger)
    except ValueError as e:
        if "order" not in str(e) and "device ingest reads" not in str(e):
            raise
        data = open_dataset(args.infile, variable_list_df, mpas=mpas)
        data = slice_domain(process_data(data, args, variable_list_df, app_logger), args, variable_list_df)
        return data if shard is None else data.held_steps(shard.ranges(len(data.time))[2:])
    try:
        plan = make_plan(raw, args, app_logger)
        held = None if shard is None else shard.ranges(len(plan.tsel))[2:]
        if app_logger is not None and any(getattr(v.data, "_filters", None) for v in raw.variables.values()):
            app_logger.info("The input is a filtered (deflated) NetCDF-4 file: its chunks are being inflated on the host's threads; "
                            "--device-ingest inflates them on the GPU instead (same results, several times faster on large files)")
        variables = {name: gather_on_host(var, plan, held) for name, var in raw.variables.items()}
        return LECDataset(variables, plan.lat, plan.lon, plan.level, plan.time, dict(raw.names), "Pa", held)
    finally:
        raw.close()
