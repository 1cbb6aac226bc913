"""Input side of the drop-in: namelist / box_limits / track parsing, a NetCDF-3 reader and the
preprocessing semantics of the reference's ``prepare_data`` (src/utils/preprocessing.py:149-413,
src/utils/select_area.py:254-338, src/utils/validation.py), without xarray.

Only the plumbing the LEC path needs is reproduced; the arrays stay NumPy on the host until the
framework moves them to the GPU.  NetCDF-4/HDF5 files need an HDF5 reader that this image lacks
(SURVEY.md section 8f-1): classic NetCDF-3 (CDF-1/2/5) files are read through scipy.
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
    "hPa/s": 100.0, "m**2/s**2": 1.0, "m2/s2": 1.0, "m**2 s**-2": 1.0, "m": 1.0, "gpm": 1.0, "dam": 10.0,
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

    @property
    def time_s(self) -> np.ndarray:
        """Seconds since the first time step (xarray's datetime_to_numeric with datetime_unit='s')."""
        return (self.time - self.time.min()) / np.timedelta64(1, "s")

    def isel(self, t=None, j=None, i=None) -> "LECDataset":
        t = slice(None) if t is None else t
        j = slice(None) if j is None else j
        i = slice(None) if i is None else i
        v = {k: np.ascontiguousarray(a[t][:, :, j][:, :, :, i]) for k, a in self.variables.items()}
        return LECDataset(v, self.lat[j], self.lon[i], self.level, self.time[t], dict(self.names), self.level_units)


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


def open_dataset(path: str, variable_list_df: pd.DataFrame) -> LECDataset:
    """get_data (preprocessing.py:35-146) for classic NetCDF files, then validate_variable_match /
    validate_required_coordinates (validation.py:247-356)."""
    from scipy.io import netcdf_file

    if not os.path.exists(path):
        raise FileNotFoundError(f"Input file not found: {path}")
    with open(path, "rb") as f:
        magic = f.read(4)
    if magic[:3] != b"CDF":
        raise ValueError(f"{path}: not a classic NetCDF-3 file (magic {magic!r}); NetCDF-4/HDF5 input needs an HDF5 "
                         "reader that is not available in this environment (convert with `nccopy -k classic`)")
    nc = netcdf_file(path, mmap=False)
    var = lambda role: str(variable_list_df.loc[role]["Variable"])
    names = {r: var(r) for r in REQUIRED_ROLES}
    geo_role = "Geopotential" if "Geopotential" in variable_list_df.index else "Geopotential Height"
    names[geo_role] = var(geo_role)
    missing = [f"{r} -> {n}" for r, n in names.items() if n not in nc.variables]
    if missing:
        raise KeyError(f"namelist variables not found in {path}: {missing}; file has {list(nc.variables)}")
    native = lambda n: np.array(nc.variables[n].data).astype(nc.variables[n].data.dtype.newbyteorder("="))
    lat, lon, lev = native(names["Latitude"]), native(names["Longitude"]), native(names["Vertical Level"])
    tv = nc.variables[names["Time"]]
    time = _decode_time(native(names["Time"]), tv.units.decode() if isinstance(tv.units, bytes) else str(tv.units))
    lv = nc.variables[names["Vertical Level"]]
    level_units = getattr(lv, "units", None)
    if isinstance(level_units, bytes):
        level_units = level_units.decode()
    want = (names["Time"], names["Vertical Level"], names["Latitude"], names["Longitude"])
    variables = {}
    for role in ("Air Temperature", "Omega Velocity", "Eastward Wind Component", "Northward Wind Component", geo_role):
        v = nc.variables[names[role]]
        a = native(names[role])
        scale, offset = getattr(v, "scale_factor", None), getattr(v, "add_offset", None)
        if scale is not None or offset is not None:       # CF packing decodes to float64 (xarray 2024.2)
            a = a.astype(np.float64) * (1.0 if scale is None else float(scale)) + (0.0 if offset is None else float(offset))
        fill = getattr(v, "_FillValue", None)
        if fill is not None and np.issubdtype(a.dtype, np.floating):
            a = np.where(a == fill, np.nan, a)
        if set(v.dimensions) != set(want):
            raise ValueError(f"{names[role]} has dimensions {v.dimensions}, expected {want}")
        variables[names[role]] = np.transpose(a, [v.dimensions.index(d) for d in want])
    nc.close()
    return LECDataset(variables, lat, lon, lev, time, names, level_units)


# --------------------------------------------------------------------------------------------
# process_data / slice_domain
# --------------------------------------------------------------------------------------------
def process_data(data: LECDataset, args, variable_list_df: pd.DataFrame, app_logger=None) -> LECDataset:
    """process_data (preprocessing.py:149-371): track-time selection, 0..360 -> -180..180 longitudes,
    level -> Pa, sort lon / level / lat ascending, drop levels above 10 hPa."""
    v, lat, lon, lev, time = dict(data.variables), data.lat, data.lon, data.level, data.time
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
        pos = pd.Index(time).get_indexer(track.index.values)
        if np.any(pos < 0):
            raise KeyError(f"track times not found in the data: {list(track.index[pos < 0])}")
        v = {k: a[pos] for k, a in v.items()}
        time = time[pos]
    if lon.min() < -180 or lon.max() > 180:
        lon = (lon + 180) % 360 - 180                                   # tools.py:76-92
    key = (data.level_units or "hPa").strip().lower()
    if data.level_units is None and app_logger:
        app_logger.warning(f"Vertical level coordinate has no units attribute. Assuming hPa (hectopascals).")
    if key not in _LEVEL_SCALE:
        raise ValueError(f"Cannot convert vertical level units to Pa. Check if '{data.names.get('Vertical Level')}' "
                         "has valid pressure units.")
    lev = lev.astype(np.float64) * _LEVEL_SCALE[key]
    io, ik, ij = np.argsort(lon, kind="stable"), np.argsort(lev, kind="stable"), np.argsort(lat, kind="stable")
    lon, lev, lat = lon[io], lev[ik], lat[ij]
    keep = lev >= 1000.0                                                # preprocessing.py:364-365
    v = {k: np.ascontiguousarray(a[:, ik][:, keep][:, :, ij][:, :, :, io]) for k, a in v.items()}
    return LECDataset(v, lat, lon, lev[keep], time, dict(data.names), "Pa")


def slice_domain(data: LECDataset, args, variable_list_df: pd.DataFrame) -> LECDataset:
    """slice_domain (select_area.py:254-338): fixed -> nearest-point crop from the hard-coded
    inputs/box_limits; track -> label slice of the track extent +- (half the largest box + one grid step)."""
    from .tables import nearest_index
    if getattr(args, "fixed", False):
        w, e, s, n = read_box_limits("inputs/box_limits")
        iw, ie = nearest_index(data.lon, w), nearest_index(data.lon, e)
        js, jn = nearest_index(data.lat, s), nearest_index(data.lat, n)
        return data.isel(j=slice(js, jn + 1), i=slice(iw, ie + 1))
    if getattr(args, "track", False):
        dx, dy = data.lon[1] - data.lon[0], data.lat[1] - data.lat[0]
        track = read_track(args.trackfile or "inputs/track")
        if "width" in track.columns:
            mw, ml = track["width"].max(), track["length"].max()
        else:
            mw, ml = 15, 15
        w, e = track["Lon"].min() - mw / 2 - dx, track["Lon"].max() + mw / 2 + dx
        s, n = track["Lat"].min() - ml / 2 - dy, track["Lat"].max() + ml / 2 + dy
        ii = np.flatnonzero((data.lon >= w) & (data.lon <= e))
        jj = np.flatnonzero((data.lat >= s) & (data.lat <= n))
        if ii.size < 2 or jj.size < 2:
            raise ValueError("track extent selects fewer than 2 grid points of the data")
        return data.isel(j=slice(jj[0], jj[-1] + 1), i=slice(ii[0], ii[-1] + 1))
    raise NotImplementedError("the interactive -c/--choose domain selection needs a GUI and is out of scope")


def field_scale(variable_list_df: pd.DataFrame, role: str) -> float:
    """Factor from the namelist's units to SI for one role (the reference converts with pint,
    box_data.py:297-310; geopotential height is additionally multiplied by g, box_data.py:233-241)."""
    units = str(variable_list_df.loc[role]["Units"]).strip()
    if units not in _UNIT_SCALE:
        raise ValueError(f"Unit error in {role}: unsupported units '{units}'")
    return _UNIT_SCALE[units] * (G if role == "Geopotential Height" else 1.0)


def prepare_data(args, varlist: str = "inputs/namelist", app_logger=None) -> LECDataset:
    """prepare_data (preprocessing.py:374-413)."""
    if getattr(args, "cdsapi", False):
        raise NotImplementedError("--cdsapi downloads need network access and are out of scope")
    variable_list_df = read_namelist(varlist, app_logger)
    data = open_dataset(args.infile, variable_list_df)
    return slice_domain(process_data(data, args, variable_list_df, app_logger), args, variable_list_df)
