"""
CPU oracle for the DATA PREPARATION the reference does before the LEC hot path  --  TEST INFRASTRUCTURE ONLY.

Only ``tests/`` may import this module (never the shipped package).  It restates, independently of
``lorenzcycletoolkit_amd/dataset.py`` and of the ``lec_ingest`` kernel:

* what ``xr.open_dataset`` (the reference's ``get_data``, src/utils/preprocessing.py:73-74) does to a variable: the CF
  decode of **xarray == 2024.2.0** (pinned in the reference's requirements.txt:88) running on **numpy == 2.0.0**
  (requirements.txt:47).  xarray is a third-party dependency that is neither vendored under /root/reference nor
  installed in this image, so its published algorithm is restated here, function by function:
    - ``xarray/coding/variables.py``  ``CFMaskCoder.decode`` -> ``_apply_mask``            (fill values -> NaN, dtype promotion)
    - ``xarray/core/dtypes.py``        ``maybe_promote``                                    (integers: float32 if itemsize <= 2 else float64)
    - ``xarray/coding/variables.py``  ``CFScaleOffsetCoder.decode`` -> ``_choose_float_dtype``, ``_scale_offset_decoding``
    - ``xarray/coding/times.py``       ``decode_cf_datetime`` for "<unit> since <origin>" with the standard calendar
  In that version the mask coder runs FIRST and the scale/offset coder picks its dtype from the already-masked data
  (the interaction that xarray changed in 2024.03.0, PR #8713): int16 + scale_factor + add_offset is float64 without a
  fill value and float32 with one.
* ``process_data`` (src/utils/preprocessing.py:275-365: longitude wrap through tools.py:76-92, level -> Pa, the three
  sorts, the >= 10 hPa selection) and ``slice_domain`` (src/utils/select_area.py:272-336) on the decoded arrays.

Parity status of THIS module: the axis handling is pinned by the reference's float32 samples through
tests/test_oracle_golden.py (the same steps as ``lec_oracle.load_ncep_sample``); the decode of PACKED variables is
**parity unpinned** -- the reference ships no packed input file (samples/testdata_ERA5.nc is missing from the snapshot,
.MISSING_LARGE_BLOBS) and xarray cannot be run here -- it is a restatement of the published algorithm only.
"""
from __future__ import annotations

import re
from typing import Dict, Optional, Tuple

import numpy as np

from . import lec_oracle as o


# ---------------------------------------------------------------------------------------------------------------------
# xarray 2024.2.0 decode
# ---------------------------------------------------------------------------------------------------------------------
def maybe_promote(dtype: np.dtype) -> np.dtype:
    """xarray/core/dtypes.py maybe_promote: the dtype that can hold NaN as the missing value."""
    dtype = np.dtype(dtype)
    if np.issubdtype(dtype, np.floating):
        return dtype
    if np.issubdtype(dtype, np.integer):
        return np.dtype(np.float32) if dtype.itemsize <= 2 else np.dtype(np.float64)
    raise TypeError(f"not a numeric field dtype: {dtype}")


def choose_float_dtype(dtype: np.dtype, has_offset: bool) -> np.dtype:
    """xarray/coding/variables.py _choose_float_dtype (2024.2.0): a float dtype that can represent `dtype` values."""
    dtype = np.dtype(dtype)
    if dtype.itemsize <= 4 and np.issubdtype(dtype, np.floating):
        return np.dtype(np.float32)                 # float32 stays, float16 is widened
    if dtype.itemsize <= 2 and np.issubdtype(dtype, np.integer):
        if not has_offset:                          # "a scale factor is entirely safe ... any offset at all -> float64"
            return np.dtype(np.float32)
    return np.dtype(np.float64)


def decode_cf_variable(raw: np.ndarray, attrs: Dict[str, object]) -> np.ndarray:
    """The two coders xr.open_dataset applies to a numeric data variable, in their order: mask, then scale/offset.
    ``raw`` in native byte order; ``attrs`` may hold _FillValue, missing_value, scale_factor, add_offset."""
    data = np.asarray(raw)
    data = data.astype(data.dtype.newbyteorder("="))
    # --- CFMaskCoder.decode
    fills = []
    for key in ("_FillValue", "missing_value"):
        fv = attrs.get(key)
        if fv is not None:
            for x in np.atleast_1d(fv):
                if not (isinstance(x, (float, np.floating)) and np.isnan(x)):
                    fills.append(x)
    if fills:
        dtype = maybe_promote(data.dtype)
        promoted = np.asarray(data, dtype=dtype)        # _apply_mask: data = np.asarray(data, dtype=dtype)
        cond = np.zeros(promoted.shape, dtype=bool)
        for fv in fills:
            cond |= promoted == fv
        data = np.where(cond, np.array(np.nan, dtype=dtype), promoted)
    # --- CFScaleOffsetCoder.decode
    scale, offset = attrs.get("scale_factor"), attrs.get("add_offset")
    if scale is not None or offset is not None:
        dtype = choose_float_dtype(data.dtype, offset is not None)
        data = data.astype(dtype, copy=True)            # _scale_offset_decoding
        if scale is not None:
            data *= np.float64(np.asarray(scale).reshape(()))      # file attributes are float64 scalars: NumPy 2 computes in fp64, stores in `dtype`
        if offset is not None:
            data += np.float64(np.asarray(offset).reshape(()))
    return data


def decode_cf_time(values: np.ndarray, units: str) -> np.ndarray:
    """decode_cf_datetime for "<unit> since <origin>" in the standard calendar -> datetime64[ns]."""
    import pandas as pd
    m = re.match(r"\s*(\w+)\s+since\s+(.+)", units)
    unit, origin = m.group(1).lower().rstrip("s"), re.sub(r"\s*(UTC|Z)$", "", m.group(2).strip())
    secs = {"second": 1, "minute": 60, "hour": 3600, "day": 86400}[unit]
    ns = np.round(np.asarray(values, dtype=np.float64) * secs * 1e9).astype("int64")
    return pd.Timestamp(origin).to_datetime64().astype("datetime64[ns]") + ns.astype("timedelta64[ns]")


# ---------------------------------------------------------------------------------------------------------------------
# a classic NetCDF file as xr.open_dataset would present it
# ---------------------------------------------------------------------------------------------------------------------
def open_classic(path: str) -> Tuple[Dict[str, np.ndarray], Dict[str, Tuple[str, ...]], Dict[str, Dict[str, object]]]:
    """(decoded variables, dimensions, attributes) of every variable of a NetCDF-3 file (scipy reads the container)."""
    from scipy.io import netcdf_file
    nc = netcdf_file(path, mmap=False)
    out, dims, attrs = {}, {}, {}
    for name, v in nc.variables.items():
        a = {k: getattr(v, k) for k in ("_FillValue", "missing_value", "scale_factor", "add_offset", "units") if hasattr(v, k)}
        attrs[name] = {k: (val.decode() if isinstance(val, bytes) else val) for k, val in a.items()}
        raw = np.array(v.data)
        if "units" in attrs[name] and " since " in str(attrs[name]["units"]):
            out[name] = decode_cf_time(raw.astype(raw.dtype.newbyteorder("=")), attrs[name]["units"])
        else:
            out[name] = decode_cf_variable(raw, attrs[name])
        dims[name] = tuple(v.dimensions)
    nc.close()
    return out, dims, attrs


_LEVEL_TO_PA = {"pa": 1.0, "hpa": 100.0, "mb": 100.0, "mbar": 100.0, "millibar": 100.0, "millibars": 100.0}


def prepare(path: str, names: Dict[str, str], fixed_limits=None, track: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]] = None,
            geo_is_height: bool = False, max_width=15, max_length=15) -> o.Domain:
    """get_data + process_data + slice_domain of the reference for a classic NetCDF file.

    ``names``: role -> variable name for tair, u, v, omega, geo, lat, lon, level, time (what the namelist says).
    ``fixed_limits``: (west, east, south, north) -> the fixed framework's nearest-point crop (select_area.py:272-295);
    ``track``: (times datetime64, lats, lons) -> track-time selection (preprocessing.py:273) and the track-extent crop
    (select_area.py:297-313).  Returns the arrays BoxData would see, geopotential already in m2 s-2."""
    return prepare_opened(open_classic(path), names, fixed_limits, track, geo_is_height, max_width, max_length)


def prepare_opened(opened, names: Dict[str, str], fixed_limits=None, track=None, geo_is_height: bool = False, max_width=15,
                   max_length=15) -> o.Domain:
    """``prepare`` on an already opened data set ``(decoded variables, dimensions, attributes)`` -- what ``xr.open_dataset`` presents,
    whatever container it came from (tests build the triple of a NetCDF-4 fixture from the arrays its writer wrote)."""
    var, dims, attrs = opened
    want = (names["time"], names["level"], names["lat"], names["lon"])
    f = {r: np.transpose(var[names[r]], [dims[names[r]].index(d) for d in want]) for r in ("tair", "u", "v", "omega", "geo")}
    lat, lon, lev, time = var[names["lat"]], var[names["lon"]], var[names["level"]], var[names["time"]]
    if track is not None:                                         # data.sel(time=track.index.values), preprocessing.py:273
        pos = np.array([int(np.flatnonzero(time == t)[0]) for t in track[0]])
        time = time[pos]
        f = {k: a[pos] for k, a in f.items()}
    if lon.min() < -180 or lon.max() > 180:                       # convert_longitude_range, tools.py:76-92
        lon = (lon + 180) % 360 - 180
    unit = str(attrs[names["level"]].get("units", "hPa")).strip().lower()
    level = lev.astype(np.float64) * _LEVEL_TO_PA[unit]           # pint: levels * units(...) -> Pa (preprocessing.py:301-314)
    io, ik, ij = np.argsort(lon, kind="stable"), np.argsort(level, kind="stable"), np.argsort(lat, kind="stable")
    lon, level, lat = lon[io], level[ik], lat[ij]                  # sortby lon, level, lat (preprocessing.py:358-362)
    f = {k: a[:, :, :, io][:, ik][:, :, ij] for k, a in f.items()}
    keep = (level >= 1000.0) & (level <= level.max())              # sel(level=slice(1000, lowest_level)) (preprocessing.py:364-365)
    level = level[keep]
    f = {k: a[:, keep] for k, a in f.items()}
    if fixed_limits is not None:
        w, e, s, n = fixed_limits
        iw, ie = o.select_nearest(lon, w), o.select_nearest(lon, e)
        js, jn = o.select_nearest(lat, s), o.select_nearest(lat, n)
        jj, ii = slice(js, jn + 1), slice(iw, ie + 1)
    elif track is not None:
        dx, dy = lon[1] - lon[0], lat[1] - lat[0]
        ii = np.flatnonzero((lon >= track[2].min() - max_width / 2 - dx) & (lon <= track[2].max() + max_width / 2 + dx))
        jj = np.flatnonzero((lat >= track[1].min() - max_length / 2 - dy) & (lat <= track[1].max() + max_length / 2 + dy))
        jj, ii = slice(jj[0], jj[-1] + 1), slice(ii[0], ii[-1] + 1)
    else:
        jj, ii = slice(None), slice(None)
    f = {k: np.ascontiguousarray(a[:, :, jj][:, :, :, ii]) for k, a in f.items()}
    geopt = f["geo"] * o.G if geo_is_height else f["geo"]          # box_data.py:233-241
    time_s = (time - time.min()) / np.timedelta64(1, "s")
    return o.Domain(f["tair"], f["u"], f["v"], f["omega"], geopt, lat[jj], lon[ii], level, time_s.astype(np.float64))
