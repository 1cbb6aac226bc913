"""850-hPa track diagnostics of the moving framework (lec_moving_framework.py:269-417,650-709 and
tools.py:95-128 of the reference): relative vorticity, wind speed, and the position of the vorticity
extremum / height minimum / wind maximum inside each time step's box, written to ``*_trackfile``.

Not part of the LEC hot path: a few (time, lat, lon) slices on the host.  Parity UNPINNED against MetPy (SURVEY.md
section 8c): the reference calls MetPy's ``vorticity`` / ``wind_speed`` and its only sample trackfile
has these columns empty; the formulas here are checked against an independent restatement (oracle/track_diagnostics.py,
tests/test_diagnostics_cpu.py), and the box / extremum logic against the reference's own get_position.  One deliberate
difference: extremum POSITIONS skip NaN like the values do (the reference's argmin / argmax land on a NaN cell).  Vorticity here is the spherical form zeta = dv/dx - du/dy + (u/Re) tan(phi) with
dx = Re cos(phi) d(lambda), dy = Re d(phi) and MetPy-style three-point derivatives (second order, also at
the edges); MetPy's default geodesic uses the WGS84 ellipsoid, so values can differ by a few 1e-3 relative.

What a maintainer with MetPy 1.6.2 at hand should check first: the reference opens its files with plain ``xr.open_dataset`` (no
``parse_cf`` / ``assign_crs`` anywhere in it), and MetPy's ``parse_grid_arguments`` falls back to the plain Cartesian
``dv/dx - du/dy`` on geodesic grid distances when a DataArray carries no CRS -- i.e. WITHOUT the curvature term u tan(phi) / Re that the
spherical form here (and MetPy's own map-factor path for data WITH a CRS) includes.  If that is what the reference's runs do, their zeta
differs from this module's by that term (about 1 % of a cyclone's extremum at 30 degrees latitude, more poleward); the positions of the
extrema and the two other columns (height minimum, wind maximum) are unaffected.
"""
from __future__ import annotations

import numpy as np

from .constants import G, RE


def first_derivative(f: np.ndarray, x: np.ndarray, axis: int) -> np.ndarray:
    """Three-point derivative on a possibly non-uniform axis, second-order one-sided at both ends
    (the stencil of metpy.calc.first_derivative)."""
    f = np.moveaxis(np.asarray(f, dtype=np.float64), axis, -1)
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    if n < 3:
        raise ValueError("first_derivative needs at least 3 points")
    out = np.empty_like(f)
    d = np.diff(x)
    d0, d1 = d[:-1], d[1:]                         # spacing left / right of the interior points
    out[..., 1:-1] = (-d1 / (d0 * (d0 + d1)) * f[..., :-2] + (d1 - d0) / (d0 * d1) * f[..., 1:-1]
                      + d0 / (d1 * (d0 + d1)) * f[..., 2:])
    a, b = d[0], d[1]
    out[..., 0] = -(2 * a + b) / (a * (a + b)) * f[..., 0] + (a + b) / (a * b) * f[..., 1] - a / (b * (a + b)) * f[..., 2]
    a, b = d[-2], d[-1]
    out[..., -1] = b / (a * (a + b)) * f[..., -3] - (a + b) / (a * b) * f[..., -2] + (a + 2 * b) / (b * (a + b)) * f[..., -1]
    return np.moveaxis(out, -1, axis)


def vorticity(u: np.ndarray, v: np.ndarray, lat_deg: np.ndarray, lon_deg: np.ndarray) -> np.ndarray:
    """Relative vorticity on a regular lat/lon grid, arrays [..., lat, lon]."""
    phi = np.deg2rad(np.asarray(lat_deg, dtype=np.float64))
    lam = np.deg2rad(np.asarray(lon_deg, dtype=np.float64))
    cosphi = np.cos(phi)[:, None]
    dvdx = first_derivative(v, lam, -1) / (RE * cosphi)
    dudy = first_derivative(u, phi, -2) / RE
    return dvdx - dudy + (np.asarray(u, dtype=np.float64) / RE) * np.tan(phi)[:, None]


def wind_speed(u, v):
    return np.sqrt(np.asarray(u, dtype=np.float64) ** 2 + np.asarray(v, dtype=np.float64) ** 2)


def box_positions(zeta, hgt, wspd, lat_deg, lon_deg, limits, track_row=None, use_track_zeta=False):
    """get_position for one time step.  ``limits``: dict with min/max lat/lon and central_lat/lon.
    Values present (and not NaN) in the track row take precedence, as in the reference."""
    lat, lon = np.asarray(lat_deg), np.asarray(lon_deg)
    jj = np.flatnonzero((lat >= limits["min_lat"]) & (lat <= limits["max_lat"]))     # label slices: inclusive
    ii = np.flatnonzero((lon >= limits["min_lon"]) & (lon <= limits["max_lon"]))
    sl = np.ix_(jj, ii)
    z, h, w = zeta[sl], hgt[sl], wspd[sl]
    south = limits["min_lat"] < 0
    have = lambda name: track_row is not None and name in track_row.index and not np.isnan(float(track_row[name]))

    if have("min_max_zeta_850"):
        zval = float(track_row["min_max_zeta_850"])
    elif use_track_zeta and track_row is not None:
        j0 = int(np.argmin(np.abs(lat - limits["central_lat"])))
        i0 = int(np.argmin(np.abs(lon - limits["central_lon"])))
        zval = float(zeta[j0, i0])
    else:
        zval = float(np.nanmin(z) if south else np.nanmax(z))
    # xarray's .min() / .max() skip NaN (below-ground points at 850 hPa): so do these, values and positions alike
    hval = float(track_row["min_hgt_850"]) if have("min_hgt_850") else float(np.nanmin(h))
    wval = float(track_row["max_wind_850"]) if have("max_wind_850") else float(np.nanmax(w))

    def where(a, use_min):
        if np.isnan(a).all():
            return float("nan"), float("nan")
        idx = np.unravel_index(np.nanargmin(a) if use_min else np.nanargmax(a), a.shape)
        return float(lat[jj][idx[0]]), float(lon[ii][idx[1]])

    zlat, zlon = where(z, lat[jj].min() < 0)
    hlat, hlon = where(h, True)
    wlat, wlon = where(w, False)
    return {
        "min_max_zeta_850_lat": zlat, "min_max_zeta_850_lon": zlon, "min_max_zeta_850": zval,
        "min_hgt_850_lat": hlat, "min_hgt_850_lon": hlon, "min_hgt_850": hval,
        "max_wind_850_lat": wlat, "max_wind_850_lon": wlon, "max_wind_850": wval,
    }


def track_diagnostics(data, variable_list_df, limits_per_step, track=None, use_track_zeta=False):
    """All time steps: u, v, geopotential height at 85000 Pa -> list of position dicts."""
    k850 = int(np.flatnonzero(data.level == 85000.0)[0])
    name = lambda role: str(variable_list_df.loc[role]["Variable"])
    if hasattr(data, "level_slice"):          # streamed data set: three level slices decoded from the mapped file
        get = lambda role: data.level_slice(role, 85000.0)
    else:
        get = lambda role: data.variables[name(role)][:, k850]
    u, v = get("Eastward Wind Component"), get("Northward Wind Component")
    if "Geopotential Height" in variable_list_df.index:
        hgt = get("Geopotential Height").astype(np.float64)
    else:
        hgt = get("Geopotential").astype(np.float64) / G       # -> gpm
    zeta = vorticity(u, v, data.lat, data.lon)
    wspd = wind_speed(u, v)
    out = []
    for t, lim in enumerate(limits_per_step):
        row = None
        if track is not None:
            row = track.iloc[int(np.argmin(np.abs(track.index - data.time[t])))]
        out.append(box_positions(zeta[t], hgt[t], wspd[t], data.lat, data.lon, lim, row, use_track_zeta))
    return out
