"""
CPU oracle for the 850-hPa track diagnostics of the moving framework  --  TEST INFRASTRUCTURE ONLY (imported by tests/ only).

Restates, independently of lorenzcycletoolkit_amd/diagnostics.py:

* ``get_position`` (src/frameworks/lec_moving_framework.py:269-417) and ``find_extremum_coordinates``
  (src/utils/tools.py:95-128): box selection by inclusive label slices, track-file values taking precedence, the hemisphere
  rule for the vorticity extremum -- these are the reference's own code, so this part is pinned by its source.  One quirk is
  kept here and NOT in the product: the reference finds the *positions* with ``np.array(data).argmin()/argmax()``, which land on
  the first NaN cell when the box holds a NaN, while it takes the *values* with xarray's NaN-skipping ``.min()/.max()``.
* the two MetPy 1.6.2 calls that feed it (``lec_moving_framework.py:660-663``): ``wind_speed`` = sqrt(u^2 + v^2) and
  ``vorticity`` on a latitude / longitude grid.  MetPy is a third-party dependency that is neither vendored in /root/reference
  nor installed here, so its published algorithm is restated: three-point finite differences on unequally spaced
  points, second order also at the ends (``metpy.calc.first_derivative``), in the two forms MetPy >= 1.5 has:
    - ``vorticity_no_crs``: DataArrays WITHOUT a CRS (what the reference passes: plain xr.open_dataset): plain
      dv/dx - du/dy, the spacings from ``lat_lon_grid_deltas`` = geodesic arcs between neighbouring grid points on
      pyproj's default sphere (a = 6,370,997 m) -- evaluated here point by point, row by row;
    - ``vorticity_sphere``: data WITH a CRS, through the parallel / meridional scale factors:
      zeta = 1 / (a cos(phi)) dv/dlambda - 1 / a du/dphi + u tan(phi) / a.
  **Parity unpinned** for these three columns: the reference's only sample trackfile has them
  empty (tests/golden/Reg1_track/*_trackfile) and MetPy cannot be run here.
"""
from __future__ import annotations

import numpy as np

from .lec_oracle import RE


def first_derivative_3pt(f: np.ndarray, x: np.ndarray, axis: int) -> np.ndarray:
    """metpy.calc.first_derivative for coordinate values `x`: the derivative of the parabola through three neighbouring points,
    evaluated at the middle one in the interior and at the end point itself at either end (written point by point)."""
    f = np.moveaxis(np.asarray(f, dtype=np.float64), axis, 0)
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    out = np.empty_like(f)

    def parabola_slope(i0, at):
        x0, x1, x2 = x[i0], x[i0 + 1], x[i0 + 2]
        f0, f1, f2 = f[i0], f[i0 + 1], f[i0 + 2]
        xa = x[at]
        # Lagrange basis derivatives at xa
        l0 = ((xa - x1) + (xa - x2)) / ((x0 - x1) * (x0 - x2))
        l1 = ((xa - x0) + (xa - x2)) / ((x1 - x0) * (x1 - x2))
        l2 = ((xa - x0) + (xa - x1)) / ((x2 - x0) * (x2 - x1))
        return l0 * f0 + l1 * f1 + l2 * f2

    out[0] = parabola_slope(0, 0)
    for i in range(1, n - 1):
        out[i] = parabola_slope(i - 1, i)
    out[n - 1] = parabola_slope(n - 3, n - 1)
    return np.moveaxis(out, 0, axis)


def vorticity_sphere(u, v, lat_deg, lon_deg, radius=RE):
    """Relative vorticity of [..., lat, lon] fields on a sphere."""
    phi, lam = np.deg2rad(np.asarray(lat_deg, dtype=np.float64)), np.deg2rad(np.asarray(lon_deg, dtype=np.float64))
    u = np.asarray(u, dtype=np.float64)
    dv_dlam = first_derivative_3pt(v, lam, -1)
    du_dphi = first_derivative_3pt(u, phi, -2)
    c, t = np.cos(phi)[:, None], np.tan(phi)[:, None]
    return dv_dlam / (radius * c) - du_dphi / radius + u * t / radius


def vorticity_no_crs(u, v, lat_deg, lon_deg, radius=6370997.0):
    """Plain dv/dx - du/dy of [..., lat, lon] fields with MetPy's grid distances for data without a CRS: dx[j][i] = the great-circle
    arc from (lon_i, lat_j) to (lon_i+1, lat_j), dy[j] = the arc from lat_j to lat_j+1 along a meridian (sphere of pyproj's default
    radius), and first_derivative on the running sums of those spacings, one row / one column at a time."""
    phi, lam = np.deg2rad(np.asarray(lat_deg, dtype=np.float64)), np.deg2rad(np.asarray(lon_deg, dtype=np.float64))
    u, v = np.asarray(u, dtype=np.float64), np.asarray(v, dtype=np.float64)

    def arc(p1, l1, p2, l2):       # central angle by the haversine formula (an independent form of the product's atan2 one)
        h = np.sin((p2 - p1) / 2) ** 2 + np.cos(p1) * np.cos(p2) * np.sin((l2 - l1) / 2) ** 2
        return 2 * radius * np.arcsin(np.sqrt(h))

    dvdx = np.empty_like(v)
    for j in range(phi.size):
        x = np.concatenate([[0.0], np.cumsum([arc(phi[j], lam[i], phi[j], lam[i + 1]) for i in range(lam.size - 1)])])
        dvdx[..., j, :] = first_derivative_3pt(v[..., j, :], x, -1)
    y = np.concatenate([[0.0], np.cumsum([arc(phi[j], 0.0, phi[j + 1], 0.0) for j in range(phi.size - 1)])])
    dudy = first_derivative_3pt(u, y, -2)
    return dvdx - dudy


def wind_speed(u, v):
    return np.sqrt(np.asarray(u, dtype=np.float64) ** 2 + np.asarray(v, dtype=np.float64) ** 2)


def get_position(zeta, hgt, wspd, lat, lon, limits, track_row=None, use_zeta=False):
    """get_position + find_extremum_coordinates for one time step ([lat, lon] arrays; `limits` as get_limits builds them).
    Returns the flat dict of the nine trackfile columns."""
    lat, lon = np.asarray(lat), np.asarray(lon)
    jj = np.flatnonzero((lat >= limits["min_lat"]) & (lat <= limits["max_lat"]))      # .sel(slice(min, max)): inclusive label slices
    ii = np.flatnonzero((lon >= limits["min_lon"]) & (lon <= limits["max_lon"]))
    z, h, w = (a[np.ix_(jj, ii)] for a in (zeta, hgt, wspd))
    south = limits["min_lat"] < 0
    has = lambda c: track_row is not None and c in track_row.index
    if has("min_max_zeta_850"):
        zval = float(track_row["min_max_zeta_850"])
    elif use_zeta and track_row is not None:
        j0 = int(np.argmin(np.abs(lat - limits["central_lat"]))); i0 = int(np.argmin(np.abs(lon - limits["central_lon"])))
        zval = float(zeta[j0, i0])
    else:
        zval = float(np.nanmin(z) if south else np.nanmax(z))
    valid = lambda c: has(c) and not np.isnan(float(track_row[c]))
    hval = float(track_row["min_hgt_850"]) if valid("min_hgt_850") else float(np.nanmin(h))          # xarray .min() skips NaN
    wval = float(track_row["max_wind_850"]) if valid("max_wind_850") else float(np.nanmax(w))

    def where(a, use_min):                                    # find_extremum_coordinates: plain argmin / argmax (NaN wins)
        idx = np.unravel_index(a.argmin() if use_min else a.argmax(), a.shape)
        return float(lat[jj][idx[0]]), float(lon[ii][idx[1]])
    zlat, zlon = where(z, lat[jj].min() < 0)
    hlat, hlon = where(h, True)
    wlat, wlon = where(w, False)
    return {"min_max_zeta_850_lat": zlat, "min_max_zeta_850_lon": zlon, "min_max_zeta_850": zval,
            "min_hgt_850_lat": hlat, "min_hgt_850_lon": hlon, "min_hgt_850": hval,
            "max_wind_850_lat": wlat, "max_wind_850_lon": wlon, "max_wind_850": wval}
