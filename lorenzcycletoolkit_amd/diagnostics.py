"""850-hPa track diagnostics of the moving framework (lec_moving_framework.py:269-417,650-709 and
tools.py:95-128 of the reference): relative vorticity, wind speed, and the position of the vorticity
extremum / height minimum / wind maximum inside each time step's box, written to ``*_trackfile``.

The numbers come from the device (``lec_track_diag``, csrc/lec_diag.hip): three 850-hPa slices are uploaded, one workgroup per
time step evaluates vorticity and wind speed in that step's box and reduces the extrema with their grid positions.  The host builds
the derivative stencils (``stencil_tables``), turns box limits into index ranges (``box_ranges``) and applies the reference's
precedence rules (``positions``).  There is no host evaluation of the fields.

Parity UNPINNED against MetPy: the reference calls MetPy's ``vorticity`` / ``wind_speed`` and its only
sample trackfile has these columns empty; the kernel is checked against an independent restatement (oracle/track_diagnostics.py,
tests/test_gpu_diagnostics.py), the box / extremum logic against the reference's own get_position.  One deliberate difference:
extremum POSITIONS skip NaN like the values do (the reference's argmin / argmax land on a NaN cell; INTEGRATION.md lists it).

The vorticity FORMULATION is an argument (``vorticity_tables``; the kernel only applies three-point stencils whose coefficients
carry the metric), because what the reference's runs evaluate cannot be executed here:

* ``"metpy_no_crs"`` (the default): the reference opens its files with plain ``xr.open_dataset`` (no ``parse_cf`` /
  ``assign_crs`` anywhere in it) and hands MetPy 1.6.2 DataArrays WITHOUT a CRS (lec_moving_framework.py:660-663).  MetPy's
  ``parse_grid_arguments`` then falls back to the plain Cartesian ``dv/dx - du/dy`` ("basic cartesian calculation if we don't have
  a CRS"): no map factors, no curvature term, grid distances from ``lat_lon_grid_deltas`` = geodesic arcs between neighbouring
  grid points on pyproj's default sphere (a = 6,370,997 m), three-point ``first_derivative`` on those distances.
* ``"spherical"``: zeta = dv/dx - du/dy + (u / Re) tan(phi) with dx = Re cos(phi) d(lambda), dy = Re d(phi) -- what MetPy's
  map-factor path gives for data WITH a CRS, and the textbook form.  The two differ by the curvature term (about 1 % of a
  cyclone's extremum at 30 degrees latitude, more poleward), by 1.2e-6 in the radius and by the chord-versus-parallel arc
  (a few 1e-6 at 0.25 degrees); positions of extrema and the two other columns (height minimum, wind maximum) are the same.

The log states which one ran; ``--vorticity-form spherical`` on the command line selects the other.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .constants import G


def _three_point(x: np.ndarray) -> np.ndarray:
    """[n][4]: per point the first index of its three-point stencil and the three derivative coefficients on the (possibly
    uneven) coordinate ``x`` -- the parabola through three neighbours, differentiated at the point itself: centred in the
    interior, one-sided (still second order) at both ends (the stencil of metpy.calc.first_derivative)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    if n < 3:
        raise ValueError("three-point derivatives need at least 3 points")
    d = np.diff(x)
    tab = np.empty((n, 4))
    d0, d1 = d[:-1], d[1:]                                   # spacing left / right of the interior points
    tab[1:-1, 0] = np.arange(n - 2)
    tab[1:-1, 1] = -d1 / (d0 * (d0 + d1))
    tab[1:-1, 2] = (d1 - d0) / (d0 * d1)
    tab[1:-1, 3] = d0 / (d1 * (d0 + d1))
    a, b = d[0], d[1]
    tab[0] = (0, -(2 * a + b) / (a * (a + b)), (a + b) / (a * b), -a / (b * (a + b)))
    a, b = d[-2], d[-1]
    tab[-1] = (n - 3, b / (a * (a + b)), -(a + b) / (a * b), (a + 2 * b) / (b * (a + b)))
    return tab


def stencil_tables(lat_deg, lon_deg):
    """(lontab [nx][4], lattab [ny][6]): d/dlambda and d/dphi stencils in 1/rad (first index, three coefficients), cos(phi), tan(phi)."""
    phi = np.deg2rad(np.asarray(lat_deg, dtype=np.float64))
    lam = np.deg2rad(np.asarray(lon_deg, dtype=np.float64))
    lattab = np.empty((phi.size, 6))
    lattab[:, :4] = _three_point(phi)
    lattab[:, 4], lattab[:, 5] = np.cos(phi), np.tan(phi)
    return np.ascontiguousarray(_three_point(lam)), lattab


FORMULATIONS = ("metpy_no_crs", "spherical")
PYPROJ_SPHERE_RADIUS = 6370997.0        # pyproj's Geod(ellps="sphere"): what metpy.calc.lat_lon_grid_deltas measures with when there is no CRS


def _three_point_rows(d: np.ndarray) -> np.ndarray:
    """[ny][nx][3]: ``_three_point`` coefficients for every row of a [ny][nx - 1] array of spacings (metpy.calc.first_derivative with
    ``delta=``: the point positions of a row are the running sum of its spacings)."""
    d = np.asarray(d, dtype=np.float64)
    ny, nx = d.shape[0], d.shape[1] + 1
    if nx < 3:
        raise ValueError("three-point derivatives need at least 3 points")
    co = np.empty((ny, nx, 3))
    d0, d1 = d[:, :-1], d[:, 1:]
    co[:, 1:-1, 0] = -d1 / (d0 * (d0 + d1))
    co[:, 1:-1, 1] = (d1 - d0) / (d0 * d1)
    co[:, 1:-1, 2] = d0 / (d1 * (d0 + d1))
    a, b = d[:, 0], d[:, 1]
    co[:, 0, 0], co[:, 0, 1], co[:, 0, 2] = -(2 * a + b) / (a * (a + b)), (a + b) / (a * b), -a / (b * (a + b))
    a, b = d[:, -2], d[:, -1]
    co[:, -1, 0], co[:, -1, 1], co[:, -1, 2] = b / (a * (a + b)), -(a + b) / (a * b), (a + 2 * b) / (b * (a + b))
    return co


def great_circle_arc(phi1, lam1, phi2, lam2, radius):
    """Geodesic distance on a sphere (what pyproj's Geod.inv returns for an ellipsoid with a = b), in the numerically safe
    atan2 form of the central angle."""
    dl = lam2 - lam1
    y = np.hypot(np.cos(phi2) * np.sin(dl), np.cos(phi1) * np.sin(phi2) - np.sin(phi1) * np.cos(phi2) * np.cos(dl))
    x = np.sin(phi1) * np.sin(phi2) + np.cos(phi1) * np.cos(phi2) * np.cos(dl)
    return radius * np.arctan2(y, x)


def vorticity_tables(lat_deg, lon_deg, formulation: str = "metpy_no_crs"):
    """(xcoef [ny][nx][3], ycoef [ny][3], curv [ny]) of struct lec_diag_args for one of FORMULATIONS (module docstring)."""
    if formulation not in FORMULATIONS:
        raise ValueError(f"vorticity formulation must be one of {FORMULATIONS}, not {formulation!r}")
    phi = np.deg2rad(np.asarray(lat_deg, dtype=np.float64))
    lam = np.deg2rad(np.asarray(lon_deg, dtype=np.float64))
    ny, nx = phi.size, lam.size
    if formulation == "spherical":
        from .constants import RE
        xc = _three_point(lam)[None, :, 1:] / (RE * np.cos(phi))[:, None, None]
        yc = _three_point(phi)[:, 1:] / RE
        curv = np.tan(phi) / RE
    else:
        a = PYPROJ_SPHERE_RADIUS
        dx = great_circle_arc(phi[:, None], lam[None, :-1], phi[:, None], lam[None, 1:], a)        # [ny][nx - 1], along each row
        xc = _three_point_rows(dx)
        dy = great_circle_arc(phi[:-1], 0.0, phi[1:], 0.0, a)                                         # [ny - 1]: the same for every column
        yc = _three_point_rows(dy[None, :])[0]
        curv = np.zeros(ny)
    return np.ascontiguousarray(xc), np.ascontiguousarray(yc), np.ascontiguousarray(curv)


def box_ranges(lat_deg, lon_deg, limits) -> tuple:
    """(iw, ie, js, jn, jc, ic): the inclusive index ranges of the label slices .sel(lat=slice(min_lat, max_lat), lon=slice(min_lon,
    max_lon)) of get_position (lec_moving_framework.py:300-310) and the grid point nearest the box centre (used with -z)."""
    lat, lon = np.asarray(lat_deg), np.asarray(lon_deg)
    jj = np.flatnonzero((lat >= limits["min_lat"]) & (lat <= limits["max_lat"]))
    ii = np.flatnonzero((lon >= limits["min_lon"]) & (lon <= limits["max_lon"]))
    if jj.size == 0 or ii.size == 0:
        raise ValueError(f"box {limits} holds no grid point of the data")
    jc = int(np.argmin(np.abs(lat - limits["central_lat"])))
    ic = int(np.argmin(np.abs(lon - limits["central_lon"])))
    return int(ii[0]), int(ii[-1]), int(jj[0]), int(jj[-1]), jc, ic


def device_extrema(u850, v850, hgt850, lat_deg, lon_deg, limits_per_step, device="cuda:0", formulation="metpy_no_crs"):
    """``lec_track_diag`` on [time, lat, lon] slices (host arrays or device tensors): returns (val [nt][5], pos [nt][8]) as NumPy
    arrays -- zeta minimum, zeta maximum, height minimum, wind maximum, zeta at the box centre; (j, i) of the four extrema."""
    import torch
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.LecLibraryError("the track diagnostics run on the GPU: there is no CPU path")
    up = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a))).to(device=dev, dtype=torch.float64).contiguous()
    u, v, h = up(u850), up(v850), up(hgt850)
    if u.dim() != 3 or v.shape != u.shape or h.shape != u.shape:
        raise ValueError("u, v and height must be [time, lat, lon] slices of one shape")
    nt, ny, nx = (int(x) for x in u.shape)
    if (ny, nx) != (np.asarray(lat_deg).size, np.asarray(lon_deg).size) or len(limits_per_step) != nt:
        raise ValueError("slices, coordinates and boxes do not match")
    xcoef, ycoef, curv = vorticity_tables(lat_deg, lon_deg, formulation)
    box = np.array([box_ranges(lat_deg, lon_deg, lim) for lim in limits_per_step], dtype=np.int32).reshape(nt, 6)
    box_d, xc_d, yc_d, cv_d = torch.as_tensor(box).to(dev), torch.as_tensor(xcoef).to(dev), torch.as_tensor(ycoef).to(dev), torch.as_tensor(curv).to(dev)
    val = torch.empty((nt, 5), dtype=torch.float64, device=dev)
    pos = torch.empty((nt, 8), dtype=torch.int32, device=dev)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    args = _lib.DiagArgs(u_d=ptr(u), v_d=ptr(v), hgt_d=ptr(h), nt=nt, ny=ny, nx=nx, reserved0=0, box_d=ptr(box_d), xcoef_d=ptr(xc_d),
                         ycoef_d=ptr(yc_d), curv_d=ptr(cv_d), val_d=ptr(val), pos_d=ptr(pos),
                         stream=C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    with torch.cuda.device(dev):
        _lib.check(lib.lec_track_diag(C.byref(args)), "lec_track_diag")
    return val.cpu().numpy(), pos.cpu().numpy()


def positions(val, pos, lat_deg, lon_deg, limits, track_row=None, use_track_zeta=False) -> dict:
    """get_position for one time step from the device's extrema (``val`` [5], ``pos`` [8] of ``device_extrema``).  Values present
    (and not NaN) in the track row take precedence, as in the reference; with -z the vorticity is read at the box centre."""
    lat, lon = np.asarray(lat_deg), np.asarray(lon_deg)
    south = limits["min_lat"] < 0                              # hemisphere rule of the VALUE (lec_moving_framework.py:330-341)
    have = lambda name: track_row is not None and name in track_row.index and not np.isnan(float(track_row[name]))
    if track_row is not None and "min_max_zeta_850" in track_row.index:
        zval = float(track_row["min_max_zeta_850"])          # the column's value as it is, NaN included (lec_moving_framework.py:312-313)
    elif use_track_zeta and track_row is not None:
        zval = float(val[4])
    else:
        zval = float(val[0] if south else val[1])
    # xarray's .min() / .max() skip NaN (below-ground points at 850 hPa): so do the kernel's extrema, values and positions alike
    hval = float(track_row["min_hgt_850"]) if have("min_hgt_850") else float(val[2])
    wval = float(track_row["max_wind_850"]) if have("max_wind_850") else float(val[3])

    def where(q):
        j, i = int(pos[2 * q]), int(pos[2 * q + 1])
        return (float("nan"), float("nan")) if j < 0 else (float(lat[j]), float(lon[i]))

    js = box_ranges(lat, lon, limits)[2]
    zlat, zlon = where(0 if lat[js] < 0 else 1)                # hemisphere rule of the POSITION: the box's southernmost latitude (tools.py:95-128)
    hlat, hlon = where(2)
    wlat, wlon = where(3)
    return {
        "min_max_zeta_850_lat": zlat, "min_max_zeta_850_lon": zlon, "min_max_zeta_850": zval,
        "min_hgt_850_lat": hlat, "min_hgt_850_lon": hlon, "min_hgt_850": hval,
        "max_wind_850_lat": wlat, "max_wind_850_lon": wlon, "max_wind_850": wval,
    }


_POSITION_KEYS = ["min_max_zeta_850_lat", "min_max_zeta_850_lon", "min_max_zeta_850", "min_hgt_850_lat", "min_hgt_850_lon", "min_hgt_850",
                  "max_wind_850_lat", "max_wind_850_lon", "max_wind_850"]


def track_diagnostics(data, variable_list_df, limits_per_step, track=None, use_track_zeta=False, device="cuda:0", shard=None,
                      formulation="metpy_no_crs", slices=None):
    """All time steps: u, v, geopotential height at 85000 Pa -> list of position dicts.  ``shard`` (parallel.ShardContext): every
    rank evaluates its own time steps (each is independent), one gather of nine numbers per step brings them to rank 0; the other
    ranks get None.  ``slices``: {"u", "v", "geopt"} device tensors [own steps, lat, lon] at 85000 Pa that the streamed pass kept
    (``lec_streamed(keep_level=...)``): the file is then not read a second time."""
    import torch
    n_steps = len(limits_per_step)
    t0, t1 = (0, n_steps) if shard is None else shard.ranges(n_steps)[:2]
    k850 = int(np.flatnonzero(data.level == 85000.0)[0])
    name = lambda role: str(variable_list_df.loc[role]["Variable"])
    if slices is not None:
        key = {"Eastward Wind Component": "u", "Northward Wind Component": "v", "Geopotential": "geopt", "Geopotential Height": "geopt"}
        get = lambda role: slices[key[role]]
    elif hasattr(data, "level_slice"):        # streamed data set: three level slices decoded from the mapped file
        get = lambda role: data.level_slice(role, 85000.0, (t0, t1))
    else:
        h0 = (getattr(data, "t_held", None) or (0, n_steps))[0]          # the variables start at step h0 of the time axis
        get = lambda role: data.variables[name(role)][t0 - h0: t1 - h0, k850]
    u, v = get("Eastward Wind Component"), get("Northward Wind Component")
    f64 = lambda a: a.to(torch.float64) if isinstance(a, torch.Tensor) else a.astype(np.float64)
    if "Geopotential Height" in variable_list_df.index:
        hgt = f64(get("Geopotential Height"))
    else:
        hgt = f64(get("Geopotential")) / G                     # -> gpm
    val, pos = device_extrema(u, v, hgt, data.lat, data.lon, limits_per_step[t0:t1], device=device, formulation=formulation)
    out = []
    for t in range(t0, t1):
        row = None
        if track is not None:
            row = track.iloc[int(np.argmin(np.abs(track.index - data.time[t])))]
        out.append(positions(val[t - t0], pos[t - t0], data.lat, data.lon, limits_per_step[t], row, use_track_zeta))
    if shard is None:
        return out
    local = torch.tensor([[d[k] for k in _POSITION_KEYS] for d in out], dtype=torch.float64, device=device)
    full = shard.gather_rows(local, n_steps)
    if full is None:
        return None
    return [dict(zip(_POSITION_KEYS, (float(x) for x in row))) for row in full.cpu().numpy()]
