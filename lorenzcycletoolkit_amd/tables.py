"""Host-side coefficient tables consumed by the HIP kernels (layouts: include/lec_hip.h).

Everything here is O(nx + ny + nl + nt) NumPy fp64 work done once per call: trapezoid weights,
np.gradient coefficients for the box-local axes, the static-stability coefficients of the
diabatic-heating residual and the (level, lat) tables of stage 2.  The 4-D arithmetic itself
happens only in the kernels.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import numpy as np

from .constants import KAPPA, P0_PA, RE


def gradient_coefs(x: np.ndarray) -> np.ndarray:
    """(n, 3) coefficients a, b, c with d f/dx [i] = a f[i-1] + b f[i] + c f[i+1], equal to
    ``np.gradient(f, x, edge_order=1)`` (which DataArray.differentiate calls): second-order
    non-uniform interior, first-order one-sided ends.  Coefficients of missing neighbours are 0."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    if n < 2:
        raise ValueError("np.gradient needs at least 2 points along the axis")
    co = np.zeros((n, 3), dtype=np.float64)
    d = np.diff(x)
    if n > 2:
        dx1, dx2 = d[:-1], d[1:]
        co[1:-1, 0] = -dx2 / (dx1 * (dx1 + dx2))
        co[1:-1, 1] = (dx2 - dx1) / (dx1 * dx2)
        co[1:-1, 2] = dx1 / (dx2 * (dx1 + dx2))
    co[0, 1], co[0, 2] = -1.0 / d[0], 1.0 / d[0]
    co[-1, 0], co[-1, 1] = -1.0 / d[-1], 1.0 / d[-1]
    return co


def trapz_weights(x: np.ndarray) -> np.ndarray:
    """w with sum(w * f) == trapezoid integral of f over x (xarray ``integrate``)."""
    x = np.asarray(x, dtype=np.float64)
    w = np.zeros_like(x)
    d = np.diff(x)
    w[:-1] += 0.5 * d
    w[1:] += 0.5 * d
    return w


def is_uniform(x: np.ndarray, rtol: float = 1e-12) -> bool:
    d = np.diff(np.asarray(x, dtype=np.float64))
    return bool(np.all(np.abs(d - d[0]) <= rtol * np.abs(d[0])))


def nearest_index(coord: np.ndarray, value: float) -> int:
    """``data[indexer].sel({indexer: value}, method="nearest")`` (box_data.py:133-135 of the
    reference): ties go to the larger label on an increasing index."""
    d = np.abs(np.asarray(coord, dtype=np.float64) - float(value))
    return int(np.flatnonzero(d == d.min())[-1])


def box_indices(lat_deg, lon_deg, west, east, south, north):
    """Inclusive grid-index box (iw, ie, js, jn) from geographic limits (box_data.py:115-131)."""
    iw, ie = nearest_index(lon_deg, west), nearest_index(lon_deg, east)
    js, jn = nearest_index(lat_deg, south), nearest_index(lat_deg, north)
    if ie - iw < 1 or jn - js < 1:
        raise ValueError(f"box [{west}, {east}] x [{south}, {north}] selects fewer than 2 grid points along an axis")
    return iw, ie, js, jn


@dataclass
class BoxTables:
    box: np.ndarray       # int32 [n_box, 4]
    boxtab: np.ndarray    # [n_box, 4]
    wlon: np.ndarray      # [n_box, nxb_max]
    glon: np.ndarray      # [n_box, nxb_max, 3]
    lattab: np.ndarray    # [n_box, nyb_max, 4]
    boxtab2: np.ndarray   # [n_box, 4]
    lattab2: np.ndarray   # [n_box, nyb_max, 8]
    nxb_max: int
    nyb_max: int
    lon_uniform: bool


def build_box_tables(lat_deg: np.ndarray, lon_deg: np.ndarray, boxes: Sequence[Sequence[int]], nyb_min: int = 0) -> BoxTables:
    """Per-box tables for both stages.  ``boxes`` = inclusive index quadruples (iw, ie, js, jn).
    ``nyb_min``: pad the latitude extent of the tables (and of the row records) to at least this many rows -- chunks of one
    series of moving boxes share one record buffer, whose row count is the tallest box of the whole series."""
    lat = np.asarray(lat_deg, dtype=np.float64)
    lon = np.asarray(lon_deg, dtype=np.float64)
    box = np.asarray(boxes, dtype=np.int32).reshape(-1, 4)
    nb = box.shape[0]
    if np.any(box[:, 0] < 0) or np.any(box[:, 1] >= lon.size) or np.any(box[:, 2] < 0) or np.any(box[:, 3] >= lat.size):
        raise ValueError("box indices outside the grid")
    nxb = box[:, 1] - box[:, 0] + 1
    nyb = box[:, 3] - box[:, 2] + 1
    if np.any(nxb < 2) or np.any(nyb < 2):
        raise ValueError("every box needs at least 2 grid points along lat and lon")
    nxm, nym = int(nxb.max()), max(int(nyb.max()), int(nyb_min))
    if nym > lat.size:
        raise ValueError("nyb_min exceeds the grid")
    t = BoxTables(box=box, boxtab=np.zeros((nb, 4)), wlon=np.zeros((nb, nxm)), glon=np.zeros((nb, nxm, 3)),
                  lattab=np.zeros((nb, nym, 4)), boxtab2=np.zeros((nb, 4)), lattab2=np.zeros((nb, nym, 8)),
                  nxb_max=nxm, nyb_max=nym, lon_uniform=is_uniform(lon))
    # (lon_uniform selects the kernels' fast path and is decided on the grid's whole longitude axis, not on the boxes at hand: a
    # chunk or shard of a moving series may hold only boxes that happen to lie in an evenly spaced part of a stretched grid -- any
    # two-point-wide box does -- and must still use the formulation the whole series uses, or it differs from it by an ulp;
    # tests/soak_gpu.py found exactly that.)
    inv_dy = 1.0 / (np.deg2rad(1.0) * RE)      # dy = deg2rad(d lat/d lat) Re, thermodynamics.py:102
    cache = {}
    for b in range(nb):
        iw, ie, js, jn = (int(v) for v in box[b])
        key = (iw, ie, js, jn)
        if key in cache:                         # a stationary track repeats its box
            src = cache[key]
            for arr in (t.boxtab, t.wlon, t.glon, t.lattab, t.boxtab2, t.lattab2):
                arr[b] = arr[src]
            continue
        cache[key] = b
        lo, la = lon[iw:ie + 1], lat[js:jn + 1]
        rlon, rlat = np.deg2rad(lo), np.deg2rad(la)
        xlen = rlon[-1] - rlon[0]                                  # box_data.py:128
        ylen = np.sin(rlat[-1]) - np.sin(rlat[0])                  # box_data.py:129-131
        n = lo.size
        t.boxtab[b] = (1.0 / xlen, xlen / (n - 1), (n - 1) / (lo[-1] - lo[0]), 0.0)
        t.wlon[b, :n] = trapz_weights(rlon)
        t.glon[b, :n] = gradient_coefs(lo)
        m = la.size
        cosl = np.cos(rlat)
        t.lattab[b, :m, :3] = gradient_coefs(la) * inv_dy
        t.lattab[b, :m, 3] = 1.0 / (np.deg2rad(1.0) * cosl * RE)  # 1/dx, thermodynamics.py:101
        wphi = trapz_weights(rlat)
        t.boxtab2[b] = (-1.0 / (RE * xlen * ylen), -1.0 / (RE * ylen), xlen, ylen)   # boundary_terms.py:122-123
        t.lattab2[b, :m, 0] = cosl * wphi / ylen
        t.lattab2[b, :m, 1] = wphi
        t.lattab2[b, :m, 2] = cosl
        t.lattab2[b, :m, 3] = np.tan(rlat)
        t.lattab2[b, :m, 4:7] = gradient_coefs(rlat)
        t.lattab2[b, m:, 2] = 1.0                                   # keeps padded rows finite
    return t


def level_tables(level_pa: np.ndarray):
    """levtab [nl, 3]: S = -(T/theta) d theta/dp = al T[k-1] + be T[k] + ga T[k+1] with
    theta = T (P0/p)^kappa (thermodynamics.py:107-117); levtab2 [nl, 4]: p and d/dp coefficients."""
    p = np.asarray(level_pa, dtype=np.float64)
    if p.size < 2 or np.any(np.diff(p) <= 0):
        raise ValueError("levels must be ascending pressures in Pa (at least 2)")
    gc = gradient_coefs(p)
    ex = (p / P0_PA) ** KAPPA
    levtab = np.zeros((p.size, 3))
    levtab[1:, 0] = -ex[1:] * gc[1:, 0] / ex[:-1]
    levtab[:, 1] = -gc[:, 1]
    levtab[:-1, 2] = -ex[:-1] * gc[:-1, 2] / ex[1:]
    levtab2 = np.concatenate([p[:, None], gc], axis=1)
    return levtab, levtab2


def time_coefs(time_s: np.ndarray) -> np.ndarray:
    """[nt, 3] np.gradient coefficients over the time axis in seconds (thermodynamics.py:109-110)."""
    return gradient_coefs(np.asarray(time_s, dtype=np.float64))


def budgets_and_residuals(scalars: dict, time_s: np.ndarray, residuals: bool = True) -> dict:
    """Budget (dX/dt by np.gradient with dt = t[1]-t[0]) and residual columns of the results CSV
    (calc_budget_and_residual.py:32-56,131-154 of the reference).  O(nt) host work on the gathered
    per-time scalars."""
    dt = float(time_s[1] - time_s[0])
    out = dict(scalars)
    for term in ("Az", "Ae", "Kz", "Ke"):
        out[f"∂{term}/∂t (finite diff.)"] = np.gradient(np.asarray(scalars[term], dtype=np.float64), dt)
    if residuals:
        out["RGz"] = out["∂Az/∂t (finite diff.)"] + out["Cz"] + out["Ca"] - out["BAz"]
        out["RKz"] = out["∂Kz/∂t (finite diff.)"] - out["Cz"] - out["Ck"] - out["BKz"]
        out["RGe"] = out["∂Ae/∂t (finite diff.)"] - out["Ca"] + out["Ce"] - out["BAe"]
        out["RKe"] = out["∂Ke/∂t (finite diff.)"] - out["Ce"] + out["Ck"] - out["BKe"]
    return out
