"""Shared helpers of the parity tests: synthetic domains and engine-vs-oracle comparison."""
from __future__ import annotations

import numpy as np

from oracle import lec_oracle as o

SCALARS = ["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "BΦZ", "BΦE", "Gz", "Ge"]


def synthetic_domain(nt, nl, ny, nx, seed=0, dtype=np.float64, lat0=-60.0, lat1=-10.0, lon0=-80.0, lon1=-20.0,
                     nonuniform_lon=False, dt_s=21600.0, pmin=10000.0):
    """Smooth + noisy fields in the spirit of SURVEY.md section 8(d), on a regional grid."""
    rng = np.random.default_rng(seed)
    lat = np.linspace(lat0, lat1, ny)
    lon = np.linspace(lon0, lon1, nx)
    if nonuniform_lon:
        lon = lon + 0.3 * (lon1 - lon0) / nx * np.sin(np.linspace(0, 7, nx))
        lon = np.sort(lon)
    level = np.linspace(pmin, 100000.0, nl)
    time_s = np.arange(nt) * dt_s
    phi, lam = np.deg2rad(lat)[None, None, :, None], np.deg2rad(lon)[None, None, None, :]
    p = level[None, :, None, None]
    tt = (np.arange(nt) / max(nt, 1))[:, None, None, None]
    shp = (nt, nl, ny, nx)
    T = 288.0 * (p / 1e5) ** 0.19 + 10.0 * np.cos(2 * phi) * (p / 1e5) + 2.0 * np.sin(3 * lam + tt) + rng.standard_normal(shp)
    u = 25.0 * np.cos(phi) * (1 - p / 1.2e5) + 5.0 * rng.standard_normal(shp)
    v = 3.0 * np.sin(2 * lam) * np.cos(phi) + 3.0 * rng.standard_normal(shp)
    w = 0.05 * np.sin(3 * lam) * np.cos(phi) + 0.1 * rng.standard_normal(shp)
    ph = o.G * 7000.0 * np.log(1e5 / p) + 100.0 * rng.standard_normal(shp)
    f = [np.ascontiguousarray(a.astype(dtype)) for a in (T, u, v, w, ph)]
    return o.Domain(f[0], f[1], f[2], f[3], f[4], lat.astype(np.float64), lon.astype(np.float64), level, time_s)


def as_f64(dom: o.Domain) -> o.Domain:
    """The same values in float64 (what the engine computes from float32 storage)."""
    c = lambda a: np.ascontiguousarray(a.astype(np.float64))
    return o.Domain(c(dom.tair), c(dom.u), c(dom.v), c(dom.omega), c(dom.geopt), c(dom.lat), c(dom.lon),
                    dom.level, dom.time_s)


def scale_err(a, r):
    """max |a - r| relative to the scale of r; NaNs must sit at the same places (else inf)."""
    a = np.asarray(a, dtype=np.float64)
    r = np.asarray(r, dtype=np.float64)
    na, nr = np.isnan(a), np.isnan(r)
    if na.any() or nr.any():
        if not np.array_equal(na, nr):
            return float("inf")
        a, r = a[~na], r[~nr]
        if a.size == 0:
            return 0.0
    den = np.max(np.abs(r))
    return float(np.max(np.abs(a - r)) / den) if den > 0 else float(np.max(np.abs(a - r)))


BUDGETS = {"∂Az/∂t (finite diff.)": ("Az",), "∂Ae/∂t (finite diff.)": ("Ae",), "∂Kz/∂t (finite diff.)": ("Kz",), "∂Ke/∂t (finite diff.)": ("Ke",)}
RESIDUALS = {"RGz": ("Az", "Cz", "Ca", "BAz"), "RKz": ("Kz", "Cz", "Ck", "BKz"), "RGe": ("Ae", "Ca", "Ce", "BAe"), "RKe": ("Ke", "Ce", "Ck", "BKe")}


def compare(engine_scalars, engine_levels, ref_scalars, ref_levels, tol, what="", *, time_s):
    """Every integrated term and level table within `tol` of the oracle, relative to the term's scale -- and the budget / residual
    columns of the results CSV (calc_budget_and_residual.py:32-56,131-154): the product's host code (tables.budgets_and_residuals) on
    the engine's series over `time_s` against the oracle's columns.  A budget is a difference of neighbouring energies over dt, so its
    error is judged against max|X| / dt (the scale of what is subtracted), a residual's against the largest of its operands' scales."""
    from lorenzcycletoolkit_amd import tables
    worst = {}
    for name in SCALARS:
        if name in ref_scalars:
            worst[name] = scale_err(engine_scalars[name], ref_scalars[name])
    if len(time_s) >= 2:
        has_res = "RGz" in ref_scalars
        full = tables.budgets_and_residuals({k: np.asarray(v, dtype=np.float64) for k, v in engine_scalars.items()}, np.asarray(time_s), residuals=has_res)
        dt = float(time_s[1] - time_s[0])
        top = lambda n: float(np.nanmax(np.abs(np.asarray(ref_scalars[n], dtype=np.float64)))) if np.isfinite(np.asarray(ref_scalars[n], dtype=np.float64)).any() else 1.0
        for col, ops in {**BUDGETS, **(RESIDUALS if has_res else {})}.items():
            a, r = np.asarray(full[col], dtype=np.float64), np.asarray(ref_scalars[col], dtype=np.float64)
            if not np.array_equal(np.isnan(a), np.isnan(r)):
                worst["budget:" + col] = float("inf")
                continue
            ok = ~np.isnan(r)
            scale = max([top(ops[0]) / dt] + [top(n) for n in ops[1:]])
            worst["budget:" + col] = float(np.max(np.abs(a[ok] - r[ok])) / scale) if ok.any() else 0.0
    for name, ref in ref_levels.items():
        ref = np.asarray(ref, dtype=np.float64)
        got = engine_levels[name]
        if ref.ndim == 1:                         # Cz_1 / Ce_1: level-only
            ref = np.broadcast_to(ref, got.shape)
        worst["lv:" + name] = scale_err(got, ref)
    bad = {k: v for k, v in worst.items() if not (v <= tol)}
    assert not bad, f"{what}: terms beyond tol={tol:g}: {bad}"
    return worst


# ---------------------------------------------------------------------------------------------
# The reference's committed Reg1 sample tables (tests/golden/Reg1_{fixed,track}/*_lv_ISBL3.csv) against testdata_NCEP-R2.nc
# ---------------------------------------------------------------------------------------------
# testdata_NCEP-R2.nc is a 5-step, 5-level (600...1000 hPa) subset of the file those samples were computed from.  Its five
# levels are CONTIGUOUS in the 17-level original and 1000 hPa is the bottom of both, so np.gradient over level gives the same
# numbers at 700, 850, 925 and 1000 hPa (600 hPa is an interior level there and an edge here); np.gradient over time gives the
# same numbers at steps 1-4 (step 5 is an interior step there and the last one here).  Which cells of which table the subset
# reproduces therefore depends on what the term differentiates:
#   Kz Ke Ce Cz        no d/dp, no sigma, no d/dt         -> every column, every row
#   Az Ae Ca Ck        sigma / d(T*)/dp / d[u]/dp         -> 700-1000 hPa
#   Ge Gz              + dT/dt inside Q                   -> 700-1000 hPa, rows 1-4 (the track tables hold 2 rows anyway)
# sign: the committed Cz / Ca tables were written by a revision with the opposite sign (tests/golden/README.md).
REG1_BOX = (-60, -30, -42.5, -17.5)      # tests/golden/inputs/box_limits_Reg1
REG1_TERMS = {   # term: (skip the 600-hPa column, usable rows of the fixed table, sign)
    "Kz": (False, 5, 1), "Ke": (False, 5, 1), "Ce": (False, 5, 1), "Cz": (False, 5, -1),
    "Az": (True, 5, 1), "Ae": (True, 5, 1), "Ca": (True, 5, -1), "Ck": (True, 5, 1),
    "Ge": (True, 4, 1), "Gz": (True, 4, 1),
}


def reg1_table(golden_dir, kind, term):
    """(reference cells, level slice into the 5-level axis of testdata, number of rows, sign) for `kind` in fixed / track / choose."""
    import os
    import pandas as pd
    skip600, rows_fixed, sign = REG1_TERMS[term]
    hpa = [600.0, 700.0, 850.0, 925.0, 1000.0][1 if skip600 else 0:]
    df = pd.read_csv(os.path.join(golden_dir, f"Reg1_{kind}", f"{term}_lv_ISBL3.csv"), index_col=0)
    cols = [str(h) if kind == "fixed" else str(h * 100.0) for h in hpa]        # hPa headers (old fixed sample) / Pa (track, choose)
    r = df[cols].values
    rows = min(rows_fixed, len(r)) if kind == "fixed" else len(r)                # the track tables hold 3 (Ge, Gz: 2) rows, the choose tables 3
    return r[:rows], slice(1 if skip600 else 0, None), rows, sign


# The boxes of the reference's `-c` (interactive chooser) sample, samples/Reg1-Representative_NCEP-R2_choose/: not recorded anywhere in the
# reference -- recovered in round 5 by searching every box of the grid for the one whose Kz table (no derivative in it) equals the sample's
# row: one box per step matches at 1.4e-7 (the table's float32 print precision), the next best is 6-12 % off.  (west, east, south, north)
REG1_CHOOSE_BOXES = [(-65.0, -50.0, -30.0, -20.0), (-67.5, -50.0, -30.0, -20.0), (-62.5, -42.5, -37.5, -25.0)]


def reg1_track_limits(golden_dir):
    import os
    import pandas as pd
    tr = pd.read_csv(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), sep=";")
    return tr, [(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(tr.Lat, tr.Lon)]


def write_packed_era5_style(path, nt=7, fill=True, offset=True):
    """ERA5-style file: int16 with scale_factor / add_offset / _FillValue, lon 0..357.5, lat N -> S, levels in hPa
    from the surface up including 5 hPa (dropped by the >= 10 hPa filter).  ``fill`` / ``offset``: leave out the fill value
    or the add_offset -- the reference's xarray 2024.2.0 decodes the three variants to float32 / float64 / float32."""
    from scipy.io import netcdf_file
    rng = np.random.default_rng(4)
    lon = np.arange(0.0, 360.0, 2.5)
    lat = np.arange(60.0, -62.5, -2.5)
    lev = np.array([1000, 850, 700, 500, 300, 200, 100, 50, 5], dtype=np.int32)
    nl, ny, nx = lev.size, lat.size, lon.size
    f = netcdf_file(path, "w", version=2)
    for n, s in (("time", nt), ("level", nl), ("latitude", ny), ("longitude", nx)):
        f.createDimension(n, s)
    tv = f.createVariable("time", "i", ("time",)); tv[:] = 6 * np.arange(nt); tv.units = "hours since 2020-01-01 00:00:00"
    lv = f.createVariable("level", "i", ("level",)); lv[:] = lev; lv.units = "millibars"
    la = f.createVariable("latitude", "f", ("latitude",)); la[:] = lat
    lo = f.createVariable("longitude", "f", ("longitude",)); lo[:] = lon
    p = (lev[None, :, None, None] * 100.0) / 1e5
    fields = {
        "t": 288.0 * p ** 0.19 + 8.0 * np.cos(np.deg2rad(2 * lat))[None, None, :, None] * p + rng.standard_normal((nt, nl, ny, nx)),
        "u": 20.0 * np.cos(np.deg2rad(lat))[None, None, :, None] * (1 - p / 1.2) + 5 * rng.standard_normal((nt, nl, ny, nx)),
        "v": 3.0 * rng.standard_normal((nt, nl, ny, nx)),
        "w": 0.1 * rng.standard_normal((nt, nl, ny, nx)),
        "z": 9.80665 * 7000.0 * np.log(1.0 / p) + 100.0 * rng.standard_normal((nt, nl, ny, nx)),
    }
    for name, a in fields.items():
        lo_, hi_ = a.min(), a.max()
        off = 0.5 * (hi_ + lo_) if offset else 0.0
        scale = (hi_ - lo_) / 65000.0 if offset else max(abs(hi_), abs(lo_)) / 32000.0
        q = np.clip(np.round((a - off) / scale), -32000, 32000).astype(np.int16)
        if name == "v" and fill:
            q[2, 7, :, :] = -32767          # 50 hPa = the top kept level, one time step: dropped for the whole series
            q[4 % nt, 3, 10, 20] = -32767   # an interior point: that level is repaired by interpolation at that step
        v = f.createVariable(name, "h", ("time", "level", "latitude", "longitude"))
        v[:] = q
        v.scale_factor = float(scale)
        if offset:
            v.add_offset = float(off)
        if fill:
            v._FillValue = np.int16(-32767)
    f.close()
