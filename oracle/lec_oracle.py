"""
CPU oracle for the Lorenz Energy Cycle hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the numerics of daniloceano/LorenzCycleToolkit
(reference v1.1.11).  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and there only as the checker / the timed CPU baseline.  The shipped engine
(``lorenzcycletoolkit_amd``) never imports anything from ``oracle/``.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this restatement against
the reference's committed sample outputs: ``tests/golden/Catarina_NCEP-R2_fixed/*.csv`` (results + ten tables, 36 steps) and every
cell of the twenty ``tests/golden/Reg1_{fixed,track}/*_lv_ISBL3.csv`` tables that the 5-level / 5-step ``testdata_NCEP-R2.nc``
reproduces (sigma, the Q stencil, Ca, Ck on a second data set; the MOVING framework's sigma / Q-with-supplied-dT/dt path) -- see
DESIGN.md "Oracle".  One committed table disagrees with v1.1.11 and is accounted for: the track sample's Ck (an older second piece,
``ck_term2_of_the_committed_track_sample``).  Restatement-only, nothing in the reference can pin them: BPhiZ / BPhiE and every
pressure-INTEGRATED value of the moving framework.  The reference itself
cannot be imported here (xarray / metpy / pint are not installed; ordinary
ModuleNotFoundError).

The restatement deliberately evaluates the *un-factored* 4-D formulas in the same
operation order as the reference so that (a) NumPy dtype promotion reproduces the
reference's float32 arithmetic when it is fed float32 inputs, and (b) it is an
independent check of the engine's factored row-statistics formulation.

Array layout everywhere: [time, level, lat, lon]; lat ascending (S->N), lon ascending,
level ascending in Pa (preprocessing.py:358-365 of the reference).

Each function cites the reference file:line it follows (paths relative to the
reference repository root).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

# --------------------------------------------------------------------------------------
# Constants: MetPy 1.6.2 values (metpy/constants/default.py), used at
# thermodynamics.py:21-22, conversion_terms.py:31, boundary_terms.py:31, energy_contents.py:31
# --------------------------------------------------------------------------------------
G = 9.80665                              # m s-2
RE = 6371008.7714                        # m
R_UNIV = 8.314462618                     # J mol-1 K-1
MD = 28.96546e-3                         # kg mol-1
RD = R_UNIV / MD                         # J kg-1 K-1
GAMMA_D = 1.4
CP_D = GAMMA_D * RD / (GAMMA_D - 1.0)    # J kg-1 K-1
KAPPA = RD / CP_D
P0_PA = 100000.0                         # 1000 hPa

LEVEL_TERMS = [
    "Az", "Ae", "Kz", "Ke", "Ge", "Gz", "Cz", "Cz_1", "Cz_2", "Ca", "Ca_1", "Ca_2",
    "Ce", "Ce_1", "Ce_2", "Ck", "Ck_1", "Ck_2", "Ck_3", "Ck_4", "Ck_5",
]  # lec_fixed_framework.py:172-194


# --------------------------------------------------------------------------------------
# xarray primitives restated
# --------------------------------------------------------------------------------------
def trapz(y: np.ndarray, x: np.ndarray, axis: int) -> np.ndarray:
    """DataArray.integrate(coord): trapezoid rule, NaN-propagating.

    Follows xarray==2024.2.0 ``duck_array_ops.trapz`` (dx * 0.5 * (y[1:] + y[:-1]), then a plain
    sum) which is what calc_averages.py:43,76 and every ``.integrate`` in src/analysis call.
    """
    y = np.asarray(y)
    x = np.asarray(x)
    if axis < 0:
        axis += y.ndim
    tail = (None,) * (y.ndim - axis - 1)
    dx = x[(slice(1, None),) + tail] - x[(slice(None, -1),) + tail]
    hi = (slice(None),) * axis + (slice(1, None),)
    lo = (slice(None),) * axis + (slice(None, -1),)
    integrand = dx * 0.5 * (y[hi] + y[lo])
    return np.sum(integrand, axis=axis)


def differentiate(y: np.ndarray, x: np.ndarray, axis: int) -> np.ndarray:
    """DataArray.differentiate(coord, edge_order=1) == np.gradient with coordinates
    (conversion_terms.py:114,120,199,205,217,226; thermodynamics.py:57,98,99,110,115)."""
    return np.gradient(y, x, axis=axis, edge_order=1)


def select_nearest(coord: np.ndarray, value: float) -> int:
    """``data[indexer].sel({indexer: value}, method="nearest")`` (box_data.py:133-135).

    pandas' nearest indexer on a monotonically increasing index resolves ties towards the
    larger label."""
    d = np.abs(np.asarray(coord, dtype=np.float64) - float(value))
    cand = np.flatnonzero(d == d.min())
    return int(cand[-1])


def interpolate_and_drop_nan_levels(f: np.ndarray, p: np.ndarray):
    """``_handle_nans`` (energy_contents.py:190-208 and its three copies) for an array whose axis 1 is
    level ([time, level] or [time, level, lat]): linear interpolation along level across interior gaps
    (no extrapolation), then levels that still hold a NaN anywhere (any time, any latitude) are
    dropped -- xarray's ``dropna(dim=level)``.  Returns (f_clean, p_clean)."""
    f = np.array(f, dtype=np.result_type(f.dtype, np.float32), copy=True)
    if not np.isnan(f).any():
        return f, p
    g = np.moveaxis(f, 1, -1)                    # view: level last
    rows = g.reshape(-1, g.shape[-1])
    for r in range(rows.shape[0]):
        row = rows[r]
        ok = ~np.isnan(row)
        if ok.any() and not ok.all():
            filled = np.interp(p, p[ok], row[ok], left=np.nan, right=np.nan)
            rows[r] = np.where(ok, row, filled)
    g = rows.reshape(g.shape)
    f = np.moveaxis(g, -1, 1)
    if np.isnan(f).any():
        keep = ~np.isnan(g).reshape(-1, g.shape[-1]).any(axis=0)
        f, p = np.compress(keep, f, axis=1), p[keep]
    return f, p


def handle_nans_table(f: np.ndarray, p: np.ndarray) -> np.ndarray:
    """What ``_handle_nans`` leaves of a [time, level] function, on the full level axis: repaired values where it
    interpolated, NaN where it dropped the level.  The reference saves its main per-level tables AFTER ``_handle_nans``
    (e.g. energy_contents.py:104-105, conversion_terms.py:131-132); a dropped level is simply missing from the row it
    appends there, which this fixed-width form shows as NaN."""
    fc, pc = interpolate_and_drop_nan_levels(f, p)
    if len(pc) == len(p):
        return fc
    out = np.full(np.shape(f), np.nan, dtype=fc.dtype)
    out[:, np.isin(p, pc)] = fc
    return out


# --------------------------------------------------------------------------------------
# Box pre-reduction  (box_data.py, calc_averages.py, thermodynamics.py)
# --------------------------------------------------------------------------------------
@dataclass
class Domain:
    """The dataset handed to BoxData: fields on [time, level, lat, lon] plus coordinates.

    ``lat``/``lon`` in degrees with the file's dtype (float32 for the NCEP samples: the
    reference then derives float32 radians / cosines, preprocessing.py:288-290); ``level`` in Pa
    (float64, preprocessing.py:301-314); ``time_s`` seconds (float64)."""
    tair: np.ndarray
    u: np.ndarray
    v: np.ndarray
    omega: np.ndarray
    geopt: np.ndarray          # geopotential m2 s-2 (already multiplied by g if the file holds height)
    lat: np.ndarray
    lon: np.ndarray
    level: np.ndarray
    time_s: np.ndarray


@dataclass
class Box:
    """Everything BoxData exposes to the four analysis classes (box_data.py:78-105)."""
    rlats: np.ndarray
    rlons: np.ndarray
    coslats: np.ndarray
    lat: np.ndarray
    lon: np.ndarray
    level: np.ndarray
    xlength: float
    ylength: float
    f: Dict[str, np.ndarray] = field(default_factory=dict)  # tair, tair_ZA, tair_AA, tair_ZE, tair_AE, ...
    sigma_AA: Optional[np.ndarray] = None
    idx: tuple = ()


def _c3(c):  # broadcast a lat vector over [t, k, j]
    return c[None, None, :]


def _c4(c):  # broadcast a lat vector over [t, k, j, i]
    return c[None, None, :, None]


def zonal_average(X, rlons, xlength):
    """CalcZonalAverage (calc_averages.py:25-43)."""
    return trapz(X, rlons, axis=-1) / xlength


def area_average(X, rlats, coslats, xlength=None, rlons=None):
    """CalcAreaAverage (calc_averages.py:46-78).  The ``ylength`` argument of the reference is
    ignored there and recomputed from the sines of the first/last latitude (:75)."""
    ZA = zonal_average(X, rlons, xlength) if xlength is not None else X
    ylength = np.sin(rlats[-1]) - np.sin(rlats[0])
    return trapz(ZA * _c3(coslats), rlats, axis=2) / ylength


def static_stability(tair, level, rlats, rlons, coslats, xlength, ylength):
    """StaticStability (thermodynamics.py:26-73): pointwise g T/Cp - (p g/Rd) dT/dp, zonal then
    cos-weighted meridional mean (with the BoxData ylength), clamped to >= 0.03 (NaN -> 0.03)."""
    first = G * tair / CP_D
    second = level * G / RD
    third = differentiate(tair, level, axis=1)
    function = first - (second[None, :, None, None] * third)
    sigma_ZA = trapz(function, rlons, axis=-1) / xlength
    sigma_AA = trapz(sigma_ZA * _c3(coslats), rlats, axis=2) / ylength
    return np.where(sigma_AA > 0.03, sigma_AA, 0.03)


def adiabatic_heating(tair, level, omega, u, v, lat, lon, coslats, time_s, dTdt=None):
    """AdiabaticHEating (thermodynamics.py:76-124): diabatic heating as the residual of the
    thermodynamic equation.  Horizontal derivatives are per *degree* inside the box with one-sided
    differences at its edges; dx = deg2rad(1) cos(phi) Re, dy = deg2rad(1) Re."""
    dTdlambda = differentiate(tair, lon, axis=3)
    dTdphi = differentiate(tair, lat, axis=2)
    dx = np.deg2rad(differentiate(lon, lon, axis=0))[None, :] * coslats[:, None] * RE   # [lat, lon]
    dy = np.deg2rad(differentiate(lat, lat, axis=0)) * RE                               # [lat]
    adv = -1 * ((u * dTdlambda / dx[None, None]) + (v * dTdphi / _c4(dy)))
    exner = (level / P0_PA) ** KAPPA
    theta = tair / exner[None, :, None, None]
    if dTdt is None:
        dTdt = differentiate(tair, time_s, axis=0)
    sigma = -1 * (tair / theta) * differentiate(theta, level, axis=1)
    res = dTdt - adv - (sigma * omega)
    return res * CP_D


def make_box(dom: Domain, west, east, south, north, dTdt=None, fixed=True) -> Box:
    """BoxData.__init__ (box_data.py:78-295)."""
    rlats_d = np.deg2rad(dom.lat)
    rlons_d = np.deg2rad(dom.lon)
    iw, ie = select_nearest(dom.lon, west), select_nearest(dom.lon, east)
    js, jn = select_nearest(dom.lat, south), select_nearest(dom.lat, north)
    xlength = rlons_d[ie] - rlons_d[iw]                                  # box_data.py:128
    ylength = np.sin(rlats_d[jn]) - np.sin(rlats_d[js])                  # box_data.py:129-131
    sl = (slice(None), slice(None), slice(js, jn + 1), slice(iw, ie + 1))
    lat, lon = dom.lat[js:jn + 1], dom.lon[iw:ie + 1]
    rlats, rlons = rlats_d[js:jn + 1], rlons_d[iw:ie + 1]
    coslats = np.cos(np.deg2rad(lat))
    b = Box(rlats=rlats, rlons=rlons, coslats=coslats, lat=lat, lon=lon, level=dom.level,
            xlength=xlength, ylength=ylength, idx=(iw, ie, js, jn))

    def add(name, X):
        ZA = zonal_average(X, rlons, xlength)
        AA = area_average(ZA, rlats, coslats)
        b.f[name] = X
        b.f[name + "_ZA"] = ZA
        b.f[name + "_AA"] = AA
        b.f[name + "_ZE"] = X - ZA[..., None]
        b.f[name + "_AE"] = ZA - AA[..., None]

    add("tair", dom.tair[sl])
    add("u", dom.u[sl])
    add("v", dom.v[sl])
    add("omega", dom.omega[sl])
    add("geopt", dom.geopt[sl])
    if dTdt is not None:
        dTdt = dTdt[sl]
    Q = adiabatic_heating(b.f["tair"], dom.level, b.f["omega"], b.f["u"], b.f["v"], lat, lon,
                          coslats, dom.time_s, dTdt=dTdt)
    add("Q", Q)
    b.sigma_AA = static_stability(b.f["tair"], dom.level, rlats, rlons, coslats, xlength, ylength)
    return b


# --------------------------------------------------------------------------------------
# The four analysis classes
# --------------------------------------------------------------------------------------
def _int_p(function, level):
    """``_handle_nans`` followed by ``function.integrate(level)``."""
    f, p = interpolate_and_drop_nan_levels(function, level)
    return trapz(f, p, axis=1)


def energy_contents(b: Box):
    """EnergyContents.calc_az/ae/kz/ke (energy_contents.py:99-165).  Returns (scalars, level tables)."""
    s = b.sigma_AA
    lv = {}
    lv["Az"] = area_average(b.f["tair_AE"] ** 2, b.rlats, b.coslats) / (2 * s)
    lv["Ae"] = area_average(b.f["tair_ZE"] ** 2, b.rlats, b.coslats, b.xlength, b.rlons) / (2 * s)
    lv["Kz"] = area_average(b.f["u_ZA"] ** 2 + b.f["v_ZA"] ** 2, b.rlats, b.coslats)
    lv["Ke"] = area_average(b.f["u_ZE"] ** 2 + b.f["v_ZE"] ** 2, b.rlats, b.coslats, b.xlength, b.rlons)
    out = {
        "Az": _int_p(lv["Az"], b.level),
        "Ae": _int_p(lv["Ae"], b.level),
        "Kz": _int_p(lv["Kz"], b.level) / (2 * G),
        "Ke": _int_p(lv["Ke"], b.level) / (2 * G),
    }
    for name in ("Az", "Ae", "Kz", "Ke"):        # saved after _handle_nans (energy_contents.py:104-105 ...)
        lv[name] = handle_nans_table(lv[name], b.level)
    return out, lv


def conversion_terms(b: Box):
    """ConversionTerms.calc_cz/ca/ck/ce (conversion_terms.py:103-245), including the source's
    Ck term 5 that multiplies by d[u]/dp (:225-229)."""
    s = b.sigma_AA
    s4 = s[:, :, None, None]
    f = b.f
    lv = {}
    aa = lambda X: area_average(X, b.rlats, b.coslats, b.xlength, b.rlons)

    # Ca (conversion_terms.py:103-139)
    dphi_tae = differentiate(f["tair_AE"] * _c3(b.coslats), b.rlats, axis=2)
    t1 = (f["v_ZE"] * f["tair_ZE"] * dphi_tae[..., None]) / (2 * RE * s4)
    lv["Ca_1"] = aa(t1)
    dp_tae = differentiate(f["tair_AE"], b.level, axis=1)
    t2 = (f["omega_ZE"] * f["tair_ZE"]) * dp_tae[..., None]
    lv["Ca_2"] = aa(t2) / s
    lv["Ca"] = -(lv["Ca_1"] + lv["Ca_2"])

    # Ce (conversion_terms.py:141-165)
    c1 = RD / (b.level * G)
    lv["Ce_1"] = c1
    lv["Ce_2"] = aa(f["omega_ZE"] * f["tair_ZE"])
    lv["Ce"] = -(c1[None, :] * lv["Ce_2"])

    # Cz (conversion_terms.py:167-191)
    lv["Cz_1"] = c1
    lv["Cz_2"] = area_average(f["omega_AE"] * f["tair_AE"], b.rlats, b.coslats)
    lv["Cz"] = -(c1[None, :] * lv["Cz_2"])

    # Ck (conversion_terms.py:193-245)
    tan_lats = np.tan(b.rlats)
    d1 = differentiate(f["u_ZA"] / _c3(b.coslats), b.rlats, axis=2)
    lv["Ck_1"] = aa((_c4(b.coslats) * f["u_ZE"] * f["v_ZE"] / RE) * d1[..., None])
    d2 = differentiate(f["v_ZA"], b.rlats, axis=2)
    lv["Ck_2"] = aa(((f["v_ZE"] ** 2) / RE) * d2[..., None])
    lv["Ck_3"] = aa((_c4(tan_lats) * (f["u_ZE"] ** 2) * f["v_ZA"][..., None]) / RE)
    d4 = differentiate(f["u_ZA"], b.level, axis=1)
    lv["Ck_4"] = aa(f["omega_ZE"] * f["u_ZE"] * d4[..., None])
    d5 = differentiate(f["u_ZA"], b.level, axis=1)          # sic: u_ZA, conversion_terms.py:225-227
    lv["Ck_5"] = aa(f["omega_ZE"] * f["v_ZE"] * d5[..., None])
    lv["Ck"] = lv["Ck_1"] + lv["Ck_2"] + lv["Ck_3"] + lv["Ck_4"] + lv["Ck_5"]

    out = {
        "Cz": _int_p(lv["Cz"], b.level),
        "Ca": _int_p(lv["Ca"], b.level),
        "Ck": _int_p(lv["Ck"], b.level) / G,
        "Ce": _int_p(lv["Ce"], b.level),
    }
    for name in ("Cz", "Ca", "Ck", "Ce"):        # the sums are saved after _handle_nans, the pieces before (conversion_terms.py:117-132 ...)
        lv[name] = handle_nans_table(lv[name], b.level)
    return out, lv


def ck_term2_of_the_committed_track_sample(b: Box):
    """NOT the current reference: the second piece of Ck as the revision that wrote
    ``samples/Reg1-Representative_NCEP-R2_track-15x15/Ck_lv_ISBL3.csv`` evaluated it -- ``{[v'^2 / Re * d([v] cos(phi))/d(phi)]}``,
    the meridional derivative taken of ``[v] cos(phi)`` where conversion_terms.py:204-208 (v1.1.11, restated in
    ``conversion_terms`` above) differentiates ``[v]`` alone.  Found in round 5 by fitting that table: with this one piece exchanged the
    committed track table is reproduced to its float32 print precision (2e-7), with the current piece it is off by 10-60 %, while
    the FIXED sample of the same data set (written later) carries the current form.  Used only by
    tests/test_oracle_golden.py to account for that table; nothing else may call it."""
    f = b.f
    d2 = differentiate(f["v_ZA"] * _c3(b.coslats), b.rlats, axis=2)
    return area_average(((f["v_ZE"] ** 2) / RE) * d2[..., None], b.rlats, b.coslats, b.xlength, b.rlons)


def boundary_terms(b: Box):
    """BoundaryTerms.calc_baz/bae/bkz/bke/boz/boe (boundary_terms.py:125-418)."""
    f = b.f
    s = b.sigma_AA
    s3 = s[:, :, None]
    s4 = s[:, :, None, None]
    c1 = -1 / (RE * b.xlength * b.ylength)          # boundary_terms.py:122
    c2 = -1 / (RE * b.ylength)                      # boundary_terms.py:123
    za = lambda X: zonal_average(X, b.rlons, b.xlength)
    ew = lambda X: X[..., -1] - X[..., 0]           # .sel(lon=east) - .sel(lon=west)
    ns = lambda X: X[:, :, -1] - X[:, :, 0]         # .sel(lat=north) - .sel(lat=south)
    bt = lambda X: X[:, -1] - X[:, 0]               # .isel(level=-1) - .isel(level=0)
    tae4 = f["tair_AE"][..., None]
    out = {}

    # BAz (boundary_terms.py:125-183)
    t1 = ((2 * tae4 * f["tair_ZE"] * f["u"]) + (tae4 ** 2 * f["u"])) / (2 * s4)
    t1 = trapz(ew(t1), b.rlats, axis=2)
    t1 = _int_p(t1, b.level) * c1
    t2 = za(f["v_ZE"] * f["tair_ZE"]) * 2 * f["tair_AE"]
    t2 = (t2 + ((f["tair_AE"] ** 2) * f["v_ZA"])) * _c3(b.coslats)
    t2 = ns(t2) / (2 * s)
    t2 = _int_p(t2, b.level) * c2
    t3 = za(2 * f["omega_ZE"] * f["tair_ZE"]) * f["tair_AE"] + f["omega_ZA"] * f["tair_AE"] ** 2
    t3, p3 = interpolate_and_drop_nan_levels(t3, b.level)                  # on [time, level, lat] (boundary_terms.py:169)
    keep3 = np.isin(b.level, p3)
    t3 = area_average(t3, b.rlats, b.coslats) / (2 * s[:, keep3])
    out["BAz"] = t1 + t2 - bt(t3)

    # BAe (boundary_terms.py:185-232)
    t1 = ew(f["u"] * (f["tair_ZE"] ** 2))
    t1 = trapz(t1 / (2 * s3), b.rlats, axis=2)
    t1 = _int_p(t1, b.level) * c1
    t2 = za(f["v"] * f["tair_ZE"] ** 2) * _c3(b.coslats)
    t2 = ns(t2 / (2 * s3))
    t2 = _int_p(t2, b.level) * c2
    t3 = (f["omega"] * f["tair_ZE"] ** 2) / (2 * s4)
    t3 = area_average(t3, b.rlats, b.coslats, b.xlength, b.rlons)
    t3, _ = interpolate_and_drop_nan_levels(t3, b.level)
    out["BAe"] = t1 + t2 - bt(t3)

    # BKz / BKe (boundary_terms.py:234-326)
    for name, K in (("BKz", f["u"] ** 2 + f["v"] ** 2 - f["u_ZE"] ** 2 - f["v_ZE"] ** 2),
                    ("BKe", f["u_ZE"] ** 2 + f["v_ZE"] ** 2)):
        t1 = ew(f["u"] * K)
        t1 = trapz(t1 / (2 * G), b.rlats, axis=2)
        t1 = _int_p(t1, b.level) * c1
        t2 = ns(za(K * f["v"] * _c4(b.coslats)))
        t2 = _int_p(t2 / (2 * G), b.level) * c2
        t3 = area_average(K * f["omega"], b.rlats, b.coslats, b.xlength, b.rlons) / (2 * G)
        t3, _ = interpolate_and_drop_nan_levels(t3, b.level)
        out[name] = t1 + t2 - bt(t3)

    # BΦZ (boundary_terms.py:328-366): no east-west difference in the first term
    t1 = trapz((f["v_ZA"] * f["geopt_AE"]) / G, b.rlats, axis=2)
    t1 = _int_p(t1, b.level) * c1
    t2 = ns((f["v_ZA"] * f["geopt_AE"]) * _c3(b.coslats) / G)
    t2 = _int_p(t2, b.level) * c2
    t3 = area_average(f["omega_AE"] * f["geopt_AE"], b.rlats, b.coslats) / G
    t3, _ = interpolate_and_drop_nan_levels(t3, b.level)
    out["BΦZ"] = t1 + t2 - bt(t3)

    # BΦE (boundary_terms.py:368-418): second term built from zonal means (:390)
    t1 = ew((f["v_ZE"] * f["geopt_AE"][..., None]) / G)
    t1 = trapz(t1, b.rlats, axis=2)
    t1 = _int_p(t1, b.level) * c1
    t2 = ns((f["v_ZA"] * f["geopt_AE"]) * _c3(b.coslats) / G)
    t2 = _int_p(t2, b.level) * c2
    t3 = area_average(f["omega_ZE"] * f["geopt_ZE"], b.rlats, b.coslats, b.xlength, b.rlons) / G
    t3, _ = interpolate_and_drop_nan_levels(t3, b.level)
    out["BΦE"] = t1 + t2 - bt(t3)
    return out


def generation_terms(b: Box):
    """GenerationDissipationTerms.calc_gz/ge (generation_and_dissipation_terms.py:122-152)."""
    f = b.f
    s = b.sigma_AA
    lv = {}
    lv["Gz"] = area_average(f["Q_AE"] * f["tair_AE"], b.rlats, b.coslats) / (CP_D * s)
    lv["Ge"] = area_average(f["Q_ZE"] * f["tair_ZE"], b.rlats, b.coslats, b.xlength, b.rlons) / (CP_D * s)
    out = {"Gz": _int_p(lv["Gz"], b.level), "Ge": _int_p(lv["Ge"], b.level)}
    for name in ("Gz", "Ge"):                    # generation_and_dissipation_terms.py:127-128,144-145
        lv[name] = handle_nans_table(lv[name], b.level)
    return out, lv


def all_terms(b: Box):
    """The sequence of calc_* calls made by lec_fixed (lec_fixed_framework.py:215-271) and
    compute_and_store_terms (lec_moving_framework.py:430-495)."""
    e, le = energy_contents(b)
    c, lc = conversion_terms(b)
    bd = boundary_terms(b)
    g, lg = generation_terms(b)
    scalars = {**e, **c, **bd, **g}
    levels = {**le, **lc, **lg}
    return scalars, levels


# --------------------------------------------------------------------------------------
# Budgets and residuals (calc_budget_and_residual.py:32-56,131-154)
# --------------------------------------------------------------------------------------
def budgets_and_residuals(scalars: Dict[str, np.ndarray], time_s: np.ndarray, residuals=True):
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


# --------------------------------------------------------------------------------------
# Frameworks
# --------------------------------------------------------------------------------------
def lec_fixed(dom: Domain, west, east, south, north):
    """lec_fixed (lec_fixed_framework.py:199-293): one BoxData over the whole cube, all terms,
    budgets + residuals.  Returns (scalars incl. budgets/residuals, per-level tables)."""
    b = make_box(dom, west, east, south, north, fixed=True)
    scalars, levels = all_terms(b)
    full = budgets_and_residuals(scalars, dom.time_s, residuals=True)
    return full, levels


def moving_dTdt(dom: Domain):
    """run_lec_analysis (lorenzcycletoolkit.py:184-186): dT/dt over the (track-selected) times
    of the pre-cropped dataset, np.gradient in seconds."""
    return differentiate(dom.tair, dom.time_s, axis=0)


def lec_moving(dom: Domain, boxes, residuals=True, per_box=None):
    """lec_moving (lec_moving_framework.py:639-745): one BoxData per time step with that step's
    box (west, east, south, north) and the precomputed dT/dt slice.  ``per_box`` (tests only): a callable
    Box -> {name: [1, level] array} whose tables are collected next to the reference's own."""
    dTdt = moving_dTdt(dom)
    nt = dom.tair.shape[0]
    acc: Dict[str, list] = {}
    lacc: Dict[str, list] = {}
    for t in range(nt):
        sub = Domain(dom.tair[t:t + 1], dom.u[t:t + 1], dom.v[t:t + 1], dom.omega[t:t + 1],
                     dom.geopt[t:t + 1], dom.lat, dom.lon, dom.level, dom.time_s[t:t + 1])
        w, e, s, n = boxes[t]
        b = make_box(sub, w, e, s, n, dTdt=dTdt[t:t + 1], fixed=False)
        sc, lv = all_terms(b)
        if per_box is not None:
            lv = {**lv, **per_box(b)}
        for k, val in sc.items():
            acc.setdefault(k, []).append(np.asarray(val).reshape(-1)[0])
        for k, val in lv.items():
            lacc.setdefault(k, []).append(np.asarray(val).reshape(-1, len(dom.level))[0]
                                          if np.asarray(val).ndim > 1 else np.asarray(val))
    scalars = {k: np.array(v) for k, v in acc.items()}
    levels = {k: np.array(v) for k, v in lacc.items()}
    full = budgets_and_residuals(scalars, dom.time_s, residuals=residuals)
    return full, levels


# --------------------------------------------------------------------------------------
# Loading the NetCDF-3 samples the way prepare_data does (preprocessing.py:149-371)
# --------------------------------------------------------------------------------------
def load_ncep_sample(path: str, dtype=None) -> Domain:
    """Reads a NetCDF-3 NCEP-R2 sample with the NCEP-R2 namelist roles (inputs/namelist_NCEP-R2),
    applies the lon wrap (tools.py:76-92), radians, level->Pa, the three sorts and the <10 hPa
    drop of process_data.  ``dtype=None`` keeps the file dtype (float32) like the reference;
    ``np.float64`` upcasts fields *and* lat/lon first (a clean fp64 evaluation)."""
    from scipy.io import netcdf_file

    nc = netcdf_file(path, mmap=False)
    get = lambda n: np.array(nc.variables[n].data).astype(nc.variables[n].data.dtype.newbyteorder("="))
    lat, lon = get("lat_2"), get("lon_2")
    lev = get("lv_ISBL3")
    hours = get("initial_time0_hours")
    dims = nc.variables["TMP_2_ISBL"].dimensions
    fields = {}
    for role, name in (("tair", "TMP_2_ISBL"), ("u", "U_GRD_2_ISBL"), ("v", "V_GRD_2_ISBL"),
                       ("omega", "V_VEL_2_ISBL"), ("hgt", "HGT_2_ISBL")):
        a = get(name)
        order = [dims.index(d) for d in ("initial_time0_hours", "lv_ISBL3", "lat_2", "lon_2")]
        fields[role] = np.transpose(a, order)
    nc.close()
    if dtype is not None:
        lat, lon = lat.astype(dtype), lon.astype(dtype)
        fields = {k: v.astype(dtype) for k, v in fields.items()}
    if lon.min() < -180 or lon.max() > 180:
        lon = (lon + 180) % 360 - 180
    level = lev.astype(np.float64) * 100.0            # hPa -> Pa
    io = np.argsort(lon, kind="stable")
    ik = np.argsort(level, kind="stable")
    ij = np.argsort(lat, kind="stable")
    lon, level, lat = lon[io], level[ik], lat[ij]
    for k in fields:
        fields[k] = np.ascontiguousarray(fields[k][:, :, :, io][:, ik][:, :, ij])
    keep = level >= 1000.0                              # preprocessing.py:364-365
    level = level[keep]
    for k in fields:
        fields[k] = np.ascontiguousarray(fields[k][:, keep])
    time_s = (hours - hours.min()) * 3600.0
    geopt = fields["hgt"] * G                           # box_data.py:233-241
    return Domain(fields["tair"], fields["u"], fields["v"], fields["omega"], geopt,
                  lat, lon, level, time_s.astype(np.float64))


def crop_domain(dom: Domain, west, east, south, north) -> Domain:
    """slice_domain, fixed branch (select_area.py:272-295,331-336): nearest-point inclusive crop."""
    iw, ie = select_nearest(dom.lon, west), select_nearest(dom.lon, east)
    js, jn = select_nearest(dom.lat, south), select_nearest(dom.lat, north)
    sl = (slice(None), slice(None), slice(js, jn + 1), slice(iw, ie + 1))
    return Domain(dom.tair[sl], dom.u[sl], dom.v[sl], dom.omega[sl], dom.geopt[sl],
                  dom.lat[js:jn + 1], dom.lon[iw:ie + 1], dom.level, dom.time_s)


def crop_domain_track(dom: Domain, track_lat, track_lon, max_width=15, max_length=15) -> Domain:
    """slice_domain, track branch (select_area.py:297-313,331-336): label slice of the track
    extent +- (half box + one grid step)."""
    dx = dom.lon[1] - dom.lon[0]
    dy = dom.lat[1] - dom.lat[0]
    w = np.min(track_lon) - max_width / 2 - dx
    e = np.max(track_lon) + max_width / 2 + dx
    s = np.min(track_lat) - max_length / 2 - dy
    n = np.max(track_lat) + max_length / 2 + dy
    ii = np.flatnonzero((dom.lon >= w) & (dom.lon <= e))
    jj = np.flatnonzero((dom.lat >= s) & (dom.lat <= n))
    sl = (slice(None), slice(None), slice(jj[0], jj[-1] + 1), slice(ii[0], ii[-1] + 1))
    return Domain(dom.tair[sl], dom.u[sl], dom.v[sl], dom.omega[sl], dom.geopt[sl],
                  dom.lat[jj[0]:jj[-1] + 1], dom.lon[ii[0]:ii[-1] + 1], dom.level, dom.time_s)
