"""850-hPa diagnostics (caller-side, parity unpinned): the vorticity stencil on analytic fields."""
import numpy as np
import pandas as pd

from lorenzcycletoolkit_amd import diagnostics as dg
from lorenzcycletoolkit_amd.constants import RE


def test_first_derivative_is_second_order_exact_on_quadratics():
    x = np.array([0.0, 0.5, 1.5, 2.0, 4.0])
    f = 3 * x ** 2 - 2 * x + 1
    assert np.allclose(dg.first_derivative(f, x, 0), 6 * x - 2, rtol=1e-13)


def test_solid_body_rotation_vorticity():
    """u = U cos(phi), v = 0  ->  zeta = 2 U sin(phi) / Re."""
    lat = np.linspace(-60, -10, 101)
    lon = np.linspace(-80, -20, 61)
    U = 30.0
    u = U * np.cos(np.deg2rad(lat))[:, None] * np.ones((1, lon.size))
    z = dg.vorticity(u, np.zeros_like(u), lat, lon)
    want = 2 * U * np.sin(np.deg2rad(lat))[:, None] / RE * np.ones((1, lon.size))
    assert np.max(np.abs(z - want)) < 2e-4 * np.max(np.abs(want))


def test_box_positions_pick_extrema_and_prefer_track_values():
    lat = np.linspace(-40, -20, 21)
    lon = np.linspace(-60, -40, 21)
    zeta = np.zeros((21, 21)); zeta[7, 9] = -5e-5
    hgt = np.full((21, 21), 1500.0); hgt[12, 3] = 1400.0
    w = np.ones((21, 21)); w[5, 15] = 33.0
    lim = {"min_lat": -38, "max_lat": -22, "min_lon": -58, "max_lon": -42, "central_lat": -30, "central_lon": -50}
    p = dg.box_positions(zeta, hgt, w, lat, lon, lim)
    assert (p["min_max_zeta_850_lat"], p["min_max_zeta_850_lon"], p["min_max_zeta_850"]) == (lat[7], lon[9], -5e-5)
    assert (p["min_hgt_850_lat"], p["min_hgt_850_lon"], p["min_hgt_850"]) == (lat[12], lon[3], 1400.0)
    assert (p["max_wind_850_lat"], p["max_wind_850_lon"], p["max_wind_850"]) == (lat[5], lon[15], 33.0)
    row = pd.Series({"Lat": -30.0, "Lon": -50.0, "min_max_zeta_850": -9e-5, "min_hgt_850": np.nan, "max_wind_850": 40.0})
    q = dg.box_positions(zeta, hgt, w, lat, lon, lim, row)
    assert q["min_max_zeta_850"] == -9e-5 and q["min_hgt_850"] == 1400.0 and q["max_wind_850"] == 40.0


def _fields(seed=3, ny=41, nx=57, nonuni=False):
    rng = np.random.default_rng(seed)
    lat = np.linspace(-50.0, -10.0, ny)
    lon = np.linspace(-80.0, -24.0, nx)
    if nonuni:
        lat = np.sort(lat + 0.2 * np.sin(np.arange(ny)))
    phi, lam = np.deg2rad(lat)[:, None], np.deg2rad(lon)[None, :]
    u = 20 * np.cos(phi) * np.sin(2 * lam) + rng.standard_normal((ny, nx))
    v = 8 * np.sin(3 * lam) * np.cos(phi) + rng.standard_normal((ny, nx))
    h = 1500 + 60 * np.sin(2 * phi) * np.cos(lam) + rng.standard_normal((ny, nx))
    return lat, lon, u, v, h


def test_diagnostics_match_the_oracle_restatement():
    """diagnostics.py against oracle/track_diagnostics.py: an independent statement of MetPy's three-point derivative and the
    spherical vorticity (parity with MetPy itself stays unpinned, see that module), and of the reference's own get_position /
    find_extremum_coordinates (lec_moving_framework.py:269-417, tools.py:95-128)."""
    from oracle import track_diagnostics as td
    for nonuni in (False, True):
        lat, lon, u, v, h = _fields(nonuni=nonuni)
        z, zr = dg.vorticity(u, v, lat, lon), td.vorticity_sphere(u, v, lat, lon)
        assert np.max(np.abs(z - zr)) <= 1e-12 * np.max(np.abs(zr))
        w = dg.wind_speed(u, v)
        assert np.array_equal(w, td.wind_speed(u, v))
        for lim in ({"min_lat": -38, "max_lat": -22, "min_lon": -58, "max_lon": -42, "central_lat": -30, "central_lon": -50},
                    {"min_lat": -50, "max_lat": -10, "min_lon": -80, "max_lon": -24, "central_lat": -30.2, "central_lon": -51.7}):
            for row, use_zeta in ((None, False), (pd.Series({"Lat": -30.0, "Lon": -50.0}), True),
                                  (pd.Series({"Lat": -30.0, "Lon": -50.0, "min_max_zeta_850": -9e-5, "min_hgt_850": np.nan, "max_wind_850": 40.0}), False)):
                a = dg.box_positions(z, h, w, lat, lon, lim, row, use_zeta)
                b = td.get_position(z, h, w, lat, lon, lim, row, use_zeta)
                assert a == b, (lim, use_zeta)


def test_nan_inside_the_box_is_skipped_values_and_positions():
    """A below-ground NaN at 850 hPa inside the box: the reference's xarray .min() / .max() skip it for the VALUES; its positions come
    from a plain argmin / argmax and land on the NaN cell -- a defect this engine does not reproduce (positions skip NaN too)."""
    from oracle import track_diagnostics as td
    lat, lon, u, v, h = _fields(seed=8)
    z, w = dg.vorticity(u, v, lat, lon), dg.wind_speed(u, v)
    for a in (z, h, w):
        a[20, 30] = np.nan
    lim = {"min_lat": -38, "max_lat": -22, "min_lon": -58, "max_lon": -42, "central_lat": -30, "central_lon": -50}
    got, ref = dg.box_positions(z, h, w, lat, lon, lim), td.get_position(z, h, w, lat, lon, lim)
    for k in ("min_max_zeta_850", "min_hgt_850", "max_wind_850"):
        assert got[k] == ref[k] and np.isfinite(got[k])                       # values: identical, NaN skipped
        assert (ref[k + "_lat"], ref[k + "_lon"]) == (lat[20], lon[30])       # the reference's position: the NaN cell
        assert (got[k + "_lat"], got[k + "_lon"]) != (lat[20], lon[30])       # ours: where the reported value actually is
    jj, ii = np.flatnonzero((lat >= -38) & (lat <= -22)), np.flatnonzero((lon >= -58) & (lon <= -42))
    sub = h[np.ix_(jj, ii)]
    j, i = np.unravel_index(np.nanargmin(sub), sub.shape)
    assert (got["min_hgt_850_lat"], got["min_hgt_850_lon"]) == (lat[jj][j], lon[ii][i])
