"""lec_track_diag (850-hPa track diagnostics on the device) against oracle/track_diagnostics.py: an independent restatement of
MetPy's three-point derivative + the spherical vorticity (parity with MetPy itself is unpinned, see that module) and of the
reference's own get_position / find_extremum_coordinates (lec_moving_framework.py:269-417, tools.py:95-128)."""
import numpy as np
import pandas as pd
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from lorenzcycletoolkit_amd import diagnostics as dg
from lorenzcycletoolkit_amd.constants import RE
from oracle import track_diagnostics as td


def _fields(seed=3, nt=3, ny=41, nx=57, nonuni=False):
    rng = np.random.default_rng(seed)
    lat = np.linspace(-50.0, -10.0, ny)
    lon = np.linspace(-80.0, -24.0, nx)
    if nonuni:
        lat = np.sort(lat + 0.2 * np.sin(np.arange(ny)))
    phi, lam = np.deg2rad(lat)[None, :, None], np.deg2rad(lon)[None, None, :]
    u = 20 * np.cos(phi) * np.sin(2 * lam) + rng.standard_normal((nt, ny, nx))
    v = 8 * np.sin(3 * lam) * np.cos(phi) + rng.standard_normal((nt, ny, nx))
    h = 1500 + 60 * np.sin(2 * phi) * np.cos(lam) + rng.standard_normal((nt, ny, nx))
    return lat, lon, u, v, h


LIMS = [{"min_lat": -38, "max_lat": -22, "min_lon": -58, "max_lon": -42, "central_lat": -30, "central_lon": -50},
        {"min_lat": -50, "max_lat": -10, "min_lon": -80, "max_lon": -24, "central_lat": -30.2, "central_lon": -51.7},   # the whole slice: one-sided ends
        {"min_lat": -12.1, "max_lat": -10, "min_lon": -26.5, "max_lon": -24, "central_lat": -11, "central_lon": -25}]    # the last corner


def test_solid_body_rotation_vorticity():
    """u = U cos(phi), v = 0  ->  zeta = 2 U sin(phi) / Re everywhere; the minimum of a southern box sits on its southern edge."""
    lat = np.linspace(-60, -10, 101)
    lon = np.linspace(-80, -20, 61)
    U = 30.0
    u = (U * np.cos(np.deg2rad(lat))[:, None] * np.ones((1, lon.size)))[None]
    lim = {"min_lat": -50, "max_lat": -20, "min_lon": -70, "max_lon": -30, "central_lat": -35, "central_lon": -50}
    val, pos = dg.device_extrema(u, np.zeros_like(u), np.full_like(u, 1500.0), lat, lon, [lim], formulation="spherical")
    want = 2 * U * np.sin(np.deg2rad(np.array([-50.0, -20.0, -35.0]))) / RE
    assert np.allclose(val[0, [0, 1, 4]], want, rtol=3e-4)
    assert lat[pos[0, 0]] == -50.0 and lat[pos[0, 2]] == -20.0
    # the plain Cartesian form (MetPy without a CRS) lacks the curvature term u tan(phi) / Re: solid-body rotation then shows
    # only -du/dy = U sin(phi) / a -- HALF the vorticity.  That is what the reference's runs report if MetPy falls back as documented.
    val2, _ = dg.device_extrema(u, np.zeros_like(u), np.full_like(u, 1500.0), lat, lon, [lim], formulation="metpy_no_crs")
    assert np.allclose(val2[0, [0, 1, 4]], want / 2 * RE / dg.PYPROJ_SPHERE_RADIUS, rtol=3e-4)


@pytest.mark.parametrize("form", ["spherical", "metpy_no_crs"])
@pytest.mark.parametrize("nonuni", [False, True])
def test_kernel_matches_the_oracle_restatement(nonuni, form):
    lat, lon, u, v, h = _fields(nonuni=nonuni)
    if nonuni and form == "metpy_no_crs":        # uneven longitudes too: the arcs along a row are then not multiples of one another
        lon = np.sort(lon + 0.3 * np.cos(np.arange(lon.size)))
    nt = u.shape[0]
    zr = (td.vorticity_sphere if form == "spherical" else td.vorticity_no_crs)(u, v, lat, lon)
    wr = td.wind_speed(u, v)
    for lim in LIMS:
        val, pos = dg.device_extrema(u, v, h, lat, lon, [lim] * nt, formulation=form)
        for t in range(nt):
            for row, use_zeta in ((None, False), (pd.Series({"Lat": -30.0, "Lon": -50.0}), True),
                                  (pd.Series({"Lat": -30.0, "Lon": -50.0, "min_max_zeta_850": -9e-5, "min_hgt_850": np.nan, "max_wind_850": 40.0}), False)):
                got = dg.positions(val[t], pos[t], lat, lon, lim, row, use_zeta)
                ref = td.get_position(zr[t], h[t], wr[t], lat, lon, lim, row, use_zeta)
                for k in ref:
                    if k.endswith("_lat") or k.endswith("_lon") or k in ("min_hgt_850",):
                        assert got[k] == ref[k], (lim, t, k)
                    else:
                        assert abs(got[k] - ref[k]) <= 1e-11 * abs(ref[k]), (lim, t, k, got[k], ref[k])


def test_nan_inside_the_box_is_skipped_values_and_positions():
    """A below-ground NaN at 850 hPa inside the box: the reference's xarray .min() / .max() skip it for the VALUES; its positions come
    from a plain argmin / argmax and land on the NaN cell -- a defect this engine does not reproduce (positions skip NaN too)."""
    lat, lon, u, v, h = _fields(seed=8, nt=1)
    for a in (u, h):
        a[0, 20, 30] = np.nan                       # NaN wind: zeta is NaN at the point and at its four stencil neighbours
    lim = LIMS[0]
    val, pos = dg.device_extrema(u, v, h, lat, lon, [lim], formulation="spherical")
    got = dg.positions(val[0], pos[0], lat, lon, lim)
    z, w = td.vorticity_sphere(u, v, lat, lon)[0], td.wind_speed(u, v)[0]
    ref = td.get_position(z, h[0], w, lat, lon, lim)
    for k in ("min_max_zeta_850", "min_hgt_850", "max_wind_850"):
        assert np.isfinite(got[k]) and abs(got[k] - ref[k]) <= 1e-11 * abs(ref[k])      # values: NaN skipped on both sides
        assert np.isnan([z, h[0], w][("min_max_zeta_850", "min_hgt_850", "max_wind_850").index(k)][
            np.searchsorted(lat, ref[k + "_lat"]), np.searchsorted(lon, ref[k + "_lon"])])   # the reference's position: a NaN cell
    jj, ii = np.flatnonzero((lat >= -38) & (lat <= -22)), np.flatnonzero((lon >= -58) & (lon <= -42))
    for k, a, fn in (("min_max_zeta_850", z, np.nanargmin), ("min_hgt_850", h[0], np.nanargmin), ("max_wind_850", w, np.nanargmax)):
        sub = a[np.ix_(jj, ii)]
        j, i = np.unravel_index(fn(sub), sub.shape)
        assert (got[k + "_lat"], got[k + "_lon"]) == (lat[jj][j], lon[ii][i]), k
    # a box with nothing but NaN
    h[:] = np.nan
    val, pos = dg.device_extrema(u, v, h, lat, lon, [lim])
    assert np.isnan(val[0, 2]) and tuple(pos[0, 4:6]) == (-1, -1)


def test_equal_values_take_the_first_in_row_major_order():
    lat, lon, u, v, h = _fields(seed=9, nt=2)
    h[:] = 1500.0
    h[0, 15, 30] = h[0, 18, 22] = h[0, 18, 35] = 1400.0           # three equal minima: numpy's argmin takes the first in C order
    u[:] = 3.0; v[:] = 4.0                                          # wind speed 5 everywhere: the first box point
    lim = LIMS[0]
    val, pos = dg.device_extrema(u, v, h, lat, lon, [lim, lim])
    iw, ie, js, jn, _, _ = dg.box_ranges(lat, lon, lim)
    assert tuple(pos[0, 4:6]) == (15, 30) and val[0, 2] == 1400.0
    assert tuple(pos[0, 6:8]) == (js, iw) and val[0, 3] == 5.0
    assert tuple(pos[1, 4:6]) == (js, iw) and val[1, 2] == 1500.0


def test_arguments_are_validated():
    lat, lon, u, v, h = _fields(nt=1)
    with pytest.raises(ValueError):
        dg.device_extrema(u, v, h[:, :-1], lat, lon, [LIMS[0]])
    with pytest.raises(ValueError):
        dg.device_extrema(u, v, h, lat, lon, [LIMS[0], LIMS[0]])
    with pytest.raises(ValueError):
        dg.device_extrema(u, v, h, lat, lon, [dict(LIMS[0], min_lat=40, max_lat=50)])
    with pytest.raises(ValueError, match="formulation"):
        dg.device_extrema(u, v, h, lat, lon, [LIMS[0]], formulation="wgs84")
