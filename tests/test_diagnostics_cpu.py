"""850-hPa diagnostics, host side (no GPU): the derivative stencils handed to lec_track_diag, the box index ranges and the
reference's precedence rules.  The kernel itself is tested in tests/test_gpu_diagnostics.py."""
import numpy as np
import pandas as pd
import pytest

from lorenzcycletoolkit_amd import diagnostics as dg


def _apply(tab, f):
    i0 = tab[:, 0].astype(int)
    return tab[:, 1] * f[i0] + tab[:, 2] * f[i0 + 1] + tab[:, 3] * f[i0 + 2]


def test_three_point_stencils_are_exact_on_quadratics_also_at_the_ends():
    x = np.array([0.0, 0.5, 1.5, 2.0, 4.0])
    f = 3 * x ** 2 - 2 * x + 1
    tab = dg._three_point(x)
    assert np.allclose(_apply(tab, f), 6 * x - 2, rtol=1e-13)
    assert tab[:, 0].tolist() == [0, 0, 1, 2, 2]
    with pytest.raises(ValueError):
        dg._three_point(np.array([0.0, 1.0]))


def test_stencils_match_the_oracle_derivative():
    """The same numbers as oracle/track_diagnostics.py's point-by-point Lagrange derivative (an independent statement of
    metpy.calc.first_derivative)."""
    from oracle import track_diagnostics as td
    rng = np.random.default_rng(1)
    lat = np.sort(np.linspace(-50, -10, 17) + 0.3 * rng.standard_normal(17))
    lon = np.linspace(-80, -24, 23)
    lontab, lattab = dg.stencil_tables(lat, lon)
    f = rng.standard_normal(23)
    assert np.allclose(_apply(lontab, f), td.first_derivative_3pt(f, np.deg2rad(lon), 0), rtol=1e-12, atol=1e-12)
    g = rng.standard_normal(17)
    assert np.allclose(_apply(lattab[:, :4], g), td.first_derivative_3pt(g, np.deg2rad(lat), 0), rtol=1e-12, atol=1e-12)
    assert np.array_equal(lattab[:, 4], np.cos(np.deg2rad(lat))) and np.array_equal(lattab[:, 5], np.tan(np.deg2rad(lat)))


def test_box_ranges_are_inclusive_label_slices():
    lat = np.linspace(-40, -20, 21)
    lon = np.linspace(-60, -40, 21)
    lim = {"min_lat": -38, "max_lat": -22, "min_lon": -58.5, "max_lon": -42, "central_lat": -30.4, "central_lon": -50.6}
    iw, ie, js, jn, jc, ic = dg.box_ranges(lat, lon, lim)
    assert (lat[js], lat[jn], lon[iw], lon[ie]) == (-38.0, -22.0, -58.0, -42.0)
    assert (lat[jc], lon[ic]) == (-30.0, -51.0)
    with pytest.raises(ValueError):
        dg.box_ranges(lat, lon, dict(lim, min_lat=10, max_lat=20))


def test_positions_prefer_track_values_and_follow_the_hemisphere_rule():
    lat = np.linspace(-40, -20, 21)
    lon = np.linspace(-60, -40, 21)
    lim = {"min_lat": -38, "max_lat": -22, "min_lon": -58, "max_lon": -42, "central_lat": -30, "central_lon": -50}
    val = np.array([-5e-5, 3e-5, 1400.0, 33.0, -1e-5])                 # zeta min, zeta max, height min, wind max, zeta at the centre
    pos = np.array([7, 9, 4, 4, 12, 3, 5, 15], dtype=np.int32)
    p = dg.positions(val, pos, lat, lon, lim)
    assert (p["min_max_zeta_850_lat"], p["min_max_zeta_850_lon"], p["min_max_zeta_850"]) == (lat[7], lon[9], -5e-5)      # southern: the minimum
    assert (p["min_hgt_850_lat"], p["min_hgt_850_lon"], p["min_hgt_850"]) == (lat[12], lon[3], 1400.0)
    assert (p["max_wind_850_lat"], p["max_wind_850_lon"], p["max_wind_850"]) == (lat[5], lon[15], 33.0)
    north = dg.positions(val, pos, -lat[::-1], lon, {**lim, "min_lat": 22, "max_lat": 38, "central_lat": 30})
    assert north["min_max_zeta_850"] == 3e-5 and north["min_max_zeta_850_lat"] == (-lat[::-1])[4]                          # northern: the maximum
    row = pd.Series({"Lat": -30.0, "Lon": -50.0, "min_max_zeta_850": -9e-5, "min_hgt_850": np.nan, "max_wind_850": 40.0})
    q = dg.positions(val, pos, lat, lon, lim, row)
    assert q["min_max_zeta_850"] == -9e-5 and q["min_hgt_850"] == 1400.0 and q["max_wind_850"] == 40.0
    # a NaN in the track file's zeta column is USED (the reference takes the column without looking, lec_moving_framework.py:312-313),
    # while NaN in the two other columns falls back to the data (:356-360, :375-379)
    nanrow = pd.Series({"Lat": -30.0, "Lon": -50.0, "min_max_zeta_850": np.nan, "min_hgt_850": np.nan, "max_wind_850": np.nan})
    qn = dg.positions(val, pos, lat, lon, lim, nanrow)
    assert np.isnan(qn["min_max_zeta_850"]) and qn["min_hgt_850"] == 1400.0 and qn["max_wind_850"] == 33.0
    z = dg.positions(val, pos, lat, lon, lim, pd.Series({"Lat": -30.0, "Lon": -50.0}), use_track_zeta=True)
    assert z["min_max_zeta_850"] == -1e-5                                                                                   # -z: vorticity at the box centre
    none = dg.positions(np.array([np.nan] * 5), np.full(8, -1, dtype=np.int32), lat, lon, lim)
    assert np.isnan(none["min_hgt_850"]) and np.isnan(none["min_hgt_850_lat"])


def test_vorticity_tables_of_both_formulations():
    """The coefficient tables handed to lec_track_diag: "spherical" = the separable stencils over Re cos(phi) / Re plus the curvature
    coefficient; "metpy_no_crs" = first_derivative on great-circle spacings of pyproj's default sphere, no curvature.  Both reproduce
    the oracle's point-by-point evaluation; they differ where they should (radius, chord vs parallel, curvature)."""
    from oracle import track_diagnostics as td
    rng = np.random.default_rng(5)
    lat = np.sort(np.linspace(-55, -12, 19) + 0.2 * rng.standard_normal(19))
    lon = np.sort(np.linspace(-80, -30, 27) + 0.2 * rng.standard_normal(27))
    u, v = rng.standard_normal((19, 27)), rng.standard_normal((19, 27))

    def apply(form):
        xc, yc, cv = dg.vorticity_tables(lat, lon, form)
        assert xc.shape == (19, 27, 3) and yc.shape == (19, 3) and cv.shape == (19,)
        i0 = np.clip(np.arange(27) - 1, 0, 24)
        j0 = np.clip(np.arange(19) - 1, 0, 16)
        dv = sum(xc[:, :, k] * v[:, i0 + k] for k in range(3))
        du = sum(yc[:, k, None] * u[j0 + k, :] for k in range(3))
        return dv - du + cv[:, None] * u

    assert np.allclose(apply("spherical"), td.vorticity_sphere(u, v, lat, lon), rtol=1e-11, atol=1e-18)
    assert np.allclose(apply("metpy_no_crs"), td.vorticity_no_crs(u, v, lat, lon), rtol=1e-9, atol=1e-17)
    assert np.all(dg.vorticity_tables(lat, lon, "metpy_no_crs")[2] == 0) and np.all(dg.vorticity_tables(lat, lon, "spherical")[2] < 0)
    # great-circle arc of one degree of longitude at 60 S on the default sphere: a little shorter than the parallel
    arc = dg.great_circle_arc(np.deg2rad(-60.0), 0.0, np.deg2rad(-60.0), np.deg2rad(1.0), dg.PYPROJ_SPHERE_RADIUS)
    par = dg.PYPROJ_SPHERE_RADIUS * np.cos(np.deg2rad(60.0)) * np.deg2rad(1.0)
    assert 0 < par - arc < 1e-4 * par
    with pytest.raises(ValueError):
        dg.vorticity_tables(lat, lon, "wgs84")
