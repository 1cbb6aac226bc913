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
