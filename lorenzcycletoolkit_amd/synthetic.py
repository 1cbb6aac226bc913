"""Deterministic synthetic (time, level, lat, lon) fields generated on the device.

The recipe of SURVEY.md section 8(d): a smooth climatological part plus Gaussian noise, seeded per
*global* time step (seed + t_global) so that time-sharded ranks generate identical data for the time
steps they share (halo steps included).  Used by bench.py and the full-size parity tests; there is
no network for real reanalysis files.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from .constants import G


def era5_like_levels() -> np.ndarray:
    """37 isobaric levels, all >= 10 hPa (the reference drops levels above 10 hPa,
    preprocessing.py:364-365), ascending, in Pa."""
    hpa = [10, 20, 30, 40, 50, 60, 70, 85, 100, 110, 125, 135, 150, 175, 200, 225, 250, 300, 350, 400, 450,
           500, 550, 600, 650, 700, 750, 775, 800, 825, 850, 875, 900, 925, 950, 975, 1000]
    return np.asarray(hpa, dtype=np.float64) * 100.0


def era5_grid():
    """0.25-degree global grid: 721 latitudes S->N, 1440 longitudes in [-180, 179.75]."""
    return np.linspace(-90.0, 90.0, 721), np.linspace(-180.0, 179.75, 1440)


def synthetic_cube(nt: int, level_pa, lat_deg, lon_deg, device, dtype=torch.float64, seed: int = 1234,
                   t0_global: int = 0) -> Dict[str, torch.Tensor]:
    """Fields for global time steps t0_global .. t0_global + nt - 1.

    T = 288 (p/1e5)^0.19 + 10 cos(2 phi) (p/1e5) + N1      K      (sigma ~ 0.8 > 0.03)
    u = 25 cos(phi) (1 - p/1.2e5) + 5 N2                   m/s
    v = 3 sin(2 lambda) cos(phi) + 3 N3                    m/s
    omega = 0.05 sin(3 lambda) cos(phi) + 0.1 N4           Pa/s
    Phi = g 7000 ln(1e5/p) + 100 N5                        m2/s2
    """
    dev = torch.device(device)
    f64 = dict(dtype=torch.float64, device=dev)
    p = torch.as_tensor(np.asarray(level_pa, dtype=np.float64), **f64)[:, None, None]
    phi = torch.deg2rad(torch.as_tensor(np.asarray(lat_deg, dtype=np.float64), **f64))[None, :, None]
    lam = torch.deg2rad(torch.as_tensor(np.asarray(lon_deg, dtype=np.float64), **f64))[None, None, :]
    nl, ny, nx = p.shape[0], phi.shape[1], lam.shape[2]
    base = {
        "tair": 288.0 * (p / 1e5) ** 0.19 + 10.0 * torch.cos(2 * phi) * (p / 1e5) + 0.0 * lam,
        "u": 25.0 * torch.cos(phi) * (1 - p / 1.2e5) + 0.0 * lam,
        "v": 3.0 * torch.sin(2 * lam) * torch.cos(phi) + 0.0 * p,
        "omega": 0.05 * torch.sin(3 * lam) * torch.cos(phi) + 0.0 * p,
        "geopt": G * 7000.0 * torch.log(1e5 / p) + 0.0 * phi + 0.0 * lam,
    }
    amp = {"tair": 1.0, "u": 5.0, "v": 3.0, "omega": 0.1, "geopt": 100.0}
    out = {k: torch.empty((nt, nl, ny, nx), dtype=dtype, device=dev) for k in base}
    # Few dispatches: per step and field only the draw itself (the generator is re-seeded per GLOBAL step, and a draw's numbers
    # depend on its size, so the draws stay one per step and field); the scaling and the smooth part are applied to a group of steps
    # at once (two launches per group and field instead of three per step and field -- a T = 512 moving cube was ~10,000 launches,
    # and rocprofv3's counter collection does not survive a backlog of that many unsynchronised instrumented launches -- stuck at
    # 10,000, a segmentation fault in its launch path at 17,000: profiles/r05_notes.md section 2).  The arithmetic is
    # the same two roundings as before, fl(fl(amp * N) + base), then the storage dtype: the cubes are bit-identical to round 4's.
    names = ("tair", "u", "v", "omega", "geopt")
    group = max(1, min(nt, (1 << 30) // (8 * nl * ny * nx)))         # steps whose draws of ONE field fit 1 GiB
    drawn = {k: torch.empty((group, nl, ny, nx), **f64) for k in (names if group > 1 else names[:1])}
    gen = torch.Generator(device=dev)
    for g0 in range(0, nt, group):
        g1 = min(nt, g0 + group)
        for t in range(g0, g1):               # per step: the five draws in the order T, u, v, omega, Phi from that step's seed
            gen.manual_seed(seed + t0_global + t)
            for k in names:
                if group == 1:                # a full-size grid: one step at a time through one buffer
                    n = drawn[names[0]][0]
                    torch.randn((nl, ny, nx), generator=gen, out=n)
                    n.mul_(amp[k])
                    torch.add(base[k], n, out=out[k][t])
                else:
                    torch.randn((nl, ny, nx), generator=gen, out=drawn[k][t - g0])
        if group > 1:
            for k in names:
                d = drawn[k][:g1 - g0]
                d.mul_(amp[k])
                torch.add(base[k], d, out=out[k][g0:g1])
        if dev.type == "cuda":                # no deep backlog of launches (the buffers are reused; and see the note above)
            torch.cuda.synchronize(dev)
    return out
