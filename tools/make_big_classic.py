#!/usr/bin/env python3
"""Writes an ERA5-size CLASSIC NetCDF file (64-bit offsets, int16 + scale_factor / add_offset / _FillValue, big-endian like every
classic file; time hours since 1900, levels hPa top down, latitudes N -> S, longitudes 0 .. 359.75) to local disk for
tools/bench_cli.py -- not a fixture.  Same synthetic fields as tools/make_big_nc4.py; --distinct steps are generated, the rest repeat.

    python tools/make_big_classic.py --out /tmp/era5_classic.nc --timesteps 24"""
import argparse
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from scipy.io import netcdf_file

G = 9.80665


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--timesteps", type=int, default=24)
    ap.add_argument("--distinct", type=int, default=4)
    ap.add_argument("--ny", type=int, default=721)
    ap.add_argument("--nx", type=int, default=1440)
    a = ap.parse_args()
    t_begin = time.time()
    T, D, ny, nx = a.timesteps, min(a.distinct, a.timesteps), a.ny, a.nx
    lev = np.array([1, 2, 3, 5, 7, 10, 20, 30, 50, 70, 100, 125, 150, 175, 200, 225, 250, 300, 350, 400, 450, 500, 550, 600, 650, 700, 750,
                    775, 800, 825, 850, 875, 900, 925, 950, 975, 1000], dtype=np.float64)
    nl = lev.size
    lat, lon = np.linspace(90.0, -90.0, ny), np.arange(nx) * (360.0 / nx)
    noise = np.random.default_rng(3).standard_normal((ny, nx)).astype(np.float32)
    phi, lam = np.deg2rad(lat)[:, None].astype(np.float32), np.deg2rad(lon)[None, :].astype(np.float32)

    def field(name, t, k):
        p = np.float32(lev[k] * 100.0 / 1e5)
        n = np.roll(noise, (17 * k + 5 * t + 3 * ord(name[0]), 31 * k + 11 * t), axis=(0, 1)) * np.float32(0.25)
        wave = np.sin(3 * lam + np.float32(0.2 * t)) * np.cos(phi)
        if name == "t":
            return 288.0 * p ** 0.19 + 10.0 * np.cos(2 * phi) * p + 2.0 * wave + n
        if name == "u":
            return 25.0 * np.cos(phi) * (1 - p / 1.2) + 4.0 * wave + 5.0 * n
        if name == "v":
            return 3.0 * np.sin(2 * lam) * np.cos(phi) + 3.0 * n
        if name == "w":
            return 0.05 * wave + 0.1 * n
        return G * 7000.0 * np.log(1.0 / max(p, 1e-5)) + 100.0 * wave + 100.0 * n

    names = ["t", "u", "v", "w", "z"]
    pack = {}
    for name in names:
        lo = min(float(field(name, 0, k).min()) for k in (0, nl // 2, nl - 1)) - 50.0
        hi = max(float(field(name, 0, k).max()) for k in (0, nl // 2, nl - 1)) + 50.0
        if name == "z":
            lo, hi = -2000.0, G * 7000.0 * np.log(1e5 / 100.0) + 3000.0
        pack[name] = ((hi - lo) / 64000.0, 0.5 * (hi + lo))

    def packed(job):
        name, t, k = job
        s, o = pack[name]
        return job, np.clip(np.round((field(name, t, k) - o) / s), -32000, 32000).astype(">i2")

    with ThreadPoolExecutor(16) as pool:
        done = dict(pool.map(packed, [(n, t, k) for n in names for t in range(D) for k in range(nl)]))
    f = netcdf_file(a.out, "w", version=2)
    for n, sz in (("time", T), ("level", nl), ("latitude", ny), ("longitude", nx)):
        f.createDimension(n, sz)
    tv = f.createVariable("time", "i", ("time",)); tv[:] = 1051896 + np.arange(T); tv.units = "hours since 1900-01-01 00:00:00.0"
    lv = f.createVariable("level", "i", ("level",)); lv[:] = lev.astype(np.int32); lv.units = "millibars"
    la = f.createVariable("latitude", "f", ("latitude",)); la[:] = lat
    lo_ = f.createVariable("longitude", "f", ("longitude",)); lo_[:] = lon
    for name in names:
        v = f.createVariable(name, "h", ("time", "level", "latitude", "longitude"))
        for t in range(T):
            for k in range(nl):
                v[t, k] = done[(name, t % D, k)]
        v.scale_factor, v.add_offset = float(pack[name][0]), float(pack[name][1])
        v._FillValue = np.int16(-32767)
    f.close()
    print(f"{a.out}: {T} steps ({D} distinct) x {nl} x {ny} x {nx} int16 classic NetCDF, {os.path.getsize(a.out) / 1e9:.2f} GB, "
          f"written in {time.time() - t_begin:.1f} s", flush=True)


if __name__ == "__main__":
    main()
