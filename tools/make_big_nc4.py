#!/opt/conda/bin/python3.9
"""Writes an ERA5-size NetCDF-4 file (shuffle + deflate chunks) to local disk for tools/bench_cli.py -- NOT a fixture, never committed.

    /opt/conda/bin/python3.9 tools/make_big_nc4.py --out /tmp/era5_like.nc --timesteps 96

Needs h5py (this image's /opt/conda interpreter).  Layout: what the CDS delivered for years and most archives hold -- int16 with
scale_factor / add_offset / _FillValue, dimensions time (hours since 1900), level (hPa, top down), latitude (N -> S), longitude
(0 .. 359.75), chunks of 1 x 1 x 361 x 720, shuffle + deflate -- or, with --layout cds_new, today's CDS layout (valid_time int64
seconds, pressure_level float64 descending, float32 fields with a NaN _FillValue, scalar `number`, string `expver`).
The fields are smooth profiles plus rolled copies of one noise plane (cheap to make, compresses like analysed fields: ratio ~1.6-2).
Only --distinct time steps are generated and compressed (16 threads, zlib releases the GIL); the later steps get the same compressed
bytes written as their own chunks (H5Dwrite_chunk), so a 96-step, 20-GB file takes well under a minute."""
import argparse
import os
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import h5py
import numpy as np

G = 9.80665


def era5_levels_hpa():
    return np.array([1, 2, 3, 5, 7, 10, 20, 30, 50, 70, 100, 125, 150, 175, 200, 225, 250, 300, 350, 400, 450, 500, 550, 600, 650, 700, 750,
                     775, 800, 825, 850, 875, 900, 925, 950, 975, 1000], dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--timesteps", type=int, default=96)
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--ny", type=int, default=721)
    ap.add_argument("--nx", type=int, default=1440)
    ap.add_argument("--layout", choices=["era5_int16", "cds_new"], default="era5_int16")
    ap.add_argument("--deflate", type=int, default=1)
    ap.add_argument("--noise", type=float, default=0.25, help="amplitude of the unresolved part relative to the synthetic recipe of bench.py")
    a = ap.parse_args()
    t_begin = time.time()
    T, D, ny, nx = a.timesteps, min(a.distinct, a.timesteps), a.ny, a.nx
    lev = era5_levels_hpa()
    nl = lev.size
    lat = np.linspace(90.0, -90.0, ny)
    lon = np.arange(nx) * (360.0 / nx)
    new = a.layout == "cds_new"
    if new:
        lev = lev[::-1].copy()                       # 1000 ... 1: descending, as the CDS stores it
    cj, ci = (ny + 1) // 2, (nx + 1) // 2
    rng = np.random.default_rng(3)
    noise = rng.standard_normal((ny, nx)).astype(np.float32)
    phi, lam = np.deg2rad(lat)[:, None].astype(np.float32), np.deg2rad(lon)[None, :].astype(np.float32)

    def field(name, t, k):
        p = np.float32(lev[k] * 100.0 / 1e5)
        n = np.roll(noise, (17 * k + 5 * t + 3 * ord(name[0]), 31 * k + 11 * t), axis=(0, 1)) * np.float32(a.noise)
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
    # value ranges for the packing (from the first step)
    pack = {}
    if not new:
        for name in names:
            lo = min(float(field(name, 0, k).min()) for k in (0, nl // 2, nl - 1)) - 50.0
            hi = max(float(field(name, 0, k).max()) for k in (0, nl // 2, nl - 1)) + 50.0
            if name == "z":
                lo, hi = -2000.0, G * 7000.0 * np.log(1e5 / 100.0) + 3000.0
            pack[name] = ((hi - lo) / 64000.0, 0.5 * (hi + lo))
    dt = np.dtype("<f4") if new else np.dtype("<i2")
    es = dt.itemsize

    def chunks_of(job):
        name, t, k = job
        f = field(name, t, k)
        if new:
            q = f.astype("<f4")
        else:
            s, o = pack[name]
            q = np.clip(np.round((f - o) / s), -32000, 32000).astype("<i2")
        out = []
        for j0 in range(0, ny, cj):
            for i0 in range(0, nx, ci):
                blk = np.zeros((cj, ci), dtype=dt)
                part = q[j0: j0 + cj, i0: i0 + ci]
                blk[: part.shape[0], : part.shape[1]] = part
                raw = blk.reshape(-1).view(np.uint8).reshape(-1, es).T.tobytes()          # the shuffle filter's byte planes
                out.append(((j0, i0), zlib.compress(raw, a.deflate)))
        return job, out

    jobs = [(name, t, k) for name in names for t in range(D) for k in range(nl)]
    with ThreadPoolExecutor(16) as pool:
        done = dict(pool.map(chunks_of, jobs))
    t_comp = time.time()
    comp_bytes = sum(len(z) for out in done.values() for _, z in out)
    with h5py.File(a.out, "w", libver=("earliest", "v110")) as h:
        h.attrs["Conventions"] = np.string_("CF-1.7")
        tn, ln = ("valid_time", "pressure_level") if new else ("time", "level")
        if new:
            tv = h.create_dataset(tn, data=(1577836800 + 3600 * np.arange(T)).astype(np.int64))
            tv.attrs["units"] = "seconds since 1970-01-01"
            tv.attrs["calendar"] = "proleptic_gregorian"
            lv = h.create_dataset(ln, data=lev.astype(np.float64))
            lv.attrs["units"] = "hPa"
            h.create_dataset("number", data=np.int64(0))
            h.create_dataset("expver", data=np.array(["0001"] * T, dtype=object), dtype=h5py.string_dtype())
        else:
            tv = h.create_dataset(tn, data=(1051896 + np.arange(T)).astype(np.int32))       # 2020-01-01 00:00
            tv.attrs["units"] = np.string_("hours since 1900-01-01 00:00:00.0")
            tv.attrs["calendar"] = np.string_("gregorian")
            lv = h.create_dataset(ln, data=lev.astype(np.int32))
            lv.attrs["units"] = np.string_("millibars")
        la = h.create_dataset("latitude", data=lat.astype(np.float64 if new else np.float32))
        la.attrs["units"] = np.string_("degrees_north")
        lo = h.create_dataset("longitude", data=lon.astype(np.float64 if new else np.float32))
        lo.attrs["units"] = np.string_("degrees_east")
        for d, n in ((tv, tn), (lv, ln), (la, "latitude"), (lo, "longitude")):
            d.make_scale(n)
        for name in names:
            kw = dict(fillvalue=np.float32(np.nan)) if new else {}
            d = h.create_dataset(name, (T, nl, ny, nx), dtype=dt, chunks=(1, 1, cj, ci), shuffle=True, compression="gzip",
                                 compression_opts=a.deflate, **kw)
            if new:
                d.attrs["_FillValue"] = np.float32(np.nan)
            else:
                d.attrs["scale_factor"], d.attrs["add_offset"] = np.float64(pack[name][0]), np.float64(pack[name][1])
                d.attrs["_FillValue"] = np.int16(-32767)
                d.attrs["missing_value"] = np.int16(-32767)
            d.attrs["units"] = np.string_("1")
            for i, dn in enumerate((tn, ln, "latitude", "longitude")):
                d.dims[i].attach_scale(h[dn])
            for t in range(T):
                for k in range(nl):
                    for (j0, i0), z in done[(name, t % D, k)]:
                        d.id.write_direct_chunk((t, k, j0, i0), z)
    size = os.path.getsize(a.out)
    raw_bytes = 5 * T * nl * ny * nx * es
    print(f"{a.out}: {T} steps ({D} distinct) x {nl} x {ny} x {nx} {dt.name}, layout {a.layout}, {size / 1e9:.2f} GB on disk, "
          f"{raw_bytes / 1e9:.1f} GB raw, ratio {raw_bytes / max(size, 1):.2f}; fields + deflate {t_comp - t_begin:.1f} s, "
          f"write {time.time() - t_comp:.1f} s ({comp_bytes * T / D / 1e9:.2f} GB of chunks)", flush=True)


if __name__ == "__main__":
    main()
