#!/usr/bin/env python3
"""PCIe-inclusive rate of the device ingest at the headline grid (37 x 721 x 1440): host memory -> pinned ->
GPU -> lec_ingest -> lec_rowstats per chunk -> one lec_reduce.  Not the bench.py metric (that one has the
inputs resident in HBM); DESIGN.md quotes this number next to it.

  python tools/bench_ingest.py --src i16 --timesteps 16 --chunk 4
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", choices=["i16", "f32", "f64", "i16z"], default="i16", help="dtype of the 'file' variables in host memory; "
                    "i16z: int16 in shuffled + deflated HDF5-style chunks (what a NetCDF-4 ERA5 file holds)")
    ap.add_argument("--inflate", choices=["auto", "host", "device"], default="auto", help="i16z: where the chunks are inflated")
    ap.add_argument("--deflate-level", type=int, default=4)
    ap.add_argument("--cycle", type=int, default=0, help="i16z: build only this many distinct time steps and repeat them (long series without "
                    "the host memory for them); 0 = all distinct")
    ap.add_argument("--slots", type=int, default=None, help="pipeline slots (default: 2, or 3 with the device inflate)")
    ap.add_argument("--chunk-shape", default="1,1,361,720", help="i16z: HDF5 chunk shape (time, level, lat, lon)")
    ap.add_argument("--timesteps", type=int, default=16)
    ap.add_argument("--chunk", type=int, default=4, help="time steps per pipeline chunk; 0 = lec_streamed's own choice")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--staging", choices=["auto", "staged", "registered"], default="auto", help="pinned staging copy, or the source memory "
                    "registered with the HIP runtime and copied from directly (auto: registered where possible)")
    ap.add_argument("--ny", type=int, default=721)
    ap.add_argument("--nx", type=int, default=1440)
    args = ap.parse_args()
    import pandas as pd
    import torch
    from lorenzcycletoolkit_amd import dataset as ds
    from lorenzcycletoolkit_amd import ingest
    from lorenzcycletoolkit_amd.synthetic import era5_like_levels, synthetic_cube

    dev = torch.device("cuda:0")
    level = era5_like_levels()
    lat = np.linspace(-90.0, 90.0, args.ny)
    lon = np.linspace(-180.0, 180.0 - 360.0 / args.nx, args.nx)
    T = args.timesteps
    T_build = min(T, args.cycle) if (args.cycle and args.src == "i16z") else T
    # the "file": ERA5 axis conventions (lat N -> S, lon 0..360, levels in hPa from the top down), big-endian
    file_lat, file_lon = lat[::-1].copy(), np.where(lon < 0, lon + 360.0, lon)
    lon_order = np.argsort(file_lon, kind="stable")
    file_lon = file_lon[lon_order]
    file_lev = (level / 100.0)
    names = {"Air Temperature": "t", "Eastward Wind Component": "u", "Northward Wind Component": "v",
             "Omega Velocity": "w", "Geopotential": "z", "Longitude": "longitude", "Latitude": "latitude",
             "Time": "time", "Vertical Level": "level"}
    variables = {}
    keys = {"t": "tair", "u": "u", "v": "v", "w": "omega", "z": "geopt"}
    lon_idx = torch.as_tensor(lon_order, device=dev)
    for t0 in range(0, T_build, 4):
        n = min(4, T_build - t0)
        f = synthetic_cube(n, level, lat, lon, device=dev, dtype=torch.float64, seed=1234, t0_global=t0)
        for name, key in keys.items():
            a = f[key].flip(2).index_select(3, lon_idx)          # file order of lat / lon
            if args.src in ("i16", "i16z"):
                if t0 == 0:
                    lo, hi = float(a.min()) - 1.0, float(a.max()) + 1.0
                    variables[name] = dict(scale=(hi - lo) / 64000.0, offset=0.5 * (hi + lo), parts=[])
                v = variables[name]
                q = torch.clamp(torch.round((a - v["offset"]) / v["scale"]), -32000, 32000).to(torch.int16)
                v["parts"].append(q.cpu().numpy())
            else:
                variables.setdefault(name, dict(scale=None, offset=None, parts=[]))["parts"].append(
                    a.to(torch.float32 if args.src == "f32" else torch.float64).cpu().numpy())
        del f
    torch.cuda.empty_cache()
    raw_vars = {}
    ratio = None
    for name, v in variables.items():
        a = np.concatenate(v["parts"], axis=0)
        if args.src == "i16z":
            dv = DeflatedVar(a, tuple(int(x) for x in args.chunk_shape.split(",")), args.deflate_level, repeat_to=T)
            ratio = dv.ratio
            raw_vars[name] = ds.RawVariable(dv, v["scale"], v["offset"], -32767.0)
            continue
        be = a.astype(a.dtype.newbyteorder(">"))                   # classic NetCDF stores big-endian
        raw_vars[name] = ds.RawVariable(be, v["scale"], v["offset"], -32767.0 if args.src == "i16" else None)
    time = np.datetime64("2020-01-01T00", "ns") + (np.arange(T) * 3600 * 10 ** 9).astype("timedelta64[ns]")
    raw = ds.RawDataset(raw_vars, file_lat, file_lon, file_lev, time, names, "hPa", "Geopotential")
    px = ds.process_index(raw.lat, raw.lon, raw.level, raw.time, raw.level_units, raw.names, argparse.Namespace(track=False))
    i32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
    plan = ingest.IngestPlan(np.arange(T), i32(px.ik), i32(px.ij), i32(px.io), px.lat, px.lon, px.level, px.time)
    assert np.allclose(plan.lat, lat) and np.allclose(plan.lon, lon)
    df = pd.DataFrame({"Variable": list(names.values()), "Units": ["K", "m/s", "m/s", "Pa/s", "m**2/s**2", "", "", "", ""]},
                      index=list(names.keys()))
    limits = (lon[0], lon[-1], lat[0], lat[-1])
    best, stats = None, {}
    for r in range(args.repeat + 1):                               # first pass = warm-up (pinned allocations, page faults)
        torch.cuda.synchronize()
        t0 = time_now()
        res = ingest.lec_fixed_streamed(raw, plan, df, limits, chunk_steps=args.chunk or None, stats=stats, staging=args.staging, inflate=args.inflate, slots=args.slots)
        torch.cuda.synchronize()
        dt = time_now() - t0
        if r > 0:
            best = dt if best is None else min(best, dt)
    finite = bool(torch.isfinite(res.scalars).all())
    print(json.dumps({
        "metric": "LEC timesteps/sec from HOST memory (PCIe-inclusive device ingest), all terms, 37x%dx%d" % (args.ny, args.nx),
        "value": T / best, "unit": "timesteps/s", "source_dtype": args.src, "timesteps": T, "chunk_steps": stats.get("chunk_steps", args.chunk),
        "seconds": best, "bytes_moved": stats["bytes_moved"], "host_to_device_GBs": stats["bytes_moved"] / best / 1e9,
        "storage_on_device": stats["storage"], "staging": stats["staging"], "host_staging_seconds": stats["host_staging_seconds"],
        "register_calls": stats.get("register_calls"), "results_finite": finite, "inflate": stats.get("inflate"),
        "deflate_ratio": ratio, "decoded_GBs": (T * 5 * len(level) * args.ny * args.nx * 2 / best / 1e9) if args.src == "i16z" else None}))


class DeflatedVar:
    """An int16 variable as a NetCDF-4 file holds it -- little-endian, HDF5 chunks, shuffle + deflate -- kept in host memory: the face
    hdf5_lite.H5Variable shows the ingest (shape, dtype, ``var[t]`` inflating on the host's thread pool, ``chunk_streams()``)."""

    def __init__(self, a, chunk, level, repeat_to=None):
        import zlib
        from concurrent.futures import ThreadPoolExecutor
        self.shape, self.dtype, self.chunk = a.shape, np.dtype("<i2"), chunk
        ct, ck, cj, ci = chunk
        origins = [(t, k, j, i) for t in range(0, a.shape[0], ct) for k in range(0, a.shape[1], ck)
                   for j in range(0, a.shape[2], cj) for i in range(0, a.shape[3], ci)]

        def pack(o):
            blk = np.zeros(chunk, dtype="<i2")
            part = a[o[0]: o[0] + ct, o[1]: o[1] + ck, o[2]: o[2] + cj, o[3]: o[3] + ci]
            blk[: part.shape[0], : part.shape[1], : part.shape[2], : part.shape[3]] = part
            return zlib.compress(blk.reshape(-1).view(np.uint8).reshape(-1, 2).T.tobytes(), level)

        with ThreadPoolExecutor(16) as pool:
            streams = list(pool.map(pack, origins))
        self.table, at = {}, 0
        for o, z in zip(origins, streams):
            self.table[o] = (at, len(z), False)
            at += len(z)
        self.blob = np.frombuffer(b"".join(streams), dtype=np.uint8)
        self.ratio = a.nbytes / self.blob.size
        if repeat_to and repeat_to > a.shape[0]:              # a long series that repeats the built steps (chunks of one time step only)
            assert ct == 1
            n = a.shape[0]
            for t in range(n, repeat_to):
                for (t0, k, j, i), loc in [(o, self.table[o]) for o in origins if o[0] == t % n]:
                    self.table[(t, k, j, i)] = loc
            self.shape = (repeat_to,) + a.shape[1:]

    def chunk_streams(self):
        return {"chunk": self.chunk, "shuffle": True, "table": self.table, "map": self.blob}

    def __getitem__(self, t):
        import zlib
        from lorenzcycletoolkit_amd.hdf5_lite import _inflate_pool
        ct, ck, cj, ci = self.chunk
        t0 = (int(t) // ct) * ct
        out = np.empty(self.shape[1:], dtype=self.dtype)
        need = [o for o in self.table if o[0] == t0]

        def one(o):
            addr, size, _ = self.table[o]
            raw = zlib.decompress(self.blob[addr: addr + size])
            n = len(raw) // 2
            blk = np.empty((n, 2), dtype=np.uint8)
            np.copyto(blk, np.frombuffer(raw, dtype=np.uint8).reshape(2, n).T)
            blk = blk.reshape(-1).view("<i2").reshape(self.chunk)[int(t) - t0]
            k1, j1, i1 = min(o[1] + ck, self.shape[1]), min(o[2] + cj, self.shape[2]), min(o[3] + ci, self.shape[3])
            out[o[1]: k1, o[2]: j1, o[3]: i1] = blk[: k1 - o[1], : j1 - o[2], : i1 - o[3]]

        list(_inflate_pool().map(one, need))
        return out


def time_now():
    return time.perf_counter()


if __name__ == "__main__":
    main()
