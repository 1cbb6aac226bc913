#!/usr/bin/env python3
"""Host inflate rate of the NetCDF-4 (HDF5 deflate + shuffle) read path against the PCIe link rate.

hdf5_lite._read_chunk does, per chunk: zlib.decompress, un-shuffle (byte transpose), reshape.  This tool builds ERA5-like int16
chunks (a smooth field + noise, quantised like the reference's packed files), deflates them the way netCDF-4 writers do (shuffle, zlib
level 4), and times exactly that work in a thread pool of 1 .. N threads.  No HDF5 library is needed (the GPU box has none).
Prints one JSON line: compression ratio, decoded MB/s per thread count, and what that means against the measured 57 GB/s link.

  python tools/bench_inflate.py [--threads 1,4,8,16,32,64] [--mb 512]
"""
import argparse
import json
import os
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,4,8,16,32,64")
    ap.add_argument("--mb", type=int, default=512, help="decoded megabytes per measurement")
    ap.add_argument("--level", type=int, default=4)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    cy, cx = 361, 720                                   # a quarter of a 721 x 1440 level: the chunking CDS / nccopy typically choose
    lat = np.linspace(-90, 0, cy)[:, None]
    lon = np.linspace(0, 180, cx)[None, :]
    chunks, raw_bytes = [], 0
    n_chunks = max(1, args.mb * (1 << 20) // (cy * cx * 2))
    distinct = min(n_chunks, 64)
    for k in range(distinct):
        f = 250.0 + 40.0 * np.cos(np.deg2rad(lat)) + 5.0 * np.sin(np.deg2rad(3 * lon + 10 * k)) + 1.5 * rng.standard_normal((cy, cx))
        q = np.round((f - 250.0) / (90.0 / 65000.0)).astype("<i2")
        shuffled = np.frombuffer(q.tobytes(), dtype=np.uint8).reshape(-1, 2).T.tobytes()      # HDF5 shuffle filter
        chunks.append(zlib.compress(shuffled, args.level))
        raw_bytes += q.nbytes
    comp_bytes = sum(len(c) for c in chunks)
    work = [chunks[i % distinct] for i in range(n_chunks)]
    decoded = n_chunks * cy * cx * 2

    def inflate(c):                                     # = hdf5_lite.H5File._read_chunk for filters (shuffle, deflate)
        raw = zlib.decompress(c)                        # releases the GIL
        n = len(raw) // 2
        out = np.empty((n, 2), dtype=np.uint8)
        np.copyto(out, np.frombuffer(raw, dtype=np.uint8)[: n * 2].reshape(2, n).T)      # un-shuffle; array assignment releases the GIL too
        return out.view("<i2").reshape(n)

    rates = {}
    for th in [int(x) for x in args.threads.split(",")]:
        if th > (os.cpu_count() or 1):
            continue
        pool = ThreadPoolExecutor(th)
        list(pool.map(inflate, work[: min(len(work), 4 * th)]))          # warm-up
        t0 = time.perf_counter()
        list(pool.map(inflate, work))
        dt = time.perf_counter() - t0
        rates[th] = decoded / dt / 1e6
        pool.shutdown()
    best = max(rates.values())
    print(json.dumps({"chunk": [cy, cx], "dtype": "int16", "filters": f"shuffle + deflate level {args.level}", "compression_ratio": raw_bytes / comp_bytes,
                      "decoded_MB_per_s_by_threads": rates, "cpu_count": os.cpu_count(),
                      "link_GB_per_s_measured": 57.0,
                      "reading": f"best host inflate {best / 1e3:.1f} GB/s of decoded data = {best / 1e3 / 57.0:.2f} of what the link moves; "
                                 "a deflated file is bound by the host's inflate, not by PCIe or the GPU"}))


if __name__ == "__main__":
    main()
