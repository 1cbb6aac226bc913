#!/usr/bin/env python3
"""Scale check of the device ingest (run by hand on a GPU box): writes a ~1.5 GB ERA5-style int16-packed classic NetCDF file
(0.5-degree grid, 37 levels), runs the CLI with and without --device-ingest and compares the CSVs byte for byte.

    python tools/scale_check_ingest.py [nt]
"""
import filecmp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from scipy.io import netcdf_file
    from lorenzcycletoolkit_amd.synthetic import era5_like_levels
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    work = tempfile.mkdtemp(prefix="lec_scale_")
    os.chdir(work)
    os.makedirs("inputs")
    lev = era5_like_levels()[::-1] / 100.0                      # hPa, surface first
    lat = np.arange(90.0, -90.25, -0.5)
    lon = np.arange(0.0, 360.0, 0.5)
    nl, ny, nx = lev.size, lat.size, lon.size
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    f = netcdf_file("era5_style.nc", "w", version=2)
    for n, s in (("time", nt), ("level", nl), ("latitude", ny), ("longitude", nx)):
        f.createDimension(n, s)
    tv = f.createVariable("time", "i", ("time",)); tv[:] = np.arange(nt); tv.units = "hours since 2020-01-01 00:00:00"
    lv = f.createVariable("level", "d", ("level",)); lv[:] = lev; lv.units = "millibars"
    la = f.createVariable("latitude", "f", ("latitude",)); la[:] = lat
    lo = f.createVariable("longitude", "f", ("longitude",)); lo[:] = lon
    p = (lev[None, :, None, None] * 100.0) / 1e5
    base = {"t": 288.0 * p ** 0.19, "u": 20.0 * (1 - p / 1.2), "v": 0 * p, "w": 0 * p, "z": 9.80665 * 7000.0 * np.log(1.0 / p)}
    amp = {"t": 3.0, "u": 6.0, "v": 4.0, "w": 0.2, "z": 300.0}
    for name in base:
        v = f.createVariable(name, "h", ("time", "level", "latitude", "longitude"))
        lo_, hi_ = float(base[name].min() - 6 * amp[name]), float(base[name].max() + 6 * amp[name])
        scale, offset = (hi_ - lo_) / 65000.0, 0.5 * (hi_ + lo_)
        for t in range(nt):
            a = base[name][0][:, None, None] if False else base[name][0] + amp[name] * rng.standard_normal((nl, ny, nx)).astype(np.float32)
            v[t] = np.clip(np.round((a - offset) / scale), -32000, 32000).astype(np.int16)
        v.scale_factor = scale; v.add_offset = offset; v._FillValue = np.int16(-32767)
    f.close()
    print(f"wrote {os.path.getsize('era5_style.nc') / 1e9:.2f} GB in {time.perf_counter() - t0:.1f} s", flush=True)
    open("inputs/namelist", "w").write(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
        "Time;time\nVertical Level;level\n")
    open("inputs/box_limits", "w").write("min_lon;-179.5\nmax_lon;179.5\nmin_lat;-80\nmax_lat;80\n")
    import lorenzcycletoolkit
    for tag, extra in (("host", []), ("device", ["--device-ingest"])):
        t0 = time.perf_counter()
        lorenzcycletoolkit.main(["era5_style.nc", "-r", "-f", "-o", tag] + extra)
        print(f"{tag}: {time.perf_counter() - t0:.1f} s", flush=True)
    out = os.path.join("LEC_Results", "era5_style_fixed")
    same = filecmp.cmp(os.path.join(out, "host.csv"), os.path.join(out, "device.csv"), shallow=False)
    import pandas as pd
    df = pd.read_csv(os.path.join(out, "device.csv"), index_col=0)
    print("identical CSVs:", same, "| finite:", bool(np.isfinite(df.values).all()), "| rows:", len(df))
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
