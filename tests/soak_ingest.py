#!/usr/bin/env python3
"""Randomised soak of the data preparation (not collected by pytest: run by hand, `python tests/soak_ingest.py --cases 200 --seed 1`).

Every case writes a classic NetCDF file with a random layout -- storage type per variable (int8 / int16 / int32 / float32 / float64), any
combination of scale_factor, add_offset, _FillValue / missing_value (attributes stored as float32 or float64), latitudes N -> S or
S -> N, longitudes 0..360 or -180..180, levels in hPa / millibars / Pa in either order with or without levels above 10 hPa, a time
axis in hours / minutes / days / seconds -- and a random box, then compares, element for element and dtype for dtype,
  * the package's host preparation (dataset.prepare_data)                      -- runs anywhere,
  * the device decode (ingest.device_cube = lec_ingest) when a GPU is present,
with the oracle's restatement of the reference's decode + process_data + slice_domain (oracle/cf_decode.py; the oracle is test
infrastructure, hence this file lives under tests/).  Prints one line per failure and a summary; exit code 1 if anything failed."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lorenzcycletoolkit_amd import dataset as ds  # noqa: E402
from oracle import cf_decode as cf  # noqa: E402

NAMES = {"tair": "t", "u": "u", "v": "v", "omega": "w", "geo": "z", "lat": "latitude", "lon": "longitude", "level": "level", "time": "time"}
NAMELIST = (";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
            "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
            "Time;time\nVertical Level;level\n")
CODES = {"b": np.int8, "h": np.int16, "i": np.int32, "f": np.float32, "d": np.float64}


def write_case(rng, path, fill_rate=0.002):
    from scipy.io import netcdf_file
    nt = int(rng.integers(2, 6))
    dlon = float(rng.choice([2.5, 5.0, 1.0]))
    nx = int(rng.integers(12, 40))
    west = float(rng.choice([0.0, 180.0, 300.0 - dlon * nx, -170.0, -60.0]))
    if west + dlon * nx > 360.0:
        west = 360.0 - dlon * nx
    lon = west + dlon * np.arange(nx)
    ny = int(rng.integers(6, 20))
    lat = -50.0 + float(rng.choice([2.5, 1.0])) * np.arange(ny)
    if rng.random() < 0.5:
        lat = lat[::-1]
    hpa = np.array([1000, 925, 850, 700, 500, 300, 200, 100, 50, 10, 5, 1][:int(rng.integers(4, 13))], dtype=np.float64)
    if rng.random() < 0.5:
        hpa = hpa[::-1]
    unit = str(rng.choice(["millibars", "hPa", "Pa", "mbar"]))
    lev = hpa * (100.0 if unit == "Pa" else 1.0)
    tunit, tstep = [("hours", 6), ("minutes", 360), ("days", 1), ("seconds", 21600)][int(rng.integers(0, 4))]
    nl = lev.size
    f = netcdf_file(path, "w", version=2)
    for n, s in (("time", nt), ("level", nl), ("latitude", ny), ("longitude", nx)):
        f.createDimension(n, s)
    tv = f.createVariable("time", "i", ("time",)); tv[:] = tstep * np.arange(nt); tv.units = f"{tunit} since 2020-01-01 00:00:00"
    lcode = "i" if rng.random() < 0.5 else "f"
    lv = f.createVariable("level", lcode, ("level",)); lv[:] = lev; lv.units = unit
    ccode = "f" if rng.random() < 0.6 else "d"
    la = f.createVariable("latitude", ccode, ("latitude",)); la[:] = lat
    lo = f.createVariable("longitude", ccode, ("longitude",)); lo[:] = lon
    p = np.where(hpa >= 10, hpa, 10.0)[None, :, None, None] / 1000.0
    shape = (nt, nl, ny, nx)
    fields = {
        "t": 288.0 * p ** 0.19 + 8.0 * np.cos(np.deg2rad(2 * lat))[None, None, :, None] * p + rng.standard_normal(shape),
        "u": 20.0 * np.cos(np.deg2rad(lat))[None, None, :, None] * (1 - p / 1.2) + 5 * rng.standard_normal(shape),
        "v": 3.0 * rng.standard_normal(shape),
        "w": 0.1 * rng.standard_normal(shape),
        "z": 9.80665 * 7000.0 * np.log(1.0 / p) + 100.0 * rng.standard_normal(shape),
    }
    layout = {}
    for name, a in fields.items():
        code = str(rng.choice(["h", "h", "i", "f", "d", "b"]))
        integer = code in "bhi"
        has_scale = integer or rng.random() < 0.15
        has_offset = has_scale and rng.random() < 0.6
        fill_kind = str(rng.choice(["none", "_FillValue", "missing_value", "both"]))
        attr_t = np.float32 if rng.random() < 0.3 else np.float64
        lo_, hi_ = float(a.min()), float(a.max())
        if integer:
            top = {"b": 120, "h": 32000, "i": 2000000000}[code]
            off = 0.5 * (hi_ + lo_) if has_offset else 0.0
            scale = (hi_ - lo_) / (1.9 * top) if has_offset else max(abs(hi_), abs(lo_)) / top
            scale, off = float(attr_t(scale)), float(attr_t(off))
            q = np.clip(np.round((a - off) / scale), -top, top).astype(CODES[code])
            fv = CODES[code]({"b": -127, "h": -32767, "i": -2147483647}[code])
        else:
            scale, off = (float(attr_t(rng.choice([0.5, 0.01, 3.0]))), float(attr_t(rng.choice([0.0, 273.15, -12.5])))) if has_scale else (1.0, 0.0)
            q = ((a - (off if has_offset else 0.0)) / scale).astype(CODES[code])
            fv = CODES[code](rng.choice([9.96921e36, -9999.0, 1e20]))
        if fill_kind != "none":
            m = rng.random(shape) < fill_rate
            if rng.random() < 0.3:
                m[int(rng.integers(0, nt)), int(rng.integers(0, nl))] = True        # a whole level of one step
            q[m] = fv
        v = f.createVariable(name, code, ("time", "level", "latitude", "longitude"))
        v[:] = q
        if has_scale:
            v.scale_factor = attr_t(scale)
        if has_offset:
            v.add_offset = attr_t(off)
        if fill_kind in ("_FillValue", "both"):
            v._FillValue = fv
        if fill_kind in ("missing_value", "both"):
            v.missing_value = fv
        layout[name] = f"{code}{'s' if has_scale else ''}{'o' if has_offset else ''}:{fill_kind}:{attr_t.__name__[-2:]}"
    f.close()
    # a box inside the longitudes / latitudes the file has AFTER the reference's wrap to (-180, 180]
    wl = np.sort((lon + 180) % 360 - 180) if (lon.min() < -180 or lon.max() > 180) else np.sort(lon)
    sl = np.sort(lat)
    i0, j0 = int(rng.integers(0, nx - 3)), int(rng.integers(0, ny - 3))
    i1, j1 = int(rng.integers(i0 + 2, nx)), int(rng.integers(j0 + 2, ny))
    jit = lambda: float(rng.uniform(-0.4, 0.4))
    limits = tuple(float(x) for x in (wl[i0] + jit(), wl[i1] + jit(), sl[j0] + jit(), sl[j1] + jit()))
    what = (f"nt={nt} nl={nl} {ny}x{nx} lon {lon[0]:g}..{lon[-1]:g} lat {lat[0]:g}..{lat[-1]:g} levels {hpa[0]:g}..{hpa[-1]:g} {unit}({lcode}) "
            f"time {tunit} coords {ccode} vars {layout} box {tuple(round(x, 2) for x in limits)}")
    return limits, what


def one_case(rng, case, tmp, gpu):
    path = os.path.join(tmp, f"case{case}.nc")
    limits, what = write_case(rng, path)
    what = f"case {case}: {what}"
    with open(os.path.join(tmp, "inputs", "box_limits"), "w") as fh:
        fh.write("min_lon;%r\nmax_lon;%r\nmin_lat;%r\nmax_lat;%r\n" % (limits[0], limits[1], limits[2], limits[3]))
    fails = []
    try:
        ref = cf.prepare(path, NAMES, fixed_limits=limits)
    except Exception as e:
        os.remove(path)
        return [f"{what}: the ORACLE raised {e!r}"]
    args = argparse.Namespace(infile=path, fixed=True, track=False, trackfile=None, cdsapi=False)
    try:
        got = ds.prepare_data(args, "inputs/namelist")
        for k in ("lat", "lon", "level", "time_s"):
            x, y = np.asarray(getattr(got, k)), getattr(ref, k)
            if not np.array_equal(x, y):
                fails.append(f"{what}: host {k} differs: {x[:4]} .. vs {y[:4]} ..")
        for role, name in (("tair", "t"), ("u", "u"), ("v", "v"), ("omega", "w"), ("geopt", "z")):
            x, y = got.variables[name], getattr(ref, role)
            if x.dtype != y.dtype:
                fails.append(f"{what}: host {name} is {x.dtype}, the oracle decodes {y.dtype}")
            elif x.shape != y.shape or not np.array_equal(x, y, equal_nan=True):
                fails.append(f"{what}: host {name} differs in {int((~((x == y) | (np.isnan(x) & np.isnan(y)))).sum()) if x.shape == y.shape else 'shape'} elements")
    except Exception as e:
        fails.append(f"{what}: host preparation raised {e!r}")
    if gpu:
        import torch
        from lorenzcycletoolkit_amd import ingest
        try:
            df = ds.read_namelist("inputs/namelist")
            raw = ds.open_raw(path, df)
            plan = ingest.make_plan(raw, argparse.Namespace(fixed=True, track=False, trackfile=None))
            for k in ("lat", "lon", "level", "time_s"):
                if not np.array_equal(np.asarray(getattr(plan, k)), getattr(ref, k)):
                    fails.append(f"{what}: plan {k} differs")
            for role, name in (("tair", "t"), ("u", "u"), ("v", "v"), ("omega", "w"), ("geopt", "z")):
                y = getattr(ref, role)
                cube = ingest.device_cube(raw.variables[name], plan)
                x = cube.cpu().numpy()
                if str(x.dtype) != str(y.dtype):
                    fails.append(f"{what}: device {name} is {x.dtype}, the oracle decodes {y.dtype}")
                elif x.shape != y.shape or not np.array_equal(x, y, equal_nan=True):
                    fails.append(f"{what}: device {name} differs")
                wide = ingest.device_cube(raw.variables[name], plan, out_dtype=np.float64).cpu().numpy()
                if not np.array_equal(wide, y.astype(np.float64), equal_nan=True):
                    fails.append(f"{what}: device {name} widened to float64 differs")
            raw.close()
            torch.cuda.synchronize()
        except Exception as e:
            fails.append(f"{what}: device ingest raised {e!r}")
    os.remove(path)
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    gpu = False
    if not a.no_gpu:
        import torch
        gpu = torch.cuda.is_available()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    fails = []
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "inputs"))
        with open(os.path.join(tmp, "inputs", "namelist"), "w") as fh:
            fh.write(NAMELIST)
        os.chdir(tmp)
        for c in range(a.cases):
            fails += one_case(rng, c, tmp, gpu)
            if (c + 1) % 25 == 0:
                print(f"{c + 1} cases, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
        os.chdir(ROOT)
    for ln in fails:
        print("FAIL", ln)
    print(f"ingest soak ({'host + device' if gpu else 'host only'}): {a.cases} cases, seed {a.seed}: {len(fails)} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
