#!/usr/bin/env python3
"""At ERA5 size: the streamed moving framework with the box-packed series (lec_ingest gathers each step's box) against the same path
with whole-crop cubes (packed=False) -- every number must be the same bit.  Writes the deflated ERA5-size file of tools/bench_cli.py
(needs the image's conda interpreter with h5py), runs a 15-degree track over it both ways, compares scalars, level tables, NaN flags
and the 850-hPa slices kept for the diagnostics.

    python tools/check_packed_at_size.py [--timesteps 48]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONDA = "/opt/conda/bin/python3.9"
NAMELIST = (";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\nEastward Wind Component;u;m/s\n"
            "Northward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\nTime;time\nVertical Level;level\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timesteps", type=int, default=48)
    a = ap.parse_args()
    import numpy as np
    import torch
    from lorenzcycletoolkit_amd import dataset as ds
    from lorenzcycletoolkit_amd import ingest
    T = a.timesteps
    big = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"era5_like_T{T}.nc")
    r = subprocess.run([CONDA, os.path.join(ROOT, "tools", "make_big_nc4.py"), "--out", big, "--timesteps", str(T)], capture_output=True, text=True)
    print(r.stdout.strip()[-200:], r.stderr[-200:], flush=True)
    try:
        with tempfile.TemporaryDirectory() as wd:
            os.chdir(wd)
            os.makedirs("inputs")
            open("inputs/namelist", "w").write(NAMELIST)
            # boxes of several sizes, moving more than a grid point per step
            open("inputs/track", "w").write("time;Lat;Lon;length;width\n" + "".join(
                "2020-01-%02d-%02d00;%.2f;%.2f;%g;%g\n" % (1 + t // 24, t % 24, -35.0 + 0.4 * t, -50.0 + 0.6 * t, 15 + 2.5 * (t % 3), 15 + 5 * (t % 2)) for t in range(T)))
            args = argparse.Namespace(fixed=False, track=True, trackfile="inputs/track", residuals=True, infile=big, cdsapi=False, mpas=False, inflate="auto")
            df = ds.read_namelist("inputs/namelist")
            track = ds.read_track("inputs/track")
            limits = [(lo - w / 2, lo + w / 2, la - ln / 2, la + ln / 2) for la, lo, ln, w in zip(track["Lat"], track["Lon"], track["length"], track["width"])]
            got = {}
            for packed in (True, False):
                st = ingest.prepare_streamed(args, "inputs/namelist")
                stats = {}
                t0 = time.perf_counter()
                res = ingest.lec_streamed(st.raw, st.plan, df, limits, per_step_boxes=True, packed=packed, stats=stats, keep_level=85000.0)
                torch.cuda.synchronize()
                print(f"packed={packed}: {time.perf_counter() - t0:.2f} s, device buffers {stats['device_buffer_bytes'] / 1e9:.1f} GB, domain {stats['domain']}, "
                      f"chunks of {stats['chunk_steps']}", flush=True)
                st.raw.close()
                got[packed] = (res, stats["level_slices"])
            (p, pl), (c, cl) = got[True], got[False]
            same = (torch.equal(p.scalars, c.scalars) and np.array_equal(p.levels.cpu().numpy(), c.levels.cpu().numpy(), equal_nan=True)
                    and torch.equal(p.nanflag, c.nanflag) and all(torch.equal(pl[k], cl[k]) for k in pl))
            print("bit-identical:", same, "finite:", bool(torch.isfinite(p.scalars).all()))
            sys.exit(0 if same else 1)
    finally:
        if os.path.exists(big):
            os.remove(big)


if __name__ == "__main__":
    main()
