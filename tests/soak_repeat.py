#!/usr/bin/env python3
"""Repeatability soak (GPU box, by hand: `python tests/soak_repeat.py --seconds 120`): the same inputs, again and again, must give the
same bits -- a race or an uninitialised read shows as a difference sooner or later.  Four loops share the time: the headline
configuration (all terms, one fixed box, fp64 and fp32 storage), per-step boxes, the streamed ingest of a deflated NetCDF-4 fixture
(device inflate, three pipeline slots, one stream per variable and slot), and lec_inflate alone on 2000 streams."""
import argparse
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lorenzcycletoolkit_amd import dataset as ds  # noqa: E402
from lorenzcycletoolkit_amd import ingest  # noqa: E402
from lorenzcycletoolkit_amd.engine import LECEngine  # noqa: E402
from tests.helpers import synthetic_domain  # noqa: E402
from tests.test_gpu_inflate import inflate, payload  # noqa: E402

DEV = "cuda:0"


def same(x, y):
    return bool(((x == y) | (torch.isnan(x) & torch.isnan(y))).all())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    a = ap.parse_args()
    share = a.seconds / 4
    fails = []
    # 1. resident engine: fixed box, both storage dtypes; per-step boxes
    for dtype, moving in ((np.float64, False), (np.float32, False), (np.float64, True)):
        dom = synthetic_domain(12, 17, 181, 360, seed=3, dtype=dtype, lat0=-80, lat1=80, lon0=-180, lon1=179)
        eng = LECEngine(dom.lat, dom.lon, dom.level, device=DEV)
        f = [torch.as_tensor(np.ascontiguousarray(x)).to(DEV) for x in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
        boxes = [(10 + 3 * t, 200 + 5 * t, 20 + t, 120 + 2 * t) for t in range(12)] if moving else [(0, 359, 0, 180)]
        ref = eng.compute(*f, boxes, time_s=dom.time_s, per_step_boxes=moving)
        t0, n = time.time(), 0
        while time.time() - t0 < share / 3:
            r = eng.compute(*f, boxes, time_s=dom.time_s, per_step_boxes=moving)
            n += 1
            if not (same(r.scalars, ref.scalars) and same(r.levels, ref.levels)):
                fails.append(f"resident {np.dtype(dtype).name} moving={moving}: pass {n} differs")
                break
        print(f"resident {np.dtype(dtype).name} moving={moving}: {n} passes", flush=True)
    # 2. streamed ingest of a deflated fixture
    tmp = os.path.join("/tmp", f"soak_repeat_{os.getpid()}")
    os.makedirs(os.path.join(tmp, "inputs"), exist_ok=True)
    os.chdir(tmp)
    open("inputs/namelist", "w").write(";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
                                       "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
                                       "Time;time\nVertical Level;level\n")
    open("inputs/box_limits", "w").write("min_lon;-170\nmax_lon;170\nmin_lat;-60\nmax_lat;60\n")
    import argparse as ap_
    args = ap_.Namespace(fixed=True, track=False, trackfile=None)
    df = ds.read_namelist("inputs/namelist")
    for name in ("packed_interleaved_v18.nc", "packed_unlimited_latest.nc"):
        raw = ds.open_raw(os.path.join(ROOT, "tests", "golden", "hdf5", name), df)
        plan = ingest.make_plan(raw, args)
        lim = [(-170.0, 170.0, -60.0, 60.0)]
        ref = ingest.lec_streamed(raw, plan, df, lim, chunk_steps=1)
        t0, n = time.time(), 0
        while time.time() - t0 < share / 2:
            r = ingest.lec_streamed(raw, plan, df, lim, chunk_steps=1 + n % 3)
            n += 1
            if not (same(r.scalars, ref.scalars) and same(r.levels, ref.levels)):
                fails.append(f"streamed {name}: pass {n} differs")
                break
        raw.close()
        print(f"streamed {name}: {n} passes", flush=True)
    # 3. lec_inflate alone
    rng = np.random.default_rng(1)
    data = [payload(rng, int(rng.integers(1000, 200000)), k % 6) for k in range(2000)]
    streams = [zlib.compress(d, int(rng.integers(1, 10))) for d in data]
    t0, n = time.time(), 0
    while time.time() - t0 < share:
        got, status, _, _ = inflate(streams, [len(d) for d in data], packed=bool(n % 2), flags=2 * ((n // 2) % 2))
        n += 1
        if not (status[:, 0] == 0).all() or any(g != d for g, d in zip(got, data)):
            fails.append(f"lec_inflate: pass {n} differs")
            break
    print(f"lec_inflate: {n} passes of 2000 streams", flush=True)
    for ln in fails:
        print("FAIL", ln)
    print(f"repeatability soak: {len(fails)} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
