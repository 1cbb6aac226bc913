#!/usr/bin/env python3
"""Randomised soak of the STREAMED pipeline on the GPU box (not collected by pytest: `python tests/soak_streamed.py --cases 150 --seed 1`).

Every case is a random classic NetCDF file of tests/soak_ingest.py (random storage types, packing, fill values -> NaN levels, axis orders,
units) and a random box; then
  * the resident framework run (host preparation -> BoxData) and ``ingest.lec_streamed`` with a random chunk length, staged or from
    registered file memory, must give the same bits (scalars, level tables, NaN flags),
  * time ranges of the streamed run (what the ranks of a sharded run compute, NaN-level mask merged by hand) must reproduce the whole,
  * the terms must agree with the oracle LEC evaluated on the ORACLE's preparation of the file (1e-9 of scale; float32-decoded files:
    the engine stores float32 and computes in fp64, the oracle computes in fp64 on the same float32 values).
Prints one line per failure and a summary; exit code 1 if anything failed."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lorenzcycletoolkit_amd import dataset as ds  # noqa: E402
from lorenzcycletoolkit_amd import ingest  # noqa: E402
from lorenzcycletoolkit_amd.frameworks import BoxData  # noqa: E402
from oracle import cf_decode as cf  # noqa: E402
from oracle import lec_oracle as o  # noqa: E402
from tests import soak_ingest as si  # noqa: E402
from tests.helpers import SCALARS, as_f64, scale_err  # noqa: E402


FILL_RATE = 0.0


def same(x, y):
    return x.shape == y.shape and bool(((x == y) | (torch.isnan(x) & torch.isnan(y))).all())


def one_case(rng, case, tmp):
    path = os.path.join(tmp, f"case{case}.nc")
    limits, what = si.write_case(rng, path, fill_rate=FILL_RATE)  # default 0: fill values mark whole levels of single steps only (NaN levels)
    what = f"case {case}: {what}"
    with open(os.path.join(tmp, "inputs", "box_limits"), "w") as fh:
        fh.write("min_lon;%r\nmax_lon;%r\nmin_lat;%r\nmax_lat;%r\n" % limits)
    fails = []
    args = argparse.Namespace(fixed=True, track=False, trackfile=None, residuals=True)
    df = ds.read_namelist("inputs/namelist")
    try:
        host = ds.slice_domain(ds.process_data(ds.open_dataset(path, df), args, df), args, df)
        if host.lat.size < 3 or host.lon.size < 3 or host.level.size < 2:
            os.remove(path)
            return []
        box = BoxData(host, df, *limits, args=args)
        res = box.result
        raw = ds.open_raw(path, df)
        plan = ingest.make_plan(raw, args)
        nt = len(plan.tsel)
        chunk = int(rng.integers(1, nt + 2))
        staging = str(rng.choice(["staged", "auto"]))
        stats = {}
        st = ingest.lec_streamed(raw, plan, df, [limits], chunk_steps=chunk, staging=staging, stats=stats)
        torch.cuda.synchronize()
        how = f"chunk {chunk} {stats['staging']}"
        if not (same(st.scalars, res.scalars) and same(st.levels, res.levels) and torch.equal(st.nanflag, res.nanflag)):
            fails.append(f"{what}: streamed ({how}) differs from the resident run")
        # time ranges, as the ranks of a sharded run take them: the any-time NaN-level mask is merged across the parts first
        if nt >= 2:
            cut = int(rng.integers(1, nt))
            parts, masks = [], []
            collect = lambda m: (masks.append(m.clone()), m)[1]
            for (a, b) in ((0, cut), (cut, nt)):
                ingest.lec_streamed(raw, plan, df, [limits], chunk_steps=chunk, t_range=(a, b), merge_dropmask=collect)
            merged = torch.stack(masks).amax(0)
            for (a, b) in ((0, cut), (cut, nt)):
                parts.append(ingest.lec_streamed(raw, plan, df, [limits], chunk_steps=chunk, t_range=(a, b), merge_dropmask=lambda m: m.copy_(merged)))
            sc = torch.cat([p.scalars for p in parts]); lv = torch.cat([p.levels for p in parts])
            if not (same(sc, res.scalars) and same(lv, res.levels)):
                fails.append(f"{what}: time ranges [0, {cut}) + [{cut}, {nt}) of the streamed run ({how}) differ from the whole")
        raw.close()
        # the oracle, on its own preparation of the file
        dom = cf.prepare(path, si.NAMES, fixed_limits=limits)
        try:
            with np.errstate(all="ignore"):
                ref_s, _ = o.lec_fixed(as_f64(dom), *limits)
        except IndexError:           # a boundary term with no level left: the reference stops at .isel(level=-1) of an empty array
            ref_s = {}
        got = res.scalars_dict()
        for name in SCALARS:
            if name in ref_s:
                e = scale_err(got[name], ref_s[name])
                if not e <= 1e-9:
                    fails.append(f"{what}: {name} off the oracle by {e:.3e}")
    except Exception as e:
        import traceback
        fails.append(f"{what}: raised {e!r} at {traceback.format_exc().splitlines()[-3].strip()}")
    os.remove(path)
    return fails


def one_track_case(rng, case, tmp):
    """The moving framework: a random track over a random selection of the file's time steps, boxes of random width / length; the
    host-prepared resident BoxData, the streamed one (random chunk length) and the oracle's lec_moving on ITS preparation (track-time
    selection and track-extent crop included)."""
    path = os.path.join(tmp, f"track{case}.nc")
    _limits, what = si.write_case(rng, path, fill_rate=0.0)
    what = f"track case {case}: {what.split(' box ')[0]}"
    fails = []
    df = ds.read_namelist("inputs/namelist")
    try:
        probe = ds.open_raw(path, df)
        times, lat, lon = probe.time, np.sort(probe.lat), np.sort(((probe.lon + 180) % 360 - 180) if (probe.lon.min() < -180 or probe.lon.max() > 180) else probe.lon)
        probe.close()
        nt = len(times)
        if nt < 2:
            os.remove(path)
            return []
        keep = np.sort(rng.choice(nt, size=int(rng.integers(2, nt + 1)), replace=False))
        dlat, dlon = float(np.diff(lat).min()), float(np.diff(lon).min())
        w = float(rng.choice([3, 4, 6]) * dlon + rng.uniform(0, 0.3))
        ln = float(rng.choice([3, 4, 5]) * dlat + rng.uniform(0, 0.3))
        if lon[-1] - lon[0] < w + 2 * dlon or lat[-1] - lat[0] < ln + 2 * dlat or np.diff(lon).max() > 1.5 * dlon:      # (a wrapped axis with a gap)
            os.remove(path)
            return []
        clat = rng.uniform(lat[0] + ln / 2 + dlat, lat[-1] - ln / 2 - dlat, size=keep.size)
        clon = rng.uniform(lon[0] + w / 2 + dlon, lon[-1] - w / 2 - dlon, size=keep.size)
        with open(os.path.join(tmp, "inputs", "track"), "w") as fh:
            fh.write("time;Lat;Lon;width;length\n")
            for t, la, lo in zip(keep, clat, clon):
                fh.write("%s;%r;%r;%r;%r\n" % (str(times[t].astype("datetime64[m]")).replace("T", "-").replace(":", ""), float(la), float(lo), w, ln))
        args = argparse.Namespace(fixed=False, track=True, trackfile="inputs/track", residuals=True, infile=path, cdsapi=False)
        host = ds.prepare_data(args, "inputs/namelist")
        track = ds.read_track("inputs/track")
        limits = [(lo - w / 2, lo + w / 2, la - ln / 2, la + ln / 2) for la, lo in zip(track["Lat"], track["Lon"])]
        a = BoxData(host, df, args=args, boxes_limits=limits).result
        chunk = int(rng.integers(1, keep.size + 2))
        st = ingest.prepare_streamed(args, "inputs/namelist", chunk_steps=chunk)
        b = BoxData(st, df, args=args, boxes_limits=limits).result
        st.raw.close()
        if not (same(a.scalars, b.scalars) and same(a.levels, b.levels)):
            fails.append(f"{what}: streamed (chunk {chunk}) differs from the resident run, steps {keep.tolist()}")
        tt = track.index.values.astype("datetime64[ns]")
        dom = cf.prepare(path, si.NAMES, track=(tt, track["Lat"].values, track["Lon"].values), max_width=w, max_length=ln)
        try:
            with np.errstate(all="ignore"):
                ref_s, _ = o.lec_moving(as_f64(dom), limits)
        except IndexError:
            ref_s = {}
        got = a.scalars_dict()
        for name in SCALARS:
            if name in ref_s:
                e = scale_err(got[name], ref_s[name])
                if not e <= 1e-9:
                    fails.append(f"{what}: {name} off the oracle by {e:.3e} (steps {keep.tolist()}, box {w:.2f} x {ln:.2f})")
    except Exception as e:
        import traceback
        fails.append(f"{what}: raised {e!r} at {traceback.format_exc().splitlines()[-3].strip()}")
    os.remove(path)
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--fill-rate", type=float, default=0.0, help="fraction of scattered fill values (then nearly every level is NaN somewhere)")
    a = ap.parse_args()
    global FILL_RATE
    FILL_RATE = a.fill_rate
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    fails = []
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "inputs"))
        with open(os.path.join(tmp, "inputs", "namelist"), "w") as fh:
            fh.write(si.NAMELIST)
        os.chdir(tmp)
        for c in range(a.cases):
            fails += one_case(rng, c, tmp) if c % 3 else one_track_case(rng, c, tmp)
            if (c + 1) % 25 == 0:
                print(f"{c + 1} cases, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
        os.chdir(ROOT)
    for ln in fails[:40]:
        print("FAIL", ln[:900])
    print(f"streamed soak: {a.cases} cases, seed {a.seed}: {len(fails)} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
