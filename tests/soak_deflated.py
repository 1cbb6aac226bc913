#!/usr/bin/env python3
"""Randomised soak of the streamed pipeline on NetCDF-4 / HDF5 files (GPU box, by hand: `python tests/soak_deflated.py --cases 60 --seed 1`;
needs the image's conda interpreter with h5py for the writer, tools/classic_to_nc4.py).

Every case is a random classic NetCDF file of tests/soak_ingest.py (storage types, packing, fill values, axis orders, units) AND the same
stored values rewritten as NetCDF-4 in a random layout (chunk shapes that do not divide the extents, shuffle, deflate, fletcher32, some
variables uncompressed or contiguous: mixed stagers in one run).  The classic file's resident run is the reference; the NetCDF-4 file must
give the same BITS through
  * the host preparation (hdf5_lite inflates on the host),
  * ``lec_streamed`` with a random chunk length, the compressed chunks copied from the registered file pages or through pinned staging,
    inflated on the device or on the host,
  * two time ranges with the NaN-level mask merged by hand (what two ranks compute),
  * for every third case a track (one box per step; the 850-hPa slices kept from the streamed pass) instead of the fixed box."""
import argparse
import os
import subprocess
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
from tests import soak_ingest as si  # noqa: E402

CONDA = "/opt/conda/bin/python3.9"
SEEN = {"checked": 0, "moving": 0, "streamed_runs": 0, "device_inflate": 0, "registered": 0, "staged": 0, "host_inflate": 0, "skipped_small": 0}


def same(x, y):
    return x.shape == y.shape and bool(((x == y) | (torch.isnan(x) & torch.isnan(y))).all())


def check(rng, case, classic, nc4, limits, what, layout):
    fails = []
    what = f"case {case}: {what} || {layout}"
    args = argparse.Namespace(fixed=True, track=False, trackfile=None, residuals=True, infile=nc4, cdsapi=False, mpas=False)
    df = ds.read_namelist("inputs/namelist")
    with open(os.path.join("inputs", "box_limits"), "w") as fh:
        fh.write("min_lon;%r\nmax_lon;%r\nmin_lat;%r\nmax_lat;%r\n" % limits)
    try:
        host = ds.slice_domain(ds.process_data(ds.open_dataset(classic, df), args, df), args, df)
        if host.lat.size < 3 or host.lon.size < 3 or host.level.size < 2:
            SEEN["skipped_small"] += 1
            return []
        SEEN["checked"] += 1
        moving = case % 3 == 0 and 85000.0 in host.level and len(host.time) >= 2
        if moving:
            nt = len(host.time)
            lo0, lo1, la0, la1 = limits
            boxes = [(lo0 + 0.3 * t, lo1 + 0.3 * t, la0, la1) for t in range(nt)]
            boxes = [b for b in boxes if b[1] <= host.lon[-1]] or [limits]
            boxes = (boxes + [boxes[-1]] * nt)[:nt]
            ref = BoxData(host, df, args=args, boxes_limits=boxes).result
        else:
            boxes = [limits]
            ref = BoxData(host, df, *limits, args=args).result
        # the NetCDF-4 file through the host preparation
        data4 = ds.prepare_data(args, "inputs/namelist")
        r4 = (BoxData(data4, df, args=args, boxes_limits=boxes) if moving else BoxData(data4, df, *limits, args=args)).result
        if not (same(r4.scalars, ref.scalars) and same(r4.levels, ref.levels) and torch.equal(r4.nanflag, ref.nanflag)):
            fails.append(f"{what}: NetCDF-4 host preparation differs from the classic file's run")
        raw = ds.open_raw(nc4, df)
        plan = ingest.make_plan(raw, args)
        nt = len(plan.tsel)
        for _ in range(2):
            chunk = int(rng.integers(1, nt + 2))
            staging = str(rng.choice(["staged", "auto"]))
            inflate = str(rng.choice(["auto", "auto", "host"]))
            slots = int(rng.integers(2, 4))
            stats = {}
            keep = 85000.0 if moving else None
            st = ingest.lec_streamed(raw, plan, df, boxes, per_step_boxes=moving, chunk_steps=chunk, staging=staging, inflate=inflate, slots=slots,
                                     stats=stats, keep_level=keep)
            torch.cuda.synchronize()
            SEEN["streamed_runs"] += 1
            SEEN["moving"] += int(moving)
            SEEN["device_inflate"] += int(stats["inflate"] == "device")
            SEEN["host_inflate"] += int(stats["inflate"] == "host")
            SEEN[stats["staging"]] += 1
            how = f"chunk {chunk} staging {staging}->{stats['staging']} inflate {inflate}->{stats['inflate']} slots {slots}"
            if not (same(st.scalars, ref.scalars) and same(st.levels, ref.levels) and torch.equal(st.nanflag, ref.nanflag)):
                fails.append(f"{what}: streamed ({how}) differs from the classic file's resident run")
            if moving:
                k = int(np.flatnonzero(plan.level == 85000.0)[0])
                for key, name in (("u", "u"), ("v", "v"), ("geopt", "z")):
                    want = torch.as_tensor(np.ascontiguousarray(host.variables[name][:, k])).to(stats["level_slices"][key].device)
                    if not same(stats["level_slices"][key].to(want.dtype), want):
                        fails.append(f"{what}: kept 850-hPa slice of {name} ({how}) differs from the host-decoded one")
        if nt >= 2 and not moving:
            cut = int(rng.integers(1, nt))
            masks = []
            collect = lambda m: (masks.append(m.clone()), m)[1]
            for (a, b) in ((0, cut), (cut, nt)):
                ingest.lec_streamed(raw, plan, df, boxes, chunk_steps=chunk, t_range=(a, b), merge_dropmask=collect)
            merged = torch.stack(masks).amax(0)
            parts = [ingest.lec_streamed(raw, plan, df, boxes, chunk_steps=chunk, t_range=(a, b), merge_dropmask=lambda m: m.copy_(merged))
                     for (a, b) in ((0, cut), (cut, nt))]
            if not (same(torch.cat([p.scalars for p in parts]), ref.scalars) and same(torch.cat([p.levels for p in parts]), ref.levels)):
                fails.append(f"{what}: time ranges [0, {cut}) + [{cut}, {nt}) differ from the whole")
        raw.close()
    except Exception as e:
        import traceback
        fails.append(f"{what}: raised {e!r} at {' | '.join(x.strip() for x in traceback.format_exc().splitlines()[-4:-1])}")
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    if not os.path.exists(CONDA):
        sys.exit(f"needs {CONDA} with h5py for the writer")
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    fails = []
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "inputs"))
        with open(os.path.join(tmp, "inputs", "namelist"), "w") as fh:
            fh.write(si.NAMELIST)
        os.chdir(tmp)
        meta, pairs = [], []
        for c in range(a.cases):
            classic, nc4 = os.path.join(tmp, f"case{c}.nc"), os.path.join(tmp, f"case{c}_4.nc")
            limits, what = si.write_case(rng, classic, fill_rate=0.0)
            meta.append((classic, nc4, limits, what))
            pairs += [classic, nc4]
        r = subprocess.run([CONDA, os.path.join(ROOT, "tools", "classic_to_nc4.py"), str(a.seed)] + pairs, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit("writer failed: " + r.stderr[-2000:])
        layouts = dict(ln.split(" | ", 1) for ln in r.stdout.splitlines() if " | " in ln)
        for c, (classic, nc4, limits, what) in enumerate(meta):
            fails += check(rng, c, classic, nc4, limits, what, layouts.get(nc4, "?"))
            if (c + 1) % 20 == 0:
                print(f"{c + 1} cases, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
        os.chdir(ROOT)
    for ln in fails[:30]:
        print("FAIL", ln[:1200])
    print(f"deflated soak: {a.cases} cases, seed {a.seed}: {len(fails)} failures; exercised: {SEEN}")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
