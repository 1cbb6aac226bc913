#!/usr/bin/env python3
"""Randomised parity soak on the GPU box (not collected by pytest: run by hand, `python tests/soak_gpu.py --cases 300 --seed 1`).

Every case draws a grid, a storage dtype, uniform or stretched longitudes, fixed or per-step boxes anywhere in the grid (edges and
one-vector-wide boxes included), an even or uneven time axis and, now and then, NaN patches; then
  * every kernel family that can run the case must agree record by record (row_sweep vs two_sweep vs row_block / box_tile),
  * the terms must agree with the oracle (1e-9 of scale) -- the oracle is test infrastructure, hence this file lives under tests/,
  * shards of the series must reproduce the whole bit for bit.
Prints one line per failure and a summary; exit code 1 if anything failed."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lorenzcycletoolkit_amd.engine import LECEngine  # noqa: E402
from oracle import lec_oracle as o  # noqa: E402
from tests.helpers import SCALARS, as_f64, scale_err, synthetic_domain  # noqa: E402

DEV = "cuda:0"


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def rows_close(a, b):
    """max over the 28 statistics of |a - b| relative to the statistic's largest finite value; NaNs must sit at the same places"""
    ra, rb = a.rows[..., :28], b.rows[..., :28]
    if bool((torch.isnan(ra) != torch.isnan(rb)).any()):
        return float("inf")
    z = torch.zeros_like(rb)
    fin = torch.where(torch.isnan(rb), z, rb.abs())
    den = fin.amax(dim=(0, 1, 2)).clamp_min(1e-300)
    d = torch.where(torch.isnan(rb), z, (ra - rb).abs())
    return float((d.amax(dim=(0, 1, 2)) / den).max())


def same_bits(x, y):
    """torch.equal with NaN == NaN"""
    return x.shape == y.shape and bool(((x == y) | (torch.isnan(x) & torch.isnan(y))).all())


def one_case(rng, case, with_oracle):
    nt = int(rng.integers(2, 7))
    nl = int(rng.choice([2, 3, 5, 8, 13, 22, 37]))
    ny = int(rng.integers(4, 70))
    nx = int(rng.choice([8, 16, 31, 64, 65, 96, 127, 128, 129, 200, 256, 257, 300]))
    dtype = np.float32 if rng.random() < 0.4 else np.float64
    nonuni = bool(rng.random() < 0.25)
    moving = bool(rng.random() < 0.45)
    dom = synthetic_domain(nt, nl, ny, nx, seed=int(rng.integers(1 << 30)), dtype=dtype, nonuniform_lon=nonuni,
                           dt_s=float(rng.choice([3600.0, 21600.0])))
    if rng.random() < 0.4:
        dom.time_s = np.cumsum(rng.integers(1, 4, nt) * 3600.0)
    nan_case = bool(rng.random() < 0.15) and not moving
    if nan_case:
        k = int(rng.integers(0, nl))
        j0, i0 = int(rng.integers(0, ny - 1)), int(rng.integers(0, nx - 2))
        dom.omega[int(rng.integers(0, nt)), k, j0:j0 + 2, i0:i0 + 3] = np.nan

    def box():
        wx, wy = int(rng.integers(2, nx + 1)), int(rng.integers(2, ny + 1))
        if rng.random() < 0.3:
            wx = min(nx, int(rng.choice([2, 3, 4, 5, 8, 63, 64, 65, 128])))
        iw, js = int(rng.integers(0, nx - wx + 1)), int(rng.integers(0, ny - wy + 1))
        if rng.random() < 0.3:
            iw = 0 if rng.random() < 0.5 else nx - wx
        return (iw, iw + wx - 1, js, js + wy - 1)

    boxes = [box() for _ in range(nt)] if moving else [box()]
    eng = LECEngine(dom.lat, dom.lon, dom.level, device=DEV)
    f = [dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    kw = dict(time_s=dom.time_s, keep_rows=True, per_step_boxes=moving)
    what = f"case {case}: nt={nt} nl={nl} {ny}x{nx} {np.dtype(dtype).name} nonuni={nonuni} moving={moving} nan={nan_case} boxes={boxes}"
    fails = []
    auto = eng.compute(*f, boxes, **kw)
    fams = ["row_sweep", "two_sweep"] + (["box_tile"] if moving else ["row_block", "box_tile"])
    for fam in fams:
        try:
            r = eng.compute(*f, boxes, tuning={"kernel": fam}, **kw)
        except Exception as e:      # two_sweep has a row-length limit only far above these sizes: anything raised is a failure
            fails.append(f"{what}: {fam} raised {e!r}")
            continue
        err = rows_close(r, auto)
        if not err <= 2e-11:
            fails.append(f"{what}: {fam} differs from the default family by {err:.3e}")
    if moving:      # time groups of the box-tile kernel (waves of consecutive steps share T through LDS on the union of their boxes): no bit may move
        for tg in (2, 4):
            r = eng.compute(*f, boxes, tuning={"kernel": "box_tile", "block_shape": tg}, **kw)
            if not (same_bits(r.rows[..., :28], auto.rows[..., :28]) and same_bits(r.scalars, auto.scalars)):
                fails.append(f"{what}: box_tile with time groups of {tg} is not bit-identical to the default")
    if moving:      # the box-packed form of the series (each step's box alone, dT/dt as the series' own data; include/lec_hip.h): no bit may move
        tc = eng.time_coefs_device(dom.time_s)
        pb = eng.prepare_boxes(boxes, nyb_min=auto.rows.shape[2], packed=True)
        pad = int(rng.integers(0, 4))
        hmax, wmax = max(b[3] - b[2] + 1 for b in boxes), max(b[1] - b[0] + 1 for b in boxes)
        ps = eng.pack_series(*f, boxes, tc, ny=min(ny, hmax + pad), nx=min(nx, wmax + pad))
        extra = {k: ps[k] for k in ("dTdt", "tm", "tp") if k in ps}
        r = eng.rowstats(ps["tair"], ps["u"], ps["v"], ps["omega"], ps["geopt"], pb, t_begin=0, t_count=nt, per_step_boxes=True,
                         **({"tcoef": tc} if "tm" in ps else {}), **extra)
        if not same_bits(r[..., :28], auto.rows[..., :28]):
            fails.append(f"{what}: the box-packed series is not bit-identical to the cube")
    # shards of the series: bit-identical
    if nt >= 3:
        a, b = 1, nt - 1
        part = eng.compute(*f, boxes[a:b] if moving else boxes, time_s=dom.time_s, t_begin=a, t_count=b - a, keep_rows=True,
                           per_step_boxes=moving, drop_any_time=(False if (nan_case or moving) else None))
        whole = auto if not nan_case else eng.compute(*f, boxes, drop_any_time=False, **kw)
        nyb = part.rows.shape[2]
        if not (same_bits(part.rows[..., :28], whole.rows[a:b, :, :nyb, :28]) and same_bits(part.scalars, whole.scalars[a:b])):
            fails.append(f"{what}: shard [{a}, {b}) is not bit-identical to the whole")
    if with_oracle and all(b[1] - b[0] >= 2 and b[3] - b[2] >= 2 for b in boxes):      # (2-point-wide boxes: eddy terms vanish to rounding)
        d64 = as_f64(dom)
        lim = [(dom.lon[b[0]], dom.lon[b[1]], dom.lat[b[2]], dom.lat[b[3]]) for b in boxes]
        with np.errstate(all="ignore"):
            ref_s, _ = o.lec_moving(d64, lim) if moving else o.lec_fixed(d64, *lim[0])
        got = auto.scalars_dict()
        for name in SCALARS:
            if name in ref_s:
                e = scale_err(got[name], ref_s[name])
                if not e <= 1e-9:
                    fails.append(f"{what}: {name} off the oracle by {e:.3e}")
    torch.cuda.synchronize()
    return fails


def config5_case(rng, case, T=512):
    """BASELINE config 5 at size: a track of 15-degree (61 x 61 point) boxes over a 37 x 162 x 243 crop, T steps (one rank's share of
    T = 4096 on 8 GPUs), the track's speed and phase random: the shipped box-tile kernel, its time groups of 2 and 4, the box-packed
    series and a shard in the middle must give the same bits; the one-wave-per-row kernel the same records to rounding."""
    from lorenzcycletoolkit_amd.synthetic import era5_like_levels, synthetic_cube
    lat, lon = np.arange(-57.75, -17.5 + 1e-9, 0.25), np.arange(-80.25, -19.75 + 1e-9, 0.25)
    level = era5_like_levels()
    eng = LECEngine(lat, lon, level, device=DEV)
    f = synthetic_cube(T, level, lat, lon, device=DEV, dtype=torch.float64, seed=int(rng.integers(1 << 20)))
    tg = np.arange(T)
    pa, pb_ = rng.uniform(150, 700), rng.uniform(200, 900)
    clat = -37.5 + rng.uniform(4, 12) * np.sin(2 * np.pi * tg / pa + rng.uniform(0, 6))
    clon = -50.0 + rng.uniform(8, 22) * np.cos(2 * np.pi * tg / pb_ + rng.uniform(0, 6))
    boxes = [eng.box_from_limits(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(clat, clon)]
    time_s = np.arange(T) * 3600.0
    tc = eng.time_coefs_device(time_s)
    cubes = (f["tair"], f["u"], f["v"], f["omega"], f["geopt"])
    what = f"config-5 case {case}: T={T}, track periods {pa:.0f} / {pb_:.0f}"
    fails = []
    plain = eng.prepare_boxes(boxes)
    base = eng.rowstats(*cubes, plain, tcoef=tc, t_begin=0, t_count=T, per_step_boxes=True)
    for g in (2, 4):
        r = eng.rowstats(*cubes, plain, tcoef=tc, t_begin=0, t_count=T, per_step_boxes=True, tuning={"kernel": "box_tile", "block_shape": g})
        if not same_bits(r[..., :28], base[..., :28]):
            fails.append(f"{what}: time groups of {g} are not bit-identical")
    sw = eng.rowstats(*cubes, plain, tcoef=tc, t_begin=0, t_count=T, per_step_boxes=True, tuning={"kernel": "row_sweep"})
    den = base[..., :28].abs().amax(dim=(0, 1, 2)).clamp_min(1e-300)
    err = float(((sw[..., :28] - base[..., :28]).abs().amax(dim=(0, 1, 2)) / den).max())
    if not err <= 2e-11:
        fails.append(f"{what}: one wave per row differs by {err:.3e}")
    ps = eng.pack_series(*cubes, boxes, tc)
    pk = eng.prepare_boxes(boxes, packed=True)
    r = eng.rowstats(ps["tair"], ps["u"], ps["v"], ps["omega"], ps["geopt"], pk, dTdt=ps["dTdt"], t_begin=0, t_count=T, per_step_boxes=True)
    if not same_bits(r[..., :28], base[..., :28]):
        fails.append(f"{what}: the box-packed series is not bit-identical")
    a, b = T // 3, T // 3 + 100
    r = eng.rowstats(*cubes, plain.part(a, b), tcoef=tc, t_begin=a, t_count=b - a, per_step_boxes=True)
    if not same_bits(r[..., :28], base[a:b, ..., :28]):
        fails.append(f"{what}: shard [{a}, {b}) is not bit-identical")
    if not bool(torch.isfinite(base).all()):
        fails.append(f"{what}: non-finite records")
    torch.cuda.synchronize()
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config5", type=int, default=0, help="run this many BASELINE-config-5-sized cases (T = 512 moving 61 x 61 boxes) instead")
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--oracle-every", type=int, default=2, help="compare every n-th case with the oracle too (it is the slow part)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    fails = []
    if args.config5:
        for c in range(args.config5):
            fails += config5_case(rng, c)
            print(f"config-5 case {c + 1}, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
        for ln in fails:
            print("FAIL", ln)
        print(f"soak config 5: {args.config5} cases, seed {args.seed}: {len(fails)} failures")
        sys.exit(1 if fails else 0)
    for c in range(args.cases):
        fails += one_case(rng, c, with_oracle=(c % args.oracle_every == 0))
        if (c + 1) % 25 == 0:
            print(f"{c + 1} cases, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
    for ln in fails:
        print("FAIL", ln)
    print(f"soak: {args.cases} cases, seed {args.seed}: {len(fails)} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
