"""Would a per-step, box-ALIGNED crop layout speed the moving-box kernel up?  (profiles/r05_notes.md section 5)

The pure-load probe (probe_boxread.hip, variant E) says rows that start on a 128-byte line with a 512-byte pitch replay 18 % faster than
today's 488-byte row fragments of a 243-column crop.  This runs the SHIPPED lec_boxtile kernel itself on cubes of both shapes -- the
kernel takes any grid and a box per step, so a [T][37][61][64] cube with the box (0..60, 0..60) at every step IS the aligned layout
as far as loads and arithmetic go (the numbers mean nothing for a moving box: the time neighbours of step t would have to be cropped on
box(t); here they are the neighbouring steps' own crops -- the access pattern of a layout that stores T(t-1), T(t+1) beside T costs
MORE than this, so this is the optimistic bound):
    today      cube [T][37][162][243], the bench's track, dT/dt from the cube's time neighbours (MODE 1)
    aligned 1  cube [T][37][61][64], box at the origin, MODE 1
    aligned 2  the same with a dT/dt cube (MODE 2: six streamed operands, no time neighbours)
    pitch 61   cube [T][37][61][61]: the crop without the alignment (rows of 488 bytes back to back)
"""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from lorenzcycletoolkit_amd.engine import LECEngine  # noqa: E402
from lorenzcycletoolkit_amd.synthetic import era5_like_levels  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
level = era5_like_levels()
time_s = np.arange(T) * 3600.0


def run(label, ny, nx, boxes_of, mode2=False):
    lat = -57.75 + 0.25 * np.arange(ny)
    lon = -80.25 + 0.25 * np.arange(nx)
    eng = LECEngine(lat, lon, level, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    f = [torch.randn((T, level.size, ny, nx), generator=g, dtype=torch.float64, device=dev) for _ in range(6 if mode2 else 5)]
    f[0] += 280.0
    boxes = eng.prepare_boxes(boxes_of(eng), nyb_min=61)
    rows = torch.empty((T, level.size, 61, 32), dtype=torch.float64, device=dev)
    tc = eng.time_coefs_device(time_s)
    kw = dict(dTdt=f[5]) if mode2 else dict(tcoef=tc)
    ms = []
    for i in range(8):
        tm = [] if i >= 3 else None
        eng.rowstats(f[0], f[1], f[2], f[3], f[4], boxes, t_begin=0, t_count=T, timing=tm, rows_out=rows, per_step_boxes=True, with_q=True, **kw)
        torch.cuda.synchronize()
        if tm:
            ms.append(sum(a.elapsed_time(b) for a, b in tm))
    alg = 5 * level.size * 61 * 61 * 8 * T
    m = float(np.median(ms))
    print(f"{label:38s} {m:7.3f} ms per {T} steps   {alg / (m * 1e-3) / 8e12:5.3f} of 8 TB/s (algorithmic 5 x 61 x 61 x 37 x 8 B per step)", flush=True)
    return m


def track_boxes(eng):
    tg = np.arange(T)
    clat = -37.5 + 12.0 * np.sin(2 * np.pi * tg / 400.0)
    clon = -50.0 + 22.0 * np.cos(2 * np.pi * tg / 700.0)
    return [eng.box_from_limits(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(clat, clon)]


def run_packed(label, nyp, nxp, dtdt_cube=False):
    """The real thing: the bench's cube and track, packed per step (T, u, v, omega, Phi + T(t-1), T(t+1) on the step's box)."""
    lat = -57.75 + 0.25 * np.arange(162)
    lon = -80.25 + 0.25 * np.arange(243)
    eng = LECEngine(lat, lon, level, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    f = [torch.randn((T, level.size, 162, 243), generator=g, dtype=torch.float64, device=dev) for _ in range(5)]
    f[0] += 280.0
    boxes = track_boxes(eng)
    pb = eng.prepare_boxes(boxes, nyb_min=61, packed=True)
    pk = [eng.pack_boxes(c, boxes, ny=nyp, nx=nxp) for c in f]
    tm, tp = eng.pack_boxes(f[0], boxes, shift=-1, ny=nyp, nx=nxp), eng.pack_boxes(f[0], boxes, shift=1, ny=nyp, nx=nxp)
    rows = torch.empty((T, level.size, 61, 32), dtype=torch.float64, device=dev)
    ref = eng.rowstats(*f, eng.prepare_boxes(boxes, nyb_min=61), tcoef=eng.time_coefs_device(time_s), t_begin=0, t_count=T, per_step_boxes=True)
    del f
    tc = eng.time_coefs_device(time_s)
    if dtdt_cube == "real":      # lec_dtdt: the stencil's own bits
        tm = eng.time_stencil(tm, pk[0], tp, tc)
        print("   boxes (w, h):", sorted({(b[1] - b[0] + 1, b[3] - b[2] + 1) for b in boxes}), "slabs", tuple(pk[0].shape[2:]), flush=True)
    ms = []
    for i in range(8):
        tmg = [] if i >= 3 else None
        if dtdt_cube:      # (timing only: the cube's VALUES are not the stencil's bits here)
            eng.rowstats(*pk, pb, dTdt=tm, t_begin=0, t_count=T, timing=tmg, rows_out=rows, per_step_boxes=True)
        else:
            eng.rowstats(*pk, pb, tcoef=tc, t_begin=0, t_count=T, timing=tmg, rows_out=rows, per_step_boxes=True, tm=tm, tp=tp)
        torch.cuda.synchronize()
        if tmg:
            ms.append(sum(a.elapsed_time(b) for a, b in tmg))
    alg = 5 * level.size * 61 * 61 * 8 * T
    m = float(np.median(ms))
    print(f"{label:38s} {m:7.3f} ms per {T} steps   {alg / (m * 1e-3) / 8e12:5.3f} of 8 TB/s   records bit-identical to the cube's: {bool(torch.equal(rows, ref))}", flush=True)
    return m


origin = lambda eng: [(0, 60, 0, 60)] * T
base = run("today: 162 x 243 crop, moving boxes", 162, 243, track_boxes)
a1 = run("aligned crop 61 x 64, neighbours in cube", 61, 64, origin)
a2 = run("aligned crop 61 x 64, dT/dt cube", 61, 64, origin, mode2=True)
p61 = run("dense crop 61 x 61 (unaligned rows)", 61, 61, origin)
k61 = run_packed("PACKED series, slabs 61 x 61 (7 cubes)", 61, 61)
k64 = run_packed("PACKED series, slabs 61 x 64 (7 cubes)", 61, 64)
d61 = run_packed("PACKED + dT/dt cube, 61 x 61 (6 cubes)", 61, 61, dtdt_cube=True)
d64 = run_packed("PACKED + dT/dt cube, 61 x 64 (6 cubes)", 61, 64, dtdt_cube=True)
dr = run_packed("PACKED + lec_dtdt cube, default slabs", None, None, dtdt_cube="real")
print(f"packed with a dT/dt cube / today: {base / d61:.3f} x (61 x 61), {base / d64:.3f} x (61 x 64)")
print(f"packed / today: {base / k61:.3f} x (61 x 61), {base / k64:.3f} x (61 x 64)")
print(f"aligned / today: {base / a1:.3f} x (MODE 1), {base / a2:.3f} x (MODE 2); dense 61 x 61: {base / p61:.3f} x")
