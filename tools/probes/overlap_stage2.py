"""Does stage 2 of one part of a moving series overlap stage 1 of the next part on a second stream?  (measurement probe)"""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lorenzcycletoolkit_amd.engine import LECEngine
from lorenzcycletoolkit_amd.synthetic import era5_like_levels, synthetic_cube
dev = torch.device("cuda:0")
level = era5_like_levels(); lat = np.arange(-57.75, -17.5 + 1e-9, 0.25); lon = np.arange(-80.25, -19.75 + 1e-9, 0.25)
T = 512
f = synthetic_cube(T + 2, level, lat, lon, device=dev, dtype=torch.float64, seed=1, t0_global=0)
eng = LECEngine(lat, lon, level, device=dev)
tg = np.arange(T + 2)
boxes = [eng.box_from_limits(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(-37.5 + 12 * np.sin(2 * np.pi * tg / 400), -50 + 22 * np.cos(2 * np.pi * tg / 700))][1:T + 1]
prep = eng.prepare_boxes(boxes)
tc = eng.time_coefs_device(np.arange(T + 2) * 3600.0)
nl = len(level); w = LECEngine.packed_width(nl)
rows = torch.empty((T, nl, prep.bt.nyb_max, 32), dtype=torch.float64, device=dev)
out = torch.empty((T, w), dtype=torch.float64, device=dev); nan = torch.empty(T, dtype=torch.int32, device=dev)
s2 = torch.cuda.Stream(device=dev)
args = (f["tair"], f["u"], f["v"], f["omega"], f["geopt"])

def sequential():
    eng.rowstats(*args, prep, tcoef=tc, t_begin=1, t_count=T, rows_out=rows, per_step_boxes=True)
    eng.reduce(rows, prep, drop_any_time=False, out=out, nanflag_out=nan)

def overlapped(parts):
    main = torch.cuda.current_stream(dev)
    bounds = np.linspace(0, T, parts + 1).astype(int)
    evs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        pb = prep.part(a, b)
        eng.rowstats(*args, pb, tcoef=tc, t_begin=1 + a, t_count=b - a, rows_out=rows[a:b], per_step_boxes=True)
        e = torch.cuda.Event(); e.record(main)
        with torch.cuda.stream(s2):
            s2.wait_event(e)
            eng.reduce(rows[a:b], pb, drop_any_time=False, out=out[a:b], nanflag_out=nan[a:b])
    e2 = torch.cuda.Event(); e2.record(s2); main.wait_event(e2)

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

sequential(); torch.cuda.synchronize(); ref = out.clone()
print("sequential        %.3f ms" % timeit(sequential))
for p in (2, 4, 8):
    overlapped(p); torch.cuda.synchronize()
    print("overlapped x%d     %.3f ms   same bits: %s" % (p, timeit(lambda: overlapped(p)), torch.equal(out, ref)))
def only1():
    eng.rowstats(*args, prep, tcoef=tc, t_begin=1, t_count=T, rows_out=rows, per_step_boxes=True)
print("stage 1 alone     %.3f ms" % timeit(only1))
