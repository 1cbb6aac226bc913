"""Do the record stores of stage 1 get cheaper when the records land in a small buffer that is written again and again (one that the
memory-side cache could hold) instead of the series-long array?  Timing probe: a box-packed T = 512 series in chunks of C steps, every
chunk's records (a) into its own slice of the full array, (b) into the same C-step buffer.  Usage: probe_rows_ring.py [C ...]"""
import os, sys
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
boxes = [eng.box_from_limits(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(-37.5 + 12 * np.sin(2 * np.pi * tg / 400), -50 + 22 * np.cos(2 * np.pi * tg / 700))]
tc = eng.time_coefs_device(np.arange(T + 2) * 3600.0)
ps = eng.pack_series(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], boxes, tc)
del f
cut = lambda x: x[1:T + 1].contiguous()
fld = {k: cut(ps[k]) for k in ("tair", "u", "v", "omega", "geopt")}; dT = cut(ps["dTdt"]); del ps
nyb = max(b[3] - b[2] + 1 for b in boxes)
prep_all = eng.prepare_boxes(boxes[1:T + 1], nyb_min=nyb, packed=True)
nl = len(level)
rows = torch.empty((T, nl, nyb, 32), dtype=torch.float64, device=dev)

def run(C, ring):
    parts = [(a, min(a + C, T)) for a in range(0, T, C)]
    preps = [prep_all.part(a, b) for a, b in parts]
    def once():
        for (a, b), pb in zip(parts, preps):
            eng.rowstats(fld["tair"], fld["u"], fld["v"], fld["omega"], fld["geopt"], pb, dTdt=dT, t_begin=a, t_count=b - a,
                         rows_out=(rows[:b - a] if ring else rows[a:b]), per_step_boxes=True)
    for _ in range(3): once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10

for C in [int(x) for x in (sys.argv[1:] or ["512", "128", "64", "32"])]:
    for rep in range(2):
        print("chunks of %4d steps: own slices %.3f ms   one %d-step buffer %.3f ms   (%.0f MB of records per chunk)" % (C, run(C, False), C, run(C, True), C * nl * nyb * 256 / 1e6), flush=True)
