#!/usr/bin/env python3
"""Debug helper (GPU box): case 0 of `tests/soak_streamed.py --seed 1 --fill-rate 0.002`: engine vs oracle, term by term and level by level."""
import argparse, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lorenzcycletoolkit_amd import dataset as ds
from lorenzcycletoolkit_amd.frameworks import BoxData
from oracle import cf_decode as cf, lec_oracle as o
from tests import soak_ingest as si
from tests.helpers import as_f64
np.set_printoptions(linewidth=200, precision=5)
rng = np.random.default_rng(1)
tmp = tempfile.mkdtemp(); os.makedirs(tmp + "/inputs"); open(tmp + "/inputs/namelist", "w").write(si.NAMELIST); os.chdir(tmp)
path = tmp + "/c.nc"
limits, what = si.write_case(rng, path, fill_rate=0.002)
print(what)
open("inputs/box_limits", "w").write("min_lon;%r\nmax_lon;%r\nmin_lat;%r\nmax_lat;%r\n" % limits)
args = argparse.Namespace(fixed=True, track=False, trackfile=None, residuals=True)
df = ds.read_namelist("inputs/namelist")
host = ds.slice_domain(ds.process_data(ds.open_dataset(path, df), args, df), args, df)
box = BoxData(host, df, *limits, args=args)
res = box.result
dom = as_f64(cf.prepare(path, si.NAMES, fixed_limits=limits))
for nm in ("tair", "u", "v", "omega", "geopt"):
    a = getattr(dom, nm)
    print(nm, "NaN count per (t, level):"); print(np.isnan(a).sum(axis=(2, 3)))
with np.errstate(all="ignore"):
    ref_s, ref_l = o.lec_fixed(dom, *limits)
got, gl = res.scalars_dict(), res.levels_dict()
for k in ("Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "Gz", "Ge"):
    print(k, "engine", np.asarray(got[k]), "oracle", np.asarray(ref_s[k]))
for k in ("Ca", "Ca_1", "Ca_2", "Gz", "Ge", "Ck"):
    if k in gl and k in ref_l:
        print(k, "levels engine\n", np.asarray(gl[k]), "\noracle\n", np.asarray(ref_l[k]))
print("nanflag", res.nanflag.cpu().numpy())
