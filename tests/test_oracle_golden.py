"""Pins the CPU oracle (oracle/lec_oracle.py) to the reference's committed sample outputs.

Golden vectors: tests/golden/Catarina_NCEP-R2.nc -> Catarina_NCEP-R2_fixed/*.csv (full),
tests/golden/testdata_NCEP-R2.nc -> rows/columns of Reg1_{fixed,track}/{Kz,Ke,Ce,Cz}_lv_ISBL3.csv (partial).
The reference computed those in float32 (file dtype); the oracle fed the same float32 arrays
reproduces them at the rounding floor, and in clean fp64 at the float32 noise level.
"""
import os

import numpy as np
import pandas as pd
import pytest

from oracle import lec_oracle as o

CAT_BOX = (-55, -36, -35, -20)      # tests/golden/Catarina_NCEP-R2_fixed/log.txt:3
REG1_BOX = (-60, -30, -42.5, -17.5)  # tests/golden/inputs/box_limits_Reg1

# max_t |a - ref| / max_t |ref| with float32 inputs (the reference's own arithmetic)
F32_TOL = {
    "Az": 1e-13, "Ae": 1e-13, "Kz": 1e-13, "Ke": 1e-13, "Cz": 1e-13, "Ca": 1e-13, "Ce": 1e-13,
    "Gz": 1e-12, "Ge": 1e-12, "Ck": 1e-8,
    "BAz": 5e-7, "BAe": 5e-7, "BKz": 5e-7, "BKe": 5e-7,
    "∂Az/∂t (finite diff.)": 1e-13, "∂Ae/∂t (finite diff.)": 1e-13,
    "∂Kz/∂t (finite diff.)": 1e-13, "∂Ke/∂t (finite diff.)": 1e-13,
    "RGz": 5e-7, "RKz": 5e-7, "RGe": 5e-7, "RKe": 5e-7,
}


def _scale_err(a, r):
    a = np.asarray(a, dtype=np.float64)
    r = np.asarray(r, dtype=np.float64)
    return np.max(np.abs(a - r)) / np.max(np.abs(r))


@pytest.fixture(scope="module")
def catarina(golden_dir):
    ref = pd.read_csv(os.path.join(golden_dir, "Catarina_NCEP-R2_fixed", "Catarina_NCEP-R2_fixed_results.csv"),
                      index_col=0)
    out = {}
    for name, dt in (("f32", None), ("f64", np.float64)):
        dom = o.load_ncep_sample(os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), dtype=dt)
        dom = o.crop_domain(dom, *CAT_BOX)
        out[name] = o.lec_fixed(dom, *CAT_BOX)
    return ref, out


def test_catarina_shape(catarina):
    ref, out = catarina
    sc, lv = out["f32"]
    assert ref.shape == (36, 22)
    assert sc["Az"].shape == (36,)
    assert lv["Az"].shape == (36, 17)


@pytest.mark.parametrize("col", list(F32_TOL))
def test_catarina_float32_matches_reference(catarina, col):
    ref, out = catarina
    sc, _ = out["f32"]
    assert _scale_err(sc[col], ref[col].values) <= F32_TOL[col]


@pytest.mark.parametrize("col", list(F32_TOL))
def test_catarina_fp64_within_float32_noise(catarina, col):
    """Tolerance policy (ii) of SURVEY.md appendix D: |d| <= 2e-4 |ref| + 1e-4 max|ref|."""
    ref, out = catarina
    sc, _ = out["f64"]
    a, r = np.asarray(sc[col], dtype=np.float64), ref[col].values
    assert np.all(np.abs(a - r) <= 2e-4 * np.abs(r) + 1e-4 * np.max(np.abs(r)))


@pytest.mark.parametrize("term,sign,tol", [
    ("Az", 1, 1e-13), ("Ae", 1, 1e-13), ("Kz", 1, 2e-7), ("Ke", 1, 2e-7), ("Ce", 1, 1e-11),
    ("Ck", 1, 2e-7), ("Ge", 1, 1e-11), ("Gz", 1, 1e-11),
    # the committed Cz/Ca level tables were written by an older revision with the opposite sign
    ("Cz", -1, 1e-11), ("Ca", -1, 1e-13),
])
def test_catarina_level_tables(catarina, golden_dir, term, sign, tol):
    _, out = catarina
    _, lv = out["f32"]
    r = pd.read_csv(os.path.join(golden_dir, "Catarina_NCEP-R2_fixed", f"{term}_lv_ISBL3.csv"), index_col=0).values
    assert _scale_err(sign * np.asarray(lv[term]), r) <= tol


@pytest.fixture(scope="module")
def testdata(golden_dir):
    return o.load_ncep_sample(os.path.join(golden_dir, "testdata_NCEP-R2.nc"))


@pytest.mark.parametrize("term,sign,tol", [("Kz", 1, 2e-7), ("Ke", 1, 2e-7), ("Ce", 1, 1e-12), ("Cz", -1, 1e-10)])
def test_testdata_fixed_levels(testdata, golden_dir, term, sign, tol):
    dom = o.crop_domain(testdata, *REG1_BOX)
    _, lv = o.lec_fixed(dom, *REG1_BOX)
    cols = ["600.0", "700.0", "850.0", "925.0", "1000.0"]
    r = pd.read_csv(os.path.join(golden_dir, "Reg1_fixed", f"{term}_lv_ISBL3.csv"), index_col=0)[cols].values[:5]
    a = sign * np.asarray(lv[term], dtype=np.float64)
    assert np.max(np.abs(a - r) / np.abs(r)) <= tol


@pytest.mark.parametrize("term,sign,tol", [("Kz", 1, 2e-7), ("Ke", 1, 2e-7), ("Ce", 1, 1e-12), ("Cz", -1, 1e-10)])
def test_testdata_moving_levels(testdata, golden_dir, term, sign, tol):
    tr = pd.read_csv(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), sep=";")
    dom = o.crop_domain_track(testdata, tr.Lat.values, tr.Lon.values)
    boxes = [(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(tr.Lat, tr.Lon)]
    _, lv = o.lec_moving(dom, boxes)
    cols = ["60000.0", "70000.0", "85000.0", "92500.0", "100000.0"]
    r = pd.read_csv(os.path.join(golden_dir, "Reg1_track", f"{term}_lv_ISBL3.csv"), index_col=0)[cols].values
    a = sign * np.asarray(lv[term], dtype=np.float64)[: len(r)]
    assert np.max(np.abs(a - r) / np.abs(r)) <= tol


def test_nan_level_repair():
    p = np.array([100.0, 200.0, 400.0, 800.0, 1000.0])
    f = np.array([[1.0, np.nan, 3.0, 4.0, 5.0], [np.nan, 1.0, 2.0, 3.0, 4.0]])
    g, q = o.interpolate_and_drop_nan_levels(f, p)
    assert q.tolist() == [200.0, 400.0, 800.0, 1000.0]
    assert np.isclose(g[0, 0], 1.0 + (3.0 - 1.0) * (100.0 / 300.0))
    assert not np.isnan(g).any()
