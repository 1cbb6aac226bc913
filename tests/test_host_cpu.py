"""CPU-side tests: coefficient tables against NumPy, the C-ABI library loads and exports every symbol
include/lec_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from lorenzcycletoolkit_amd import _lib, tables
from oracle import lec_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("x", [
    np.linspace(-50, 10, 13),
    np.array([0.0, 1.0, 2.5, 2.75, 6.0, 7.0]),
    np.array([100.0, 250.0]),
])
def test_gradient_coefs_match_numpy(x):
    rng = np.random.default_rng(0)
    f = rng.standard_normal(x.size)
    co = tables.gradient_coefs(x)
    fm, fp = np.r_[f[0], f[:-1]], np.r_[f[1:], f[-1]]
    got = co[:, 0] * fm + co[:, 1] * f + co[:, 2] * fp
    assert np.allclose(got, np.gradient(f, x, edge_order=1), rtol=1e-13, atol=1e-13)
    assert co[0, 0] == 0 and co[-1, 2] == 0


def test_trapz_weights():
    x = np.array([0.0, 0.5, 2.0, 2.25, 5.0])
    f = np.array([1.0, -2.0, 3.0, 0.5, 4.0])
    assert np.isclose(np.sum(tables.trapz_weights(x) * f), o.trapz(f, x, 0))


def test_nearest_index_ties_go_up():
    c = np.array([-5.0, -2.5, 0.0, 2.5])
    assert tables.nearest_index(c, -3.75) == 1
    assert tables.nearest_index(c, 100) == 3
    assert tables.nearest_index(c, -2.4) == 1
    assert tables.nearest_index(c, -3.75) == o.select_nearest(c, -3.75)


def test_level_tables_reproduce_static_stability_term():
    p = np.array([10000.0, 20000.0, 50000.0, 85000.0, 100000.0])
    T = np.array([210.0, 220.0, 255.0, 280.0, 290.0])
    levtab, levtab2 = tables.level_tables(p)
    theta = T / (p / 1e5) ** o.KAPPA
    ref = -(T / theta) * np.gradient(theta, p)
    Tm, Tp = np.r_[T[0], T[:-1]], np.r_[T[1:], T[-1]]
    assert np.allclose(levtab[:, 0] * Tm + levtab[:, 1] * T + levtab[:, 2] * Tp, ref, rtol=1e-12)
    assert np.array_equal(levtab2[:, 0], p)


def test_box_tables_layout_and_errors():
    lat = np.linspace(-40, -10, 13)
    lon = np.linspace(-60, -30, 25)
    bt = tables.build_box_tables(lat, lon, [(2, 20, 1, 10), (0, 24, 0, 12)])
    assert bt.nxb_max == 25 and bt.nyb_max == 13 and bt.lon_uniform
    xlen = np.deg2rad(lon[20]) - np.deg2rad(lon[2])
    assert np.isclose(bt.boxtab[0, 0], 1 / xlen)
    assert np.isclose(bt.wlon[0, :19].sum(), xlen) and np.all(bt.wlon[0, 19:] == 0)
    ylen = np.sin(np.deg2rad(lat[10])) - np.sin(np.deg2rad(lat[1]))
    assert np.isclose(bt.boxtab2[0, 1], -1 / (o.RE * ylen))
    assert np.isclose(bt.lattab2[0, :10, 0].sum() * ylen,
                      o.trapz(np.cos(np.deg2rad(lat[1:11])), np.deg2rad(lat[1:11]), 0))
    with pytest.raises(ValueError):
        tables.build_box_tables(lat, lon, [(3, 3, 1, 5)])
    with pytest.raises(ValueError):
        tables.build_box_tables(lat, lon, [(0, 30, 1, 5)])
    stretched = np.sort(lon + 0.3 * np.sin(np.arange(25)))
    assert not tables.build_box_tables(lat, stretched, [(2, 20, 1, 10)]).lon_uniform


def test_budgets_match_oracle():
    rng = np.random.default_rng(1)
    s = {k: rng.standard_normal(9) for k in ["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe"]}
    t = np.arange(9) * 21600.0
    a, b = tables.budgets_and_residuals(s, t), o.budgets_and_residuals(s, t)
    for k in b:
        assert np.array_equal(a[k], b[k]), k


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "lec_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(lec_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.lec_version() == _lib.LEC_ABI_VERSION
    assert lib.lec_max_row(_lib.LEC_F64, 1, 0) >= 1440 and lib.lec_max_row(_lib.LEC_F32, 1, _lib.KERNEL_TWO_SWEEP) >= 1440


def test_struct_sizes_match_header_layout():
    # 6 pointers + 14 int32 + 7 pointers + 2 pointers + lec_tuning (8 int32) ; 1 pointer + 4 int32 + 4 pointers + double + 2 int32 + 7 pointers
    assert ctypes.sizeof(_lib.Tuning) == 8 * 4
    assert ctypes.sizeof(_lib.RowstatsArgs) == 6 * 8 + 14 * 4 + 9 * 8 + 8 * 4
    assert ctypes.sizeof(_lib.ReduceArgs) == 8 + 4 * 4 + 4 * 8 + 8 + 2 * 4 + 8 + 6 * 8
    # lec_ingest_args: pointer + 2 int32 + 4 int32 + 3 int32 (+4 padding) + 3 pointers + 2 int32 + 4 doubles + 2 int32 + 2 pointers
    assert ctypes.sizeof(_lib.IngestArgs) == 8 + 9 * 4 + 4 + 3 * 8 + 2 * 4 + 4 * 8 + 2 * 4 + 2 * 8
    assert ctypes.sizeof(_lib.DiagArgs) == 3 * 8 + 4 * 4 + 6 * 8        # lec_diag_args: 3 pointers + 4 int32 + 6 pointers


def test_argument_errors_without_gpu():
    """Argument validation happens before any HIP call, so it is testable on CPU."""
    lib = _lib.load()
    a = _lib.RowstatsArgs()
    assert lib.lec_rowstats(ctypes.byref(a)) == 1
    assert b"null" in lib.lec_last_error()
    with pytest.raises(ValueError):
        _lib.check(lib.lec_rowstats(ctypes.byref(a)), "lec_rowstats")
    r = _lib.ReduceArgs()
    assert lib.lec_reduce(ctypes.byref(r)) == 1
    assert lib.lec_rowstats(None) == 1
    assert lib.lec_dropmask(ctypes.byref(r)) == 1
    g = _lib.IngestArgs()
    assert lib.lec_ingest(ctypes.byref(g)) == 1 and b"null" in lib.lec_last_error()
    assert lib.lec_ingest(None) == 1
    d = _lib.DiagArgs()
    assert lib.lec_track_diag(ctypes.byref(d)) == 1 and b"null" in lib.lec_last_error()
    assert lib.lec_track_diag(None) == 1
