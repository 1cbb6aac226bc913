"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances: fp64 arithmetic on both sides; the engine factors every term through row statistics
 while the oracle evaluates the reference's un-factored 4-D formulas, so they
agree to rounding.  TOL = 1e-9 of each term's scale (north_star asks for 1e-6 relative).
"""
import os

import numpy as np
import pandas as pd
import pytest

torch = pytest.importorskip("torch")

from oracle import lec_oracle as o
from tests.helpers import SCALARS, as_f64, compare, scale_err, synthetic_domain

pytestmark = pytest.mark.gpu

TOL = 1e-9


def _engine(dom):
    from lorenzcycletoolkit_amd.engine import LECEngine
    return LECEngine(dom.lat, dom.lon, dom.level, device="cuda:0")


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")


def run_fixed(dom, limits, **kw):
    eng = _engine(dom)
    box = eng.box_from_limits(*limits)
    res = eng.compute(_dev(dom.tair), _dev(dom.u), _dev(dom.v), _dev(dom.omega), _dev(dom.geopt), [box],
                      time_s=dom.time_s, **kw)
    torch.cuda.synchronize()
    return res


def run_moving(dom, limits_per_t):
    eng = _engine(dom)
    boxes = [eng.box_from_limits(*lim) for lim in limits_per_t]
    dTdt = o.moving_dTdt(dom)      # host-side here; the framework computes it on the device
    res = eng.compute(_dev(dom.tair), _dev(dom.u), _dev(dom.v), _dev(dom.omega), _dev(dom.geopt), boxes,
                      dTdt=_dev(dTdt.astype(dom.tair.dtype)))
    torch.cuda.synchronize()
    return res


def check_fixed(dom, limits, tol=TOL, what=""):
    res = run_fixed(dom, limits)
    assert int(res.nanflag.sum()) == 0
    ref_s, ref_l = o.lec_fixed(as_f64(dom), *limits)
    return compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, tol, what, time_s=dom.time_s)


# ---------------------------------------------------------------------------------------------
# the reference's own samples
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, None])
def test_catarina_fixed(golden_dir, dtype):
    """BASELINE config 2 stand-in (testdata_ERA5.nc is absent): Catarina sample, fixed box, fp64 and
    fp32 storage.  Oracle = fp64 evaluation of the same values."""
    dom = o.load_ncep_sample(os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), dtype=dtype)
    limits = (-55, -36, -35, -20)
    worst = check_fixed(o.crop_domain(dom, *limits), limits, what="catarina")
    print(worst)


def test_catarina_against_committed_csv(golden_dir):
    """Engine (fp64) vs the reference's committed float32-generated CSV: float32 noise only
    (tolerance policy (ii) of SURVEY.md appendix D)."""
    dom = o.load_ncep_sample(os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), dtype=np.float64)
    limits = (-55, -36, -35, -20)
    res = run_fixed(o.crop_domain(dom, *limits), limits)
    ref = pd.read_csv(os.path.join(golden_dir, "Catarina_NCEP-R2_fixed", "Catarina_NCEP-R2_fixed_results.csv"), index_col=0)
    got = res.scalars_dict()
    for name in ["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "Gz", "Ge"]:
        a, r = got[name], ref[name].values
        assert np.all(np.abs(a - r) <= 2e-4 * np.abs(r) + 1e-4 * np.max(np.abs(r))), name


def test_testdata_fixed_box_inside_domain(golden_dir):
    """Box strictly inside a larger domain with an odd row length (nx = 41): exercises the box offsets
    and the unaligned (scalar-load) kernel.  Q differentiates T in time over the whole cube."""
    dom = o.load_ncep_sample(os.path.join(golden_dir, "testdata_NCEP-R2.nc"), dtype=np.float64)
    limits = (-60, -30, -42.5, -17.5)
    res = run_fixed(dom, limits)
    ref_s, ref_l = o.lec_fixed(o.crop_domain(dom, *limits), *limits)
    compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, TOL, "testdata fixed", time_s=dom.time_s)


def test_testdata_moving(golden_dir):
    """Semi-Lagrangian boxes from the reference's track file, dT/dt supplied as a cube."""
    dom = o.load_ncep_sample(os.path.join(golden_dir, "testdata_NCEP-R2.nc"), dtype=np.float64)
    tr = pd.read_csv(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), sep=";")
    domt = o.crop_domain_track(dom, tr.Lat.values, tr.Lon.values)
    limits = [(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(tr.Lat, tr.Lon)]
    res = run_moving(domt, limits)
    ref_s, ref_l = o.lec_moving(domt, limits)
    compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, TOL, "testdata moving", time_s=dom.time_s)


# ---------------------------------------------------------------------------------------------
# synthetic shapes: every kernel configuration
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nx,dtype,nonuni", [
    (48, np.float64, False),     # one trip per row (the moving-box instantiation)
    (200, np.float64, False),    # two trips: an edge trip at either end, no middle trip
    (300, np.float64, False),    # three trips: one middle trip
    (1000, np.float64, False),   # 8 trips
    (1001, np.float64, False),   # odd row length -> scalar-load kernel
    (1800, np.float64, False),
    (2400, np.float64, False),
    (3500, np.float64, False),   # 28 trips (beyond the two-sweep kernel's row limit)
    (3500, np.float32, False),
    (300, np.float32, False),    # float4 trips
    (1000, np.float32, False),
    (303, np.float32, False),    # float, unaligned
    (300, np.float64, True),     # non-uniform longitudes: table path
    (1000, np.float32, True),
])
def test_synthetic_fixed(nx, dtype, nonuni):
    dom = synthetic_domain(4, 5, 12, nx, seed=nx, dtype=dtype, nonuniform_lon=nonuni)
    limits = (dom.lon[3], dom.lon[-3], dom.lat[1], dom.lat[-2])     # odd box start: misaligned rows
    check_fixed(dom, limits, what=f"synthetic nx={nx} {np.dtype(dtype).name} nonuni={nonuni}")


@pytest.mark.parametrize("nt,nl,ny,nx,box", [
    (2, 2, 3, 4, (1, 2, 0, 1)),        # the smallest everything: two time steps, two levels, a 2 x 2 box
    (2, 3, 5, 9, (0, 8, 0, 4)),        # two time steps: the first-step kernel plus a one-step row-block launch
    (3, 2, 4, 6, (2, 5, 1, 3)),
    (5, 4, 6, 130, (1, 128, 2, 3)),    # two latitudes only, rows that start on an odd element
])
def test_minimal_extents(nt, nl, ny, nx, box):
    dom = synthetic_domain(nt, nl, ny, nx, seed=100 + nx)
    iw, ie, js, jn = box
    limits = (dom.lon[iw], dom.lon[ie], dom.lat[js], dom.lat[jn])
    check_fixed(dom, limits, what=f"minimal nt={nt} nl={nl} ny={ny} nx={nx}")
    # every single-step shard equals the whole-series result bit for bit
    full = run_fixed(dom, limits)
    for t in range(nt):
        part = run_fixed(dom, limits, t_begin=t, t_count=1)
        assert torch.equal(full.scalars[t:t + 1], part.scalars), f"single-step shard {t}"


def test_synthetic_full_domain_box():
    dom = synthetic_domain(3, 6, 9, 64, seed=5)
    limits = (dom.lon[0], dom.lon[-1], dom.lat[0], dom.lat[-1])
    check_fixed(dom, limits, what="full-domain box")


def test_synthetic_moving_variable_boxes():
    """Boxes of different sizes per time step (track with width/length columns)."""
    dom = synthetic_domain(5, 6, 40, 60, seed=11)
    cen = [(-35.0 + 2 * t, -50.0 + 3 * t) for t in range(5)]
    size = [(10, 12), (12, 10), (14, 14), (8, 16), (10, 10)]
    limits = [(lo - w / 2, lo + w / 2, la - l / 2, la + l / 2) for (la, lo), (w, l) in zip(cen, size)]
    res = run_moving(dom, limits)
    ref_s, ref_l = o.lec_moving(dom, limits)
    compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, TOL, "moving variable boxes", time_s=dom.time_s)


def _rows_close(a, b, what, tol=1e-11, scalar_rtol=1e-10):
    ra, rb = a.rows.cpu().numpy(), b.rows.cpu().numpy()
    for s in range(28):
        scale = np.max(np.abs(rb[..., s]))
        assert np.max(np.abs(ra[..., s] - rb[..., s])) <= tol * max(scale, 1e-300), f"{what}: row statistic {s}"
    if scalar_rtol is not None:
        assert torch.allclose(a.scalars, b.scalars, rtol=scalar_rtol, atol=0), what


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_config5_shape_moving_boxes(dtype):
    """BASELINE config 5's shape: a 0.25-degree track-extent crop (162 x 243), 37 levels, one 15 x 15 degree box (61 x 61
    points) per time step along a track, dT/dt differentiated on the device over the series' time axis (what lec_moving does).
    Engine (box-tile kernel) vs oracle; then the same series in shards and chunks: every piece is bit-identical to the whole."""
    nt = 6
    dom = synthetic_domain(nt, 37, 162, 243, seed=55, dtype=dtype, lat0=-57.75, lat1=-17.5, lon0=-80.25, lon1=-19.75, dt_s=3600.0)
    clat = -37.5 + 12.0 * np.sin(2 * np.pi * np.arange(nt) / 5.0)
    clon = -50.0 + 22.0 * np.cos(2 * np.pi * np.arange(nt) / 7.0)
    limits = [(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(clat, clon)]
    eng = _engine(dom)
    boxes = [eng.box_from_limits(*lim) for lim in limits]
    assert all(b[1] - b[0] == 60 and b[3] - b[2] == 60 for b in boxes)
    f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    res = eng.compute(*f, boxes, time_s=dom.time_s, keep_rows=True)
    torch.cuda.synchronize()
    ref_s, ref_l = o.lec_moving(as_f64(dom), limits)
    compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, TOL, f"config-5 shape {np.dtype(dtype).name}", time_s=dom.time_s)
    # shards / chunks of the series (each sees the cube's time neighbours as its halo)
    for a, b in ((0, 2), (2, 5), (5, 6)):
        part = eng.compute(*f, boxes[a:b], time_s=dom.time_s, t_begin=a, t_count=b - a, keep_rows=True, per_step_boxes=True)
        assert torch.equal(part.rows, res.rows[a:b]) and torch.equal(part.scalars, res.scalars[a:b]), (a, b)
    # a chunk held as its own halo'd cube (what a time-sharded rank or a streamed chunk holds): same bits again
    h0, h1 = 1, 5
    g = [x[h0:h1].contiguous() for x in f]
    part = eng.compute(*g, boxes[2:4], time_s=dom.time_s[h0:h1], t_begin=1, t_count=2, keep_rows=True)
    assert torch.equal(part.rows, res.rows[2:4])
    # the one-wave-per-row kernel is an independent formulation of the same records
    sweep = eng.compute(*f, boxes, time_s=dom.time_s, keep_rows=True, tuning={"kernel": "row_sweep"})
    _rows_close(res, sweep, "box tiles vs one wave per row, 61 x 61 boxes")


def test_box_tile_kernel_random_geometries():
    """Six corner geometries and forty random ones (box 2..150 columns x 2..90 rows anywhere in the grid, 2..23 levels, 1..5 time steps, uniform or not,
    fp32 / fp64 storage, dT/dt from the time axis or from a cube, per-step boxes of different sizes): the box-tile kernel against
    the one-wave-per-row kernel, record by record, and the padding rows of the lower boxes must be zero."""
    rng = np.random.default_rng(2024)
    # the corners of the cube first (the kernel's halo and prefetch addresses are clamped, never masked): whole grid, the first and the
    # last 2 x 2 / 3 x 3 points, a two-column strip over all latitudes at the east edge, a two-row strip at the north edge
    corner_cases = [
        (2, 5, 12, 70, [(0, 69, 0, 11)] * 2), (2, 4, 9, 10, [(0, 1, 0, 1), (7, 9, 6, 8)]), (3, 3, 40, 30, [(28, 29, 0, 39)] * 3),
        (1, 2, 30, 140, [(0, 139, 28, 29)]), (2, 6, 66, 64, [(0, 63, 0, 64), (0, 63, 1, 65)]), (2, 3, 8, 65, [(0, 64, 0, 7), (1, 64, 2, 7)]),
    ]
    for case in range(40 + len(corner_cases)):
        if case < len(corner_cases):
            nt, nl, ny, nx, boxes = corner_cases[case]
            dtype, nonuni, use_cube = (np.float64, np.float32)[case % 2], case % 3 == 2, case % 2 == 1 or nt == 1
            dom = synthetic_domain(nt, nl, ny, nx, seed=900 + case, dtype=dtype, nonuniform_lon=nonuni)
        else:
            nt, nl = int(rng.integers(1, 6)), int(rng.integers(2, 24))
            ny, nx = int(rng.integers(8, 100)), int(rng.integers(8, 170))
            dtype = np.float32 if rng.random() < 0.3 else np.float64
            nonuni = bool(rng.random() < 0.3)
            use_cube = bool(rng.random() < 0.4) or nt == 1
            dom = synthetic_domain(nt, nl, ny, nx, seed=1000 + case, dtype=dtype, nonuniform_lon=nonuni)
            boxes = []
            for _ in range(nt):
                wx, wy = int(rng.integers(2, min(nx, 150) + 1)), int(rng.integers(2, min(ny, 90) + 1))
                iw, js = int(rng.integers(0, nx - wx + 1)), int(rng.integers(0, ny - wy + 1))
                boxes.append((iw, iw + wx - 1, js, js + wy - 1))
        eng = _engine(dom)
        f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
        kw = dict(keep_rows=True, per_step_boxes=True)
        if use_cube:
            kw["dTdt"] = _dev(rng.standard_normal(dom.tair.shape).astype(dtype) * 1e-4)
        else:
            kw["time_s"] = dom.time_s
        a = eng.compute(*f, boxes, tuning={"kernel": "box_tile"}, **kw)
        b = eng.compute(*f, boxes, tuning={"kernel": "row_sweep"}, **kw)
        _rows_close(a, b, f"case {case}: nt={nt} nl={nl} grid {ny}x{nx} boxes {boxes} {np.dtype(dtype).name} nonuni={nonuni} cube={use_cube}")
        nyb_max = a.rows.shape[2]
        for t, bx in enumerate(boxes):
            nyb = bx[3] - bx[2] + 1
            assert torch.all(a.rows[t, :, nyb:nyb_max] == 0), (case, t)
        assert torch.isfinite(a.rows).all(), case


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_box_tile_long_level_walks(dtype):
    """The box-tile kernel with MANY levels per wave (a wave keeps its chunk's static-stability coefficients one per lane,
    21 levels at most; every earlier moving test had <= 6 time steps, so the launch rule always picked 5-level chunks).  37 and 45
    levels, Q from the time axis: level chunks of 5, 8, 19, 21 and the automatic choice give the SAME BITS, agree with the independent
    one-wave-per-row kernel record by record and with the oracle; 22 levels per wave is refused."""
    for nl in (37, 45):
        nt = 3
        dom = synthetic_domain(nt, nl, 30, 70, seed=300 + nl, dtype=dtype, dt_s=3600.0)
        boxes = [(4 + t, 4 + t + 40, 3 + 2 * t, 3 + 2 * t + 17) for t in range(nt)]
        limits = [(dom.lon[b[0]], dom.lon[b[1]], dom.lat[b[2]], dom.lat[b[3]]) for b in boxes]
        eng = _engine(dom)
        f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
        kw = dict(time_s=dom.time_s, keep_rows=True, per_step_boxes=True)
        auto = eng.compute(*f, boxes, **kw)
        for tj in (5, 8, 19, 21):
            r = eng.compute(*f, boxes, tuning={"kernel": "box_tile", "tile_j": tj}, **kw)
            assert torch.equal(r.rows, auto.rows), (nl, tj)
            assert torch.equal(r.scalars, auto.scalars) and torch.equal(r.levels, auto.levels), (nl, tj)
        sweep = eng.compute(*f, boxes, tuning={"kernel": "row_sweep"}, **kw)
        _rows_close(auto, sweep, f"box tiles vs one wave per row, {nl} levels")
        with pytest.raises(ValueError, match="21 levels"):
            eng.compute(*f, boxes, tuning={"kernel": "box_tile", "tile_j": 22}, **kw)
        if nl == 37:
            ref_s, ref_l = o.lec_moving(as_f64(dom), limits)
            compare(auto.scalars_dict(), auto.levels_dict(), ref_s, ref_l, TOL, f"long level walk {np.dtype(dtype).name}", time_s=dom.time_s)


def test_box_tile_launch_rule_at_long_series():
    """A series long enough (2048 steps of a low box) for the launch rule to pick its LONGEST level walk (two chunks of 19 for 37
    levels -- it used to pick one of 37 and read other levels' coefficients): same bits as 5-level chunks, and the one-wave-per-row
    kernel agrees."""
    nt, nl = 2048, 37
    dom = synthetic_domain(4, nl, 20, 40, seed=77, dt_s=3600.0)
    rep = lambda a: _dev(a).repeat((nt // 4, 1, 1, 1)).contiguous()
    f = [rep(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    f[0] += torch.linspace(0, 1, nt, dtype=torch.float64, device="cuda:0")[:, None, None, None]     # dT/dt differs per step
    boxes = [(2 + t % 5, 2 + t % 5 + 30, 1 + t % 3, 1 + t % 3 + 15) for t in range(nt)]      # 16 rows: 8 x 256 x 4 = 8192 waves per level chunk
    eng = _engine(dom)
    kw = dict(time_s=np.arange(nt) * 3600.0, keep_rows=True, per_step_boxes=True)
    auto = eng.compute(*f, boxes, **kw)
    five = eng.compute(*f, boxes, tuning={"kernel": "box_tile", "tile_j": 5}, **kw)
    assert torch.equal(auto.rows, five.rows) and torch.equal(auto.scalars, five.scalars)
    sweep = eng.compute(*f, boxes, tuning={"kernel": "row_sweep"}, **kw)
    _rows_close(auto, sweep, "2048-step series: box tiles vs one wave per row")


@pytest.mark.parametrize("nonuni", [False, True])
def test_moving_boxes_of_mixed_widths_shard_bit_identically(nonuni):
    """A track whose boxes are 40 to 90 columns wide.  The box-tile kernel keeps the level window in registers only when every row of
    the CALL fits one 64-column chunk; a shard that holds just the narrow boxes takes that instantiation, the whole series the other
    one -- same arithmetic, so the records must agree bit for bit (as must the oracle, to rounding)."""
    nt = 6
    dom = synthetic_domain(nt, 7, 50, 120, seed=91, nonuniform_lon=nonuni, lon0=-80.0, lon1=-20.5)
    widths = [40, 90, 50, 64, 65, 30]           # grid columns
    j0 = [3, 5, 8, 10, 6, 2]
    i0 = [5, 10, 20, 30, 12, 70]
    boxes = [(i, i + w - 1, j, j + 20 + t) for t, (i, j, w) in enumerate(zip(i0, j0, widths))]
    limits = [(dom.lon[b[0]], dom.lon[b[1]], dom.lat[b[2]], dom.lat[b[3]]) for b in boxes]
    eng = _engine(dom)
    assert [eng.box_from_limits(*lim) for lim in limits] == boxes
    f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    whole = eng.compute(*f, boxes, time_s=dom.time_s, keep_rows=True)
    ref_s, ref_l = o.lec_moving(dom, limits)
    compare(whole.scalars_dict(), whole.levels_dict(), ref_s, ref_l, TOL, "mixed widths", time_s=dom.time_s)
    for a, b in ((0, 1), (2, 4), (5, 6), (0, 3)):
        part = eng.compute(*f, boxes[a:b], time_s=dom.time_s, t_begin=a, t_count=b - a, keep_rows=True, per_step_boxes=True)
        ny = part.rows.shape[2]
        assert torch.equal(part.rows, whole.rows[a:b, :, :ny]), (a, b)
        assert torch.equal(part.scalars, whole.scalars[a:b]), (a, b)


@pytest.mark.parametrize("moving", [False, True])
def test_small_box_level_kernel_gives_the_bits_of_the_general_one(moving):
    """Stage 2 has a kernel for boxes of up to 64 rows (one row per lane, area means formed in place) and a general one (rows staged
    64 at a time).  The same records in a buffer padded to 70 rows take the general path: every output must carry the same bits,
    NaN repair included."""
    nt, nl, ny, nx = 6, 8, 80, 60
    dom = synthetic_domain(nt, nl, ny, nx, seed=91)
    dom.tair[1:3, -1, 5:9, 10:20] = np.nan
    dom.omega[1:3, 4, 12:15, 10:20] = np.nan
    eng = _engine(dom)
    f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    boxes = [(3 + t, 50 + t, 2, 40 - t) for t in range(nt)] if moving else [(3, 55, 1, 41)]
    kw = dict(time_s=dom.time_s, per_step_boxes=moving)
    small = eng.compute(*f, boxes, keep_rows=True, **kw)
    nyb = small.rows.shape[2]
    assert nyb <= 64
    padded = torch.zeros((nt, nl, 70, small.rows.shape[3]), dtype=torch.float64, device="cuda:0")
    eng.rowstats(*f, boxes, rows_out=padded, **kw)
    assert torch.equal(torch.nan_to_num(padded[:, :, :nyb], nan=-7.0), torch.nan_to_num(small.rows, nan=-7.0))
    general = eng.reduce(padded, boxes, drop_any_time=not moving)
    for name in ("scalars", "levels"):
        a, b = getattr(small, name), getattr(general, name)
        assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), name
    assert torch.equal(small.nanflag, general.nanflag) and int(small.nanflag.sum()) > 0


def test_random_domains_against_the_oracle():
    """Twelve random small domains (3..6 time steps on an uneven time axis, 3..9 levels, boxes of 3..70 columns x 3..18 rows, uniform
    or table longitudes, fp32 / fp64 storage): all 16 terms, budgets, residuals and 21 level tables of the fixed and of the moving
    framework against the un-factored oracle."""
    rng = np.random.default_rng(4242)
    for case in range(12):
        nt, nl = int(rng.integers(3, 7)), int(rng.integers(3, 10))
        ny, nx = int(rng.integers(6, 24)), int(rng.integers(8, 90))
        dtype = np.float32 if case % 3 == 2 else np.float64
        dom = synthetic_domain(nt, nl, ny, nx, seed=5000 + case, dtype=dtype, nonuniform_lon=bool(case % 4 == 1))
        dom.time_s = np.cumsum(rng.integers(1, 4, nt) * 3600.0)            # 1..3-hourly steps: np.gradient's non-uniform stencil

        def limits():
            wx, wy = int(rng.integers(3, min(nx, 70) + 1)), int(rng.integers(3, min(ny, 18) + 1))
            iw, js = int(rng.integers(0, nx - wx + 1)), int(rng.integers(0, ny - wy + 1))
            return (dom.lon[iw], dom.lon[iw + wx - 1], dom.lat[js], dom.lat[js + wy - 1])

        lim = limits()
        what = f"case {case}: nt={nt} nl={nl} grid {ny}x{nx} {np.dtype(dtype).name}"
        check_fixed(dom, lim, what=what + f" fixed {lim}")
        per_step = [limits() for _ in range(nt)]
        eng = _engine(dom)
        res = eng.compute(*[_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)], [eng.box_from_limits(*lm) for lm in per_step],
                          time_s=dom.time_s)          # dT/dt over the series' time axis on the device, as the moving framework does
        ref_s, ref_l = o.lec_moving(as_f64(dom), per_step)
        compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, TOL, what + " moving", time_s=dom.time_s)


def test_fixed_box_kernels_random_geometries():
    """Thirty random fixed boxes (2..2600 columns starting at even and odd columns -- aligned and unaligned vector trips, one-trip and
    many-trip rows -- 2..40 rows, 2..9 levels, 2..5 time steps, uniform or table longitudes, fp32 / fp64 storage, dT/dt from the time
    axis, from a cube, or no Q at all): the library's default kernels (row-block / row-sweep) against the two-sweep formulation,
    which forms deviations from the zonal mean first like the reference does.  Also a time shard of each case, bit for bit."""
    rng = np.random.default_rng(77)
    lib = __import__("lorenzcycletoolkit_amd._lib", fromlist=["load"])
    for case in range(30):
        nt, nl = int(rng.integers(2, 6)), int(rng.integers(2, 10))
        ny = int(rng.integers(4, 44))
        nx = int(rng.choice([rng.integers(6, 140), rng.integers(140, 700), rng.integers(700, 2700)]))
        dtype = np.float32 if rng.random() < 0.35 else np.float64
        nonuni = bool(rng.random() < 0.25)
        mode = ("time", "cube", "noq")[int(rng.integers(0, 3))]
        dom = synthetic_domain(nt, nl, ny, nx, seed=3000 + case, dtype=dtype, nonuniform_lon=nonuni)
        wx, wy = int(rng.integers(2, nx + 1)), int(rng.integers(2, min(ny, 40) + 1))
        iw, js = int(rng.integers(0, nx - wx + 1)), int(rng.integers(0, ny - wy + 1))
        box = (iw, iw + wx - 1, js, js + wy - 1)
        eng = _engine(dom)
        f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
        kw = dict(keep_rows=True)
        if mode == "cube":
            kw["dTdt"] = _dev(rng.standard_normal(dom.tair.shape).astype(dtype) * 1e-4)
        elif mode == "time":
            kw["time_s"] = dom.time_s
        else:
            kw["with_q"] = False
        what = f"case {case}: nt={nt} nl={nl} grid {ny}x{nx} box {box} {np.dtype(dtype).name} nonuni={nonuni} {mode}"
        a = eng.compute(*f, [box], **kw)
        if wx <= lib.load().lec_max_row(lib.LEC_F64 if dtype == np.float64 else lib.LEC_F32, 0, lib.KERNEL_TWO_SWEEP):
            b = eng.compute(*f, [box], tuning={"kernel": "two_sweep"}, **kw)
            _rows_close(a, b, what, scalar_rtol=None)
            # the integrated terms against each term's own scale (a term that nearly cancels at one time step has no relative accuracy)
            sa, sb = a.scalars.cpu().numpy(), b.scalars.cpu().numpy()
            assert np.all(np.abs(sa - sb) <= 1e-9 * np.maximum(np.max(np.abs(sb), axis=0), 1e-300)), what
        assert torch.isfinite(a.rows[..., :28]).all(), what
        t0 = int(rng.integers(0, nt - 1))
        part = eng.compute(*f, [box], t_begin=t0, t_count=nt - t0, **kw)
        assert torch.equal(part.rows[..., :28], a.rows[t0:, ..., :28]) and torch.equal(part.scalars, a.scalars[t0:]), what


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_kernel_families_agree(dtype):
    """Three independent formulations of the same row statistics, selected through ``lec_tuning.kernel`` (the
    library reads no environment): the default single-sweep kernels (sums about a shift, lanes along longitude),
    the two-sweep kernel (deviation from the zonal mean, then products -- the reference's own order) and the
    box-tile kernel (one lane per latitude row, serial sums over longitude)."""
    dom = synthetic_domain(4, 6, 14, 1000, seed=77, dtype=dtype)
    limits = (dom.lon[3], dom.lon[-4], dom.lat[1], dom.lat[-2])
    a = run_fixed(dom, limits, keep_rows=True)
    b = run_fixed(dom, limits, keep_rows=True, tuning={"kernel": "two_sweep"})
    c = run_fixed(dom, limits, keep_rows=True, tuning={"kernel": "box_tile"})
    d = run_fixed(dom, limits, keep_rows=True, tuning={"kernel": "row_sweep", "order": "memory"})
    _rows_close(a, b, "default vs two-sweep")
    _rows_close(c, b, "box tiles vs two-sweep")
    assert torch.equal(a.rows, d.rows) and torch.equal(a.scalars, d.scalars)     # block order is speed only


@pytest.mark.parametrize("ny,nx,nonuni,dtype", [
    (70, 23, False, np.float64),     # two row groups (the second one partial), two strips
    (9, 16, False, np.float64),      # exactly one strip
    (12, 17, False, np.float32),     # one point into the second strip
    (130, 50, True, np.float64),     # three row groups, non-uniform longitudes
    (5, 2, False, np.float64),       # the smallest row
])
def test_box_tile_kernel_shapes(ny, nx, nonuni, dtype):
    """The box-tile kernel on boxes that exercise partial strips (16 columns) and partial row groups (64 rows),
    with and without Q / a dT/dt cube, against the oracle and the one-wave-per-row kernel."""
    dom = synthetic_domain(3, 4, ny + 3, nx + 4, seed=ny * nx, dtype=dtype, nonuniform_lon=nonuni)
    limits = (dom.lon[1], dom.lon[nx], dom.lat[2], dom.lat[ny + 1])
    res = run_fixed(dom, limits, tuning={"kernel": "box_tile"}, keep_rows=True)
    ref_s, ref_l = o.lec_fixed(as_f64(o.crop_domain(dom, *limits)), *limits)
    compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, TOL, f"box tiles {ny}x{nx}", time_s=dom.time_s)
    sweep = run_fixed(dom, limits, tuning={"kernel": "row_sweep"}, keep_rows=True)
    _rows_close(res, sweep, "box tiles vs one wave per row")
    eng = _engine(dom)
    box = eng.box_from_limits(*limits)
    noq = [eng.compute(_dev(dom.tair), _dev(dom.u), _dev(dom.v), _dev(dom.omega), None, [box], with_q=False,
                       keep_rows=True, tuning={"kernel": k}) for k in ("box_tile", "row_sweep")]
    _rows_close(noq[0], noq[1], "no-Q box tiles vs one wave per row")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("blk", ["212", "222", "122", "211", "121", "112"])
def test_row_block_kernel_is_bit_identical_to_one_wave_per_row(dtype, blk):
    """The row-block kernel (default for all terms on a fixed box: block mates exchange T through LDS) must give
    the very same row records as the one-wave-per-row kernel; odd extents exercise partial blocks, the shard
    exercises halo time steps on both sides."""
    dom = synthetic_domain(7, 5, 13, 600, seed=21, dtype=dtype)
    limits = (dom.lon[3], dom.lon[-4], dom.lat[1], dom.lat[-2])
    # the row-block kernel walks float2 vectors: f32_vec = 2 gives the one-wave-per-row kernel the same lane-to-element map
    for kw in ({}, {"t_begin": 1, "t_count": 5}):
        a = run_fixed(dom, limits, keep_rows=True, tuning={"kernel": "row_sweep", "f32_vec": 2}, **kw)
        b = run_fixed(dom, limits, keep_rows=True, tuning={"kernel": "row_block", "block_shape": int(blk)}, **kw)
        assert torch.equal(a.rows, b.rows), f"rows differ for block shape {blk} {kw}"
        assert torch.equal(a.scalars, b.scalars)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nonuni", [False, True])
def test_time_shard_invariance(nonuni, dtype):
    """Processing [t0, t1) of a cube gives bit-identical results to processing the whole cube -- on stretched longitudes too, where the
    trapezoid weights are not powers of two: the first step of a shard forms <a-> itself, every other step takes <a> of the row
    before, and the two must be the same sum formed the same way (found by tests/soak_gpu.py: they once differed by an ulp)."""
    dom = synthetic_domain(6, 5, 10, 128, seed=3, dtype=dtype, nonuniform_lon=nonuni)
    limits = (dom.lon[2], dom.lon[-2], dom.lat[1], dom.lat[-2])
    full = run_fixed(dom, limits, keep_rows=True)
    for (a, n) in ((2, 3), (1, 1), (0, 2), (4, 2)):
        part = run_fixed(dom, limits, t_begin=a, t_count=n, keep_rows=True)
        assert torch.equal(full.rows[a:a + n, ..., :28], part.rows[..., :28]), (a, n)
        assert torch.equal(full.scalars[a:a + n], part.scalars), (a, n)
        assert torch.equal(full.levels[a:a + n], part.levels), (a, n)


def test_two_point_wide_box_of_a_stretched_grid_shards_bit_identically():
    """Two grid points are always evenly spaced; the step of a track that holds such a box must still run the table formulation the
    rest of its series runs on a stretched grid (tables.build_box_tables decides on the whole axis), alone in a shard or not."""
    dom = synthetic_domain(3, 5, 20, 129, seed=29, dtype=np.float32, nonuniform_lon=True)
    boxes = [(64, 71, 8, 17), (55, 56, 5, 18), (8, 120, 1, 19)]
    eng = _engine(dom)
    f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    whole = eng.compute(*f, boxes, time_s=dom.time_s, keep_rows=True)
    for a, b in ((1, 2), (0, 2), (1, 3)):
        part = eng.compute(*f, boxes[a:b], time_s=dom.time_s, t_begin=a, t_count=b - a, keep_rows=True, per_step_boxes=True)
        ny = part.rows.shape[2]
        assert torch.equal(part.rows, whole.rows[a:b, :, :ny]) and torch.equal(part.scalars, whole.scalars[a:b]), (a, b)


def test_without_q_and_without_geopotential():
    dom = synthetic_domain(3, 5, 10, 128, seed=4)
    limits = (dom.lon[2], dom.lon[-2], dom.lat[1], dom.lat[-2])
    eng = _engine(dom)
    box = eng.box_from_limits(*limits)
    res = eng.compute(_dev(dom.tair), _dev(dom.u), _dev(dom.v), _dev(dom.omega), None, [box], with_q=False)
    ref_s, ref_l = o.lec_fixed(dom, *limits)
    got = res.scalars_dict()
    for name in ["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe"]:
        assert scale_err(got[name], ref_s[name]) <= TOL, name
    assert np.all(got["Gz"] == 0) and np.all(got["Ge"] == 0)


def test_geopotential_height_scale(golden_dir):
    """phi_scale = g turns geopotential height into geopotential (box_data.py:233-241)."""
    dom = synthetic_domain(3, 5, 10, 64, seed=6)
    limits = (dom.lon[1], dom.lon[-2], dom.lat[1], dom.lat[-2])
    eng = _engine(dom)
    box = eng.box_from_limits(*limits)
    hgt = dom.geopt / o.G
    res = eng.compute(_dev(dom.tair), _dev(dom.u), _dev(dom.v), _dev(dom.omega), _dev(hgt), [box],
                      time_s=dom.time_s, phi_scale=o.G)
    ref_s, _ = o.lec_fixed(dom, *limits)
    got = res.scalars_dict()
    for name in ["BΦZ", "BΦE"]:
        assert scale_err(got[name], ref_s[name]) <= TOL, name


def test_nan_levels_are_repaired_like_handle_nans():
    """A NaN level at one latitude row propagates to that level's tables; _handle_nans interpolates it."""
    dom = synthetic_domain(3, 7, 10, 64, seed=8)
    dom.u[:, 3, 4, 10] = np.nan
    limits = (dom.lon[1], dom.lon[-2], dom.lat[1], dom.lat[-2])
    res = run_fixed(dom, limits)
    assert int(res.nanflag.min()) > 0
    ref_s, _ = o.lec_fixed(dom, *limits)
    got = res.scalars_dict()
    for name in ["Kz", "Ke", "Ck", "BKz", "BKe", "Az", "Ae"]:
        assert np.isfinite(got[name]).all(), name
        assert scale_err(got[name], ref_s[name]) <= 1e-8, name


def test_static_stability_clamp():
    """sigma = {[g T / cp - (p g / Rd) dT/dp]} is clamped to >= 0.03 (thermodynamics.py:62-70; the reference says this is
    what keeps Az / Ca stable).  A block of levels with T ~ p^0.35 (steeper than the adiabat p^(2/7)) has sigma < 0 there:
    the clamp decides Az, Ae, Ca, BAz, BAe, Gz, Ge on those levels."""
    dom = synthetic_domain(4, 13, 12, 96, seed=31)
    for k in range(4, 9):
        dom.tair[:, k] = (300.0 * (dom.level[k] / 1e5) ** 0.35 + (dom.tair[:, k] - dom.tair[:, k].mean())).astype(dom.tair.dtype)
    limits = (dom.lon[1], dom.lon[-2], dom.lat[1], dom.lat[-2])
    box = o.make_box(dom, *limits)
    clamped = box.sigma_AA == 0.03
    assert clamped.any() and not clamped.all(), "the test data must drive some (time, level) onto the clamp and leave others free"
    check_fixed(dom, limits, what="sigma clamp")


@pytest.mark.parametrize("case", ["bottom_T_and_omega", "interior_omega_rows", "repair_reaches_the_boundary", "interior_T_patch",
                                  "column_to_the_ground"])
def test_nans_in_T_and_omega_follow_handle_nans(case):
    """Below-ground style NaNs in T and omega (not only in the winds): every one of the 16 terms and every level table against
    the oracle.  `interior_omega_rows` is the case where the reference's ORDER matters: BAz's bottom-top term is repaired per
    latitude before the area mean and divided by sigma afterwards (boundary_terms.py:165-176)."""
    dom = synthetic_domain(5, 8, 12, 64, seed=41)
    nl = dom.level.size
    if case == "bottom_T_and_omega":          # a patch of the lowest level at two time steps: nothing below to interpolate from -> level dropped
        dom.tair[1:3, nl - 1, 3:6, 10:20] = np.nan
        dom.omega[1:3, nl - 1, 3:6, 10:20] = np.nan
    elif case == "interior_omega_rows":       # omega only, an interior level, some latitude rows, one time step
        dom.omega[2, 4, 4:7, 5:9] = np.nan
    elif case == "repair_reaches_the_boundary":
        # the lowest level is NaN at some latitudes (-> dropped), the level above it at OTHER latitudes: there the per-latitude
        # repair interpolates between its neighbours (the lowest level is valid at those latitudes) and the repaired level becomes
        # the bottom of BAz's bottom-top difference -- repairing after the area mean gives another BAz (20 % off in that term)
        dom.omega[2, nl - 2, 4:7, 5:9] = np.nan
        dom.omega[2, nl - 1, 8:10, 20:24] = np.nan
    elif case == "interior_T_patch":          # T at an interior level: [T] and {[T]} are NaN there, the level is interpolated
        dom.tair[3, 3, 5, 30] = np.nan
    else:                                     # a column that is NaN from level 5 to the ground at a few points, every time step
        for f in (dom.tair, dom.omega, dom.u, dom.v, dom.geopt):
            f[:, 5:, 2:4, 40:44] = np.nan
    limits = (dom.lon[1], dom.lon[-2], dom.lat[1], dom.lat[-2])
    res = run_fixed(dom, limits)
    assert int(res.nanflag.max()) > 0
    with np.errstate(invalid="ignore"):
        ref_s, ref_l = o.lec_fixed(dom, *limits)
    worst = compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, 1e-9, f"NaN case {case}", time_s=dom.time_s)
    got = res.scalars_dict()
    assert all(np.isfinite(got[k]).all() for k in SCALARS), "every integrated term survives: levels were repaired or dropped"
    print(case, max(worst.values()))


def test_a_term_with_no_level_left_integrates_to_zero():
    """T is NaN at one point of levels 1 and 3 of one time step (5 levels): dT*/dp spreads that to every level of Ca's, Gz's and Ge's
    functions of level, the any-time mask drops them all for the whole series, and the reference then integrates an empty array:
    0.0 (xarray), not NaN -- found by tests/soak_streamed.py with scattered fill values.  The other terms keep their repaired levels."""
    dom = synthetic_domain(4, 5, 12, 64, seed=43)
    dom.tair[1, 1, 4, 20] = np.nan
    dom.tair[1, 3, 7, 30] = np.nan
    limits = (dom.lon[1], dom.lon[-2], dom.lat[1], dom.lat[-2])
    res = run_fixed(dom, limits)
    with np.errstate(invalid="ignore"):
        ref_s, ref_l = o.lec_fixed(dom, *limits)
    compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, 1e-9, "no level left", time_s=dom.time_s)
    got = res.scalars_dict()
    assert all(np.all(np.asarray(got[k]) == 0.0) for k in ("Ca", "Gz", "Ge")) and all(np.all(np.asarray(ref_s[k]) == 0.0) for k in ("Ca", "Gz", "Ge"))
    assert all(np.isfinite(got[k]).all() and np.abs(got[k]).min() > 0 for k in ("Az", "Ae", "Kz", "Ke", "Cz", "Ce", "Ck"))


def test_polar_rows_off_the_pole():
    """A latitude axis that comes within 1/8 degree of the pole without touching it: [u] / cos(phi), tan(phi) and
    dx ~ cos(phi) are large but finite in the reference, and the engine must follow them."""
    dom = synthetic_domain(3, 5, 64, 128, seed=51, lat0=-89.875, lat1=-74.125, lon0=-180.0, lon1=-148.25)
    limits = (dom.lon[0], dom.lon[-1], dom.lat[0], dom.lat[-1])
    check_fixed(dom, limits, tol=1e-8, what="polar rows")
    domn = synthetic_domain(3, 5, 64, 128, seed=52, lat0=74.125, lat1=89.875, lon0=100.0, lon1=131.75)
    check_fixed(domn, (domn.lon[0], domn.lon[-1], domn.lat[0], domn.lat[-1]), tol=1e-8, what="polar rows north")


def test_nan_at_top_level_drops_that_level_for_every_time_step():
    """xarray's dropna(dim=level) works on the whole [time, level] array: a top level that is NaN at ONE time
    step (nothing above it to interpolate from) leaves the pressure integrals of ALL time steps
    (energy_contents.py:203-207).  The moving framework builds one BoxData per time step, so there only
    that step loses the level."""
    dom = synthetic_domain(4, 7, 10, 64, seed=12)
    dom.v[2, 0, :, :] = np.nan                       # top level, one time step
    limits = (dom.lon[1], dom.lon[-2], dom.lat[1], dom.lat[-2])
    res = run_fixed(dom, limits)
    ref_s, _ = o.lec_fixed(dom, *limits)
    got = res.scalars_dict()
    for name in ["Kz", "Ke", "Ck", "BKz", "BKe"]:
        assert np.isfinite(got[name]).all(), name
        assert scale_err(got[name], ref_s[name]) <= 1e-9, name
    # per-time-step semantics (drop_any_time=False): the other time steps keep the level
    eng = _engine(dom)
    box = eng.box_from_limits(*limits)
    per = eng.compute(_dev(dom.tair), _dev(dom.u), _dev(dom.v), _dev(dom.omega), _dev(dom.geopt), [box],
                      time_s=dom.time_s, drop_any_time=False).scalars_dict()
    clean = synthetic_domain(4, 7, 10, 64, seed=12)
    ref_clean, _ = o.lec_fixed(clean, *limits)
    assert scale_err(per["Kz"][[0, 1, 3]], np.asarray(ref_clean["Kz"])[[0, 1, 3]]) <= 1e-9
    assert abs(per["Kz"][2] - ref_s["Kz"][2]) <= 1e-9 * abs(ref_s["Kz"][2])


# ---------------------------------------------------------------------------------------------
# argument checking mirrors the reference's error behaviour
# ---------------------------------------------------------------------------------------------
def test_long_rows_and_the_two_sweep_limit():
    """The default kernel walks a row in trips, so long rows just take more trips; the two-sweep cross-check
    kernel holds the row in registers and refuses rows beyond lec_max_row() (no silent truncation)."""
    from lorenzcycletoolkit_amd import _lib
    dom = synthetic_domain(2, 3, 4, 6000, seed=1)
    limits = (dom.lon[1], dom.lon[-2], dom.lat[0], dom.lat[-1])
    check_fixed(dom, limits, what="6000-point rows")
    assert _lib.load().lec_max_row(_lib.LEC_F64, 1, _lib.KERNEL_TWO_SWEEP) < 5000
    assert _lib.load().lec_max_row(_lib.LEC_F64, 1, _lib.KERNEL_AUTO) >= 6000
    with pytest.raises(_lib.LecLibraryError, match="longer than lec_max_row"):
        run_fixed(dom, limits, tuning={"kernel": "two_sweep"})


def test_tuning_is_validated():
    """Bad tuning values are refused with LEC_ERR_ARG (ValueError), never silently clamped."""
    dom = synthetic_domain(3, 4, 8, 64, seed=2)
    limits = (dom.lon[1], dom.lon[-2], dom.lat[1], dom.lat[-2])
    for bad in ({"tile_t": -1}, {"tile_j": -3}, {"block_shape": 111}, {"block_shape": 312}, {"f32_vec": 3}):
        with pytest.raises(ValueError):
            run_fixed(dom, limits, tuning=bad)
    with pytest.raises(ValueError):
        run_fixed(dom, limits, tuning={"kernel": "fastest"})
    with pytest.raises(ValueError, match="21 levels"):          # the box-tile kernel's level walk is bounded by its coefficient register
        run_fixed(dom, limits, tuning={"kernel": "box_tile", "tile_j": 22})


def test_errors():
    from lorenzcycletoolkit_amd.engine import LECEngine
    dom = synthetic_domain(3, 5, 10, 64, seed=9)
    eng = LECEngine(dom.lat, dom.lon, dom.level, device="cuda:0")
    with pytest.raises(ValueError):
        eng.box_from_limits(dom.lon[5], dom.lon[5], dom.lat[1], dom.lat[4])    # single-column box
    T = _dev(dom.tair)
    with pytest.raises(ValueError):
        eng.compute(T, T, T, T, T, [(0, 10, 0, 5)])                             # with_q but no time axis
    with pytest.raises(ValueError):
        eng.compute(T[:, :, :, :32], T, T, T, T, [(0, 10, 0, 5)], time_s=dom.time_s)
    with pytest.raises(ValueError):
        eng.compute(T, T, T, T, T, [(0, 70, 0, 5)], time_s=dom.time_s)          # box outside the grid


# ---------------------------------------------------------------------------------------------
# BASELINE.json size: a latitude band of the 37 x 721 x 1440 grid against the oracle, and
# size-independent properties on the whole grid
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_full_size_band_and_properties(dtype):
    from lorenzcycletoolkit_amd.engine import LECEngine
    from lorenzcycletoolkit_amd.synthetic import era5_grid, era5_like_levels, synthetic_cube
    lat, lon = era5_grid()
    level = era5_like_levels()
    nt = 3
    time_s = np.arange(nt) * 3600.0
    f = synthetic_cube(nt, level, lat, lon, device="cuda:0", dtype=dtype, seed=1234)
    eng = LECEngine(lat, lon, level, device="cuda:0")
    # (1) 15-degree band against the oracle
    box = eng.box_from_limits(-180, 179.75, -40.0, -25.0)
    res = eng.compute(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], [box], time_s=time_s)
    iw, ie, js, jn = box
    crop = lambda a: a[:, :, js:jn + 1, iw:ie + 1].double().cpu().numpy()
    dom = o.Domain(crop(f["tair"]), crop(f["u"]), crop(f["v"]), crop(f["omega"]), crop(f["geopt"]),
                   lat[js:jn + 1], lon[iw:ie + 1], level, time_s)
    ref_s, ref_l = o.lec_fixed(dom, lon[iw], lon[ie], lat[js], lat[jn])
    compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, 1e-8, "full-size band", time_s=dom.time_s)
    # (2) whole grid without the polar rows: finite, shard-invariant, and energy scaling:
    #     doubling u and v multiplies Kz, Ke, BKz, BKe by 4 / 4 / 8 / 8 exactly (powers of two).
    box = eng.box_from_limits(-180, 179.75, -89.75, 89.75)
    r1 = eng.compute(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], [box], time_s=time_s)
    s1 = r1.scalars_dict()
    assert all(np.isfinite(s1[k]).all() for k in SCALARS)
    r2 = eng.compute(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], [box], time_s=time_s, t_begin=1, t_count=1)
    assert torch.equal(r1.scalars[1:2], r2.scalars)
    r3 = eng.compute(f["tair"], f["u"] * 2, f["v"] * 2, f["omega"], f["geopt"], [box], time_s=time_s)
    s3 = r3.scalars_dict()
    assert np.array_equal(s3["Kz"], 4 * s1["Kz"]) and np.array_equal(s3["Ke"], 4 * s1["Ke"])
    assert np.array_equal(s3["Az"], s1["Az"]) and np.array_equal(s3["Ae"], s1["Ae"])
    # omega -> -omega: the conversions that are linear in omega (Cz, Ce) change sign exactly, the energies do not move,
    # and so do the level tables of those terms
    r4 = eng.compute(f["tair"], f["u"], f["v"], -f["omega"], f["geopt"], [box], time_s=time_s)
    s4 = r4.scalars_dict()
    for name in ("Az", "Ae", "Kz", "Ke"):
        assert np.array_equal(s4[name], s1[name]), name
    for name in ("Cz", "Ce"):
        assert np.array_equal(s4[name], -s1[name]), name
    l1, l4 = r1.levels_dict(), r4.levels_dict()
    assert np.array_equal(l4["Ce"], -l1["Ce"]) and np.array_equal(l4["Cz_2"], -l1["Cz_2"])


def test_full_size_zonally_symmetric_fields_have_no_eddy_energy():
    """Fields that do not depend on longitude: every eddy statistic vanishes exactly (T - [T] == 0 point by
    point), so Ae, Ke, Ca, Ce, Ck, Ge, BAe and BKe are exactly 0 on the whole 37 x 721 x 1440 grid while the
    zonal terms are not."""
    from lorenzcycletoolkit_amd.engine import LECEngine
    from lorenzcycletoolkit_amd.synthetic import era5_grid, era5_like_levels
    lat, lon = era5_grid()
    level = era5_like_levels()
    nt = 2
    dev = torch.device("cuda:0")
    p = torch.as_tensor(level, device=dev)[None, :, None, None]
    phi = torch.deg2rad(torch.as_tensor(lat, device=dev))[None, None, :, None]
    tt = torch.arange(nt, device=dev, dtype=torch.float64)[:, None, None, None]
    shape = (nt, level.size, lat.size, lon.size)
    mk = lambda a: a.expand(shape).contiguous()
    T = mk(288.0 * (p / 1e5) ** 0.19 + 10.0 * torch.cos(2 * phi) * (p / 1e5) + 0.5 * tt)
    u = mk(25.0 * torch.cos(phi) * (1 - p / 1.2e5) + 0 * tt)
    v = mk(0.3 * torch.cos(phi) * torch.sin(3 * phi) + 0 * p + 0 * tt)
    w = mk(0.02 * torch.sin(2 * phi) * (p / 1e5) + 0 * tt)
    ph = mk(9.80665 * 7000.0 * torch.log(1e5 / p) + 50.0 * torch.cos(phi) + 0 * tt)
    eng = LECEngine(lat, lon, level, device=dev)
    box = eng.box_from_limits(-180, 179.75, -89.75, 89.75)
    s = eng.compute(T, u, v, w, ph, [box], time_s=np.arange(nt) * 3600.0).scalars_dict()
    for name in ["Ae", "Ke", "Ca", "Ce", "Ck", "Ge", "BAe", "BKe"]:
        assert np.all(s[name] == 0.0), (name, s[name])
    for name in ["Az", "Kz", "Cz", "Gz"]:
        assert np.all(np.isfinite(s[name])) and np.all(s[name] != 0.0), name
