"""Box-packed series of the moving framework (include/lec_hip.h, lec_rowstats_args.tm_d / tp_d): cubes that hold, per time step, only
that step's box (at the origin of its slab) plus T of the previous / next step on that box.  The reference slices every step's box out
of the data before it computes anything (src/utils/box_data.py:297-310); here a producer that gathers anyway hands over just the
slices.  Same arithmetic on the same values: the row records -- and so every term -- must be BIT-identical to the unpacked cubes'."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import lec_oracle as o
from tests.helpers import compare, synthetic_domain

pytestmark = pytest.mark.gpu


def _engine(dom):
    from lorenzcycletoolkit_amd.engine import LECEngine
    return LECEngine(dom.lat, dom.lon, dom.level, device="cuda:0")


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")


def _both(dom, boxes, ny=None, nx=None, t_range=None):
    """(rows of the unpacked cubes, rows of the packed series) for steps t_range of a moving series."""
    eng = _engine(dom)
    f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    nt = dom.tair.shape[0]
    t0, t1 = t_range or (0, nt)
    tc = eng.time_coefs_device(dom.time_s)
    plain = eng.prepare_boxes(boxes)
    a = eng.rowstats(*f, plain.part(t0, t1), tcoef=tc, t_begin=t0, t_count=t1 - t0, per_step_boxes=True)
    packed_boxes = eng.prepare_boxes(boxes, packed=True)
    pk = [eng.pack_boxes(c, boxes, ny=ny, nx=nx) for c in f]
    tm, tp = eng.pack_boxes(f[0], boxes, shift=-1, ny=ny, nx=nx), eng.pack_boxes(f[0], boxes, shift=+1, ny=ny, nx=nx)
    b = eng.rowstats(*pk, packed_boxes.part(t0, t1), tcoef=tc, t_begin=t0, t_count=t1 - t0, per_step_boxes=True, tm=tm, tp=tp)
    # ... and as pack_series hands it over: fp64 storage with dT/dt as a cube of its own (lec_dtdt), fp32 with the two neighbours
    ps = eng.pack_series(*f, boxes, tc, ny=ny, nx=nx)
    assert ("dTdt" in ps) == (dom.tair.dtype == np.float64) and ("tm" in ps) == (dom.tair.dtype == np.float32)
    extra = {k: ps[k] for k in ("dTdt", "tm", "tp") if k in ps}
    c = eng.rowstats(ps["tair"], ps["u"], ps["v"], ps["omega"], ps["geopt"], packed_boxes.part(t0, t1), t_begin=t0, t_count=t1 - t0, per_step_boxes=True,
                     **({"tcoef": tc} if "tm" in ps else {}), **extra)
    torch.cuda.synchronize()
    assert torch.equal(c, b), "pack_series (dT/dt cube / neighbours) against the explicit neighbours"
    return eng, a, b, (pk, tm, tp, packed_boxes, tc)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nonuni", [False, True])
def test_packed_series_gives_the_bits_of_the_cube(dtype, nonuni):
    """61 x 61 boxes wandering over a 100 x 140 grid (a column or a row per step, as a track does), 9 levels, 7 steps: every record
    equal, whole series and a shard in its middle; the terms agree with the oracle like the unpacked path's."""
    dom = synthetic_domain(7, 9, 100, 140, seed=5, dtype=dtype, nonuniform_lon=nonuni)
    boxes = [(20 + t, 80 + t, 15 + (t // 2), 75 + (t // 2)) for t in range(7)]
    eng, a, b, (pk, tm, tp, pb, tc) = _both(dom, boxes)
    assert torch.equal(a, b) and torch.isfinite(b).all()
    _, a2, b2, _ = _both(dom, boxes, t_range=(2, 5))
    assert torch.equal(a2, a[2:5]) and torch.equal(b2, a[2:5])
    if dtype == np.float64 and not nonuni:
        res = eng.reduce(b, pb, drop_any_time=False)
        limits = [(dom.lon[iw], dom.lon[ie], dom.lat[js], dom.lat[jn]) for iw, ie, js, jn in boxes]
        ref_s, ref_l = o.lec_moving(dom, limits)
        compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, 1e-9, "packed series vs oracle", time_s=dom.time_s)


def test_pack_boxes_is_the_products_gather():
    """LECEngine.pack_boxes runs lec_ingest with per-step origins (the gather the streamed moving framework packs its series with):
    inside every step's box the values of the cube -- held here to an independent torch gather --, both dtypes, time-shifted too,
    NaN and -0.0 as they are, boxes at the grid's edges gathered with a slab that reaches past the grid."""
    dom = synthetic_domain(6, 4, 23, 31, seed=12)
    dom.tair[2, 1, 5:9, 4:7] = np.nan
    dom.tair[3, 0, 7, 7] = -0.0
    boxes = [(0, 9, 0, 5), (21, 30, 17, 22), (3, 30, 0, 22), (5, 12, 4, 19), (0, 30, 0, 22), (29, 30, 21, 22)]
    for dtype in (np.float64, np.float32):
        eng = _engine(dom)
        cube = _dev(dom.tair.astype(dtype))
        for shift in (0, -1, 1):
            for ny, nx in ((None, None), (23, 31)):
                got, ref = eng.pack_boxes(cube, boxes, shift=shift, ny=ny, nx=nx), eng.pack_boxes_by_indexing(cube, boxes, shift=shift, ny=ny, nx=nx)
                assert got.shape == ref.shape and got.dtype == ref.dtype
                for t, (iw, ie, js, jn) in enumerate(boxes):
                    a, b = got[t, :, :jn - js + 1, :ie - iw + 1], ref[t, :, :jn - js + 1, :ie - iw + 1]
                    assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a, nan=1.0).view(torch.int64 if dtype == np.float64 else torch.int32),
                                                                                       torch.nan_to_num(b, nan=1.0).view(torch.int64 if dtype == np.float64 else torch.int32)), (t, shift)
    with pytest.raises(ValueError, match="tallest"):
        eng.pack_boxes(cube, boxes, ny=10)


def test_packed_series_random_geometries():
    """Boxes of any size anywhere in the grid (also at its edges, also wider than one 64-column pass), slabs larger than the boxes,
    1..6 steps, fp32 / fp64, uniform or stretched longitudes: records equal bit for bit, padding rows of the lower boxes zero."""
    rng = np.random.default_rng(77)
    for case in range(30):
        nt, nl = int(rng.integers(1, 7)), int(rng.integers(2, 14))
        ny, nx = int(rng.integers(8, 90)), int(rng.integers(8, 170))
        dtype = np.float32 if rng.random() < 0.3 else np.float64
        nonuni = bool(rng.random() < 0.3)
        if nt < 2:
            nt = 2                                   # dT/dt from time neighbours needs two steps in the series
        dom = synthetic_domain(nt, nl, ny, nx, seed=3000 + case, dtype=dtype, nonuniform_lon=nonuni)
        boxes = []
        for _ in range(nt):
            wx, wy = int(rng.integers(2, min(nx, 150) + 1)), int(rng.integers(2, min(ny, 80) + 1))
            iw, js = int(rng.integers(0, nx - wx + 1)), int(rng.integers(0, ny - wy + 1))
            boxes.append((iw, iw + wx - 1, js, js + wy - 1))
        pad = case % 3 == 0                          # slabs with room to spare (a 64-column pitch for 61-column boxes, say)
        wmax, hmax = max(b[1] - b[0] + 1 for b in boxes), max(b[3] - b[2] + 1 for b in boxes)
        eng, a, b, _ = _both(dom, boxes, ny=min(ny, hmax + 3) if pad else None, nx=min(nx, wmax + 3) if pad else None)
        assert torch.equal(a, b), f"case {case}: nt={nt} nl={nl} grid {ny}x{nx} boxes {boxes} {np.dtype(dtype).name} nonuni={nonuni} pad={pad}"
        for t, bx in enumerate(boxes):
            assert torch.all(b[t, :, bx[3] - bx[2] + 1:] == 0), (case, t)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_one_step_of_a_packed_series_is_a_series(dtype):
    """A packed cube brings its time neighbours along, so ONE step can be handed over alone (the last rank of a time-sharded track
    holds a single step; found by the three-rank CLI test): the records of that step of the whole series."""
    dom = synthetic_domain(4, 5, 30, 50, seed=21, dtype=dtype)
    boxes = [(3 + t, 40 + t, 2, 25) for t in range(4)]
    eng, a, b, (pk, tm, tp, pb, tc) = _both(dom, boxes)
    for t in (0, 2, 3):
        one = eng.rowstats(*[c[t:t + 1].contiguous() for c in pk], pb.part(t, t + 1), tcoef=tc[t:t + 1].contiguous(), t_begin=0, t_count=1, per_step_boxes=True,
                           tm=tm[t:t + 1].contiguous(), tp=tp[t:t + 1].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(one[0], a[t]), t


def test_nan_beside_the_box_does_not_leak_in():
    """What lies outside a step's box never enters its sums: NaNs all around the boxes in the cube, and NaN-filled slab padding in
    the packed series, change nothing (the one-sided stencils at a box's edge multiply nothing by 0)."""
    dom = synthetic_domain(4, 5, 40, 90, seed=9)
    boxes = [(10 + t, 70 + t, 5, 30) for t in range(4)]
    eng, a, b, (pk, tm, tp, pb, tc) = _both(dom, boxes, ny=30, nx=64)
    for c in pk + [tm, tp]:
        c[:, :, 26:, :] = float("nan")
        c[:, :, :, 61:] = float("nan")
    b2 = eng.rowstats(*pk, pb, tcoef=tc, t_begin=0, t_count=4, per_step_boxes=True, tm=tm, tp=tp)
    torch.cuda.synchronize()
    assert torch.equal(b2, a)


def test_packed_arguments_are_checked():
    dom = synthetic_domain(3, 4, 20, 30, seed=1)
    boxes = [(2, 20, 3, 15)] * 3
    eng, a, b, (pk, tm, tp, pb, tc) = _both(dom, boxes)
    kw = dict(tcoef=tc, t_begin=0, t_count=3, per_step_boxes=True)
    with pytest.raises(ValueError, match="box-packed"):
        eng.rowstats(*pk, pb, tm=tm, **kw)                                           # one neighbour cube only
    with pytest.raises(ValueError, match="box-packed"):
        eng.rowstats(*pk, eng.prepare_boxes(boxes), tm=tm, tp=tp, **kw)              # boxes not prepared for a packed series
    with pytest.raises(ValueError, match="box-tile"):
        eng.rowstats(*pk, pb, tm=tm, tp=tp, tuning={"kernel": "row_sweep"}, **kw)    # another kernel family would read the wrong neighbours
    with pytest.raises(ValueError, match="box-tile"):
        eng.rowstats(*pk, pb, tm=tm, tp=tp, tuning={"kernel": "box_tile", "block_shape": 2}, **kw)      # time groups share grid rows: not packed
    with pytest.raises(ValueError, match="tallest"):
        eng.rowstats(*[c[:, :, :5] .contiguous() for c in pk], pb, tm=tm[:, :, :5].contiguous(), tp=tp[:, :, :5].contiguous(), **kw)
