"""lec_boxplane.hip (LEC_KERNEL_BOX_PLANE): stage 1 of a box-packed fp64 series with the planes brought into LDS by DMA.

It replaces lec_boxtile.hip for that kind of call (what every -t path of the product hands stage 1 for fp64 data) and must give that
kernel's row records BIT for bit: the same products, rounding points, summation groups and order -- only where the operands come
from differs (global_load_lds pieces of contiguous plane rows, T's vertical neighbours one level back / ahead in registers).  The
box-tile kernel itself is held to the independent one-wave-per-row kernel and to the oracle elsewhere (test_gpu_parity.py,
test_gpu_packed.py); here: every record equal, across geometries, level chunks, slab paddings, NaNs and shards -- and against the
oracle once more (the reference's moving framework: lec_moving_framework.py:639-745, box_data.py:297-310)."""
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


def _series(dom, boxes, ny=None, nx=None):
    eng = _engine(dom)
    f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    tc = eng.time_coefs_device(dom.time_s)
    ps = eng.pack_series(*f, boxes, tc, ny=ny, nx=nx)
    return eng, f, tc, ps, eng.prepare_boxes(boxes, packed=True)


def _dkw(ps, tc):
    """dT/dt as the packed series carries it: a cube (fp64 storage), or T of the two time neighbours + the time coefficients (fp32)."""
    return {"dTdt": ps["dTdt"]} if "dTdt" in ps else {"tm": ps["tm"], "tp": ps["tp"], "tcoef": tc}


def _rows(eng, ps, pb, kernel, t0=0, t1=None, tc=None, **tuning):
    t1 = len(pb) if t1 is None else t1
    r = eng.rowstats(ps["tair"], ps["u"], ps["v"], ps["omega"], ps["geopt"], pb.part(t0, t1), t_begin=t0, t_count=t1 - t0,
                     per_step_boxes=True, tuning=dict(kernel=kernel, **tuning), **_dkw(ps, tc))
    torch.cuda.synchronize()
    return r


def test_fp32_series_with_time_neighbours_and_with_a_dtdt_cube():
    """fp32 storage: the packed series of the moving framework carries T of the two time neighbours (no dT/dt cube: a float32 one would
    not be the engine's fp64 dT/dt) -- every record of lec_boxtile's, shards too; and a caller's own float32 dT/dt cube is served as well."""
    dom = synthetic_domain(9, 17, 70, 90, seed=32, dtype=np.float32)
    boxes = [(10 + t, 70 + t, 5 + (t // 2), 60 + (t // 3)) for t in range(9)]
    eng, f, tc, ps, pb = _series(dom, boxes)
    assert "tm" in ps and ps["tair"].dtype == torch.float32
    ref = _rows(eng, ps, pb, "box_tile", tc=tc)
    for tj in (0, 5, 9, 17):
        assert torch.equal(_rows(eng, ps, pb, "box_plane", tc=tc, tile_j=tj), ref), tj
    auto = eng.rowstats(ps["tair"], ps["u"], ps["v"], ps["omega"], ps["geopt"], pb, per_step_boxes=True, **_dkw(ps, tc))
    assert torch.equal(auto, ref) and torch.isfinite(ref).all()
    assert torch.equal(_rows(eng, ps, pb, "box_plane", 2, 7, tc=tc), ref[2:7])
    d32 = _dev(o.moving_dTdt(dom).astype(np.float32))
    pk = {k: eng.pack_boxes(c, boxes) for k, c in zip(("tair", "u", "v", "omega", "geopt"), f)}
    dp = eng.pack_boxes(d32, boxes)
    kw = dict(dTdt=dp, per_step_boxes=True)
    a = eng.rowstats(pk["tair"], pk["u"], pk["v"], pk["omega"], pk["geopt"], pb, tuning={"kernel": "box_tile"}, **kw)
    b = eng.rowstats(pk["tair"], pk["u"], pk["v"], pk["omega"], pk["geopt"], pb, tuning={"kernel": "box_plane"}, **kw)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_config5_shape_bit_identical_and_against_the_oracle():
    """BASELINE config 5's boxes: 61 x 61 points, 37 levels, a track that moves a column or a row per step -- 11 steps.  Both kernels,
    every level-chunk length the launch rule can pick (37 levels: 5 .. 21 per wave), the automatic choice, and a shard."""
    dom = synthetic_domain(11, 37, 100, 140, seed=61)
    boxes = [(20 + t, 80 + t, 15 + (t // 2), 75 + (t // 2)) for t in range(11)]
    eng, f, tc, ps, pb = _series(dom, boxes)
    ref = _rows(eng, ps, pb, "box_tile")
    assert torch.isfinite(ref).all()
    for tj in (0, 5, 8, 13, 19, 21, 22, 30, 37):              # (the box-plane kernel walks up to 42 levels per wave: 37 = one walk)
        got = _rows(eng, ps, pb, "box_plane", tile_j=tj)
        assert torch.equal(got, ref), f"tile_j={tj}"
    auto = eng.rowstats(ps["tair"], ps["u"], ps["v"], ps["omega"], ps["geopt"], pb, dTdt=ps["dTdt"], per_step_boxes=True)
    assert torch.equal(auto, ref)
    assert torch.equal(_rows(eng, ps, pb, "box_plane", 3, 8), ref[3:8]) and torch.equal(_rows(eng, ps, pb, "box_plane", 10, 11), ref[10:11])
    res = eng.reduce(auto, pb, drop_any_time=False)
    limits = [(dom.lon[iw], dom.lon[ie], dom.lat[js], dom.lat[jn]) for iw, ie, js, jn in boxes]
    ref_s, ref_l = o.lec_moving(dom, limits)
    compare(res.scalars_dict(), res.levels_dict(), ref_s, ref_l, 1e-9, "box-plane kernel vs oracle", time_s=dom.time_s)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_random_geometries_bit_identical(dtype):
    """Boxes from 2 x 2 to 64 x 90 points anywhere in the grid, 2 .. 45 levels, 2 .. 19 steps with another box each, slabs with and
    without room to spare (odd and even pitches: the runs then start on 4-, 8- and 16-byte boundaries), random level chunks.  fp64
    storage: dT/dt as a cube; fp32 storage: T of the two time neighbours as cubes of their own -- the box-packed series the product's
    -t paths hand over for either kind of data."""
    rng = np.random.default_rng(606)
    for case in range(40):
        nt, nl = int(rng.integers(2, 20)), int(rng.integers(2, 46))
        ny, nx = int(rng.integers(8, 110)), int(rng.integers(8, 90))
        dom = synthetic_domain(nt, nl, ny, nx, seed=6000 + case, dtype=dtype)
        boxes = []
        for _ in range(nt):
            wx, wy = int(rng.integers(2, min(nx, 64) + 1)), int(rng.integers(2, min(ny, 90) + 1))
            iw, js = int(rng.integers(0, nx - wx + 1)), int(rng.integers(0, ny - wy + 1))
            boxes.append((iw, iw + wx - 1, js, js + wy - 1))
        wmax, hmax = max(b[1] - b[0] + 1 for b in boxes), max(b[3] - b[2] + 1 for b in boxes)
        pad = case % 3
        eng, f, tc, ps, pb = _series(dom, boxes, ny=min(ny, hmax + pad), nx=min(nx, 64, wmax + pad))
        tj = int(rng.integers(0, min(nl, 42) + 1))
        ref, got = _rows(eng, ps, pb, "box_tile", tc=tc, tile_j=min(tj, 21)), _rows(eng, ps, pb, "box_plane", tc=tc, tile_j=tj)
        assert torch.equal(got, ref), f"case {case}: nt={nt} nl={nl} grid {ny}x{nx} slab {tuple(ps['tair'].shape[2:])} tile_j={tj} boxes {boxes}"
        for t, bx in enumerate(boxes):
            assert torch.all(got[t, :, bx[3] - bx[2] + 1:] == 0), (case, t)


def test_boxes_anywhere_in_a_narrow_grid_with_a_supplied_dtdt_cube():
    """Not only packed series: any per-step-box call with a dT/dt cube whose cubes are at most 64 columns wide (a caller's own dT/dt
    on a small regional grid) -- the boxes then sit anywhere in the planes, rows are cut out of wider rows."""
    dom = synthetic_domain(6, 12, 50, 64, seed=17)
    boxes = [(3 + t, 40 + 2 * t, 5 + t, 30 + 3 * t) for t in range(6)]
    eng = _engine(dom)
    f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    d = _dev(o.moving_dTdt(dom))
    kw = dict(dTdt=d, per_step_boxes=True)
    ref = eng.rowstats(*f, boxes, tuning={"kernel": "box_tile"}, **kw)
    got = eng.rowstats(*f, boxes, tuning={"kernel": "box_plane"}, **kw)
    auto = eng.rowstats(*f, boxes, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got, ref) and torch.equal(auto, ref)
    sweep = eng.rowstats(*f, boxes, tuning={"kernel": "row_sweep"}, **kw)       # the independent formulation: to rounding
    torch.cuda.synchronize()
    err = ((got[..., :28] - sweep[..., :28]).abs().amax(dim=(0, 1, 2)) / sweep[..., :28].abs().amax(dim=(0, 1, 2)).clamp_min(1e-300)).max()
    assert float(err) <= 1e-10


def test_nans_travel_as_in_the_box_tile_kernel():
    """NaN inside a box (a below-ground patch, a single point, a whole row) reaches the same records; NaN in the slab's padding and in
    the rows / columns beside a box reaches none."""
    dom = synthetic_domain(5, 9, 40, 60, seed=33)
    boxes = [(4 + t, 50 + t, 3, 33 - t) for t in range(5)]
    dom.tair[1, 7:, 10:14, 20:30] = np.nan
    dom.omega[2, 3, 12, 25] = np.nan
    dom.u[3, :, 8, :] = np.nan
    eng, f, tc, ps, pb = _series(dom, boxes, ny=36, nx=52)
    ref, got = _rows(eng, ps, pb, "box_tile"), _rows(eng, ps, pb, "box_plane")
    assert torch.isnan(ref).any() and torch.equal(torch.isnan(got), torch.isnan(ref))
    assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(ref, nan=-7.0))
    clean = synthetic_domain(5, 9, 40, 60, seed=33)
    eng, f, tc, ps, pb = _series(clean, boxes, ny=36, nx=52)
    ref = _rows(eng, ps, pb, "box_plane")
    for t, (iw, ie, js, jn) in enumerate(boxes):
        for c in [ps[k] for k in ("tair", "u", "v", "omega", "geopt", "dTdt")]:
            c[t, :, jn - js + 1:, :] = float("nan")
            c[t, :, :, ie - iw + 1:] = float("nan")
    assert torch.equal(_rows(eng, ps, pb, "box_plane"), ref)


def test_box_plane_refuses_what_it_does_not_serve():
    """Asked for by name the kernel must run or say no -- never another kernel silently: stretched longitudes, cubes wider than 64
    columns, dT/dt from the cube's own time axis, fp64 time neighbours, a fixed box."""
    dom = synthetic_domain(3, 4, 20, 30, seed=1)
    boxes = [(2, 20, 3, 15)] * 3
    eng, f, tc, ps, pb = _series(dom, boxes)
    with pytest.raises(ValueError, match="BOX_PLANE"):
        eng.rowstats(*f, boxes, time_s=dom.time_s, per_step_boxes=True, tuning={"kernel": "box_plane"})             # dT/dt per point from the cube
    with pytest.raises(ValueError, match="BOX_PLANE"):
        eng.rowstats(*f, [boxes[0]], time_s=dom.time_s, tuning={"kernel": "box_plane"})                              # one fixed box
    wide = synthetic_domain(3, 4, 20, 80, seed=1)
    eng2 = _engine(wide)
    f2 = [_dev(a) for a in (wide.tair, wide.u, wide.v, wide.omega, wide.geopt)]
    with pytest.raises(ValueError, match="BOX_PLANE"):
        eng2.rowstats(*f2, boxes, dTdt=_dev(o.moving_dTdt(wide)), per_step_boxes=True, tuning={"kernel": "box_plane"})
    # fp64 storage with the time neighbours as operands (six 32-byte operands per prefetch set): lec_boxtile's
    tm, tp = eng.pack_boxes(f[0], boxes, shift=-1), eng.pack_boxes(f[0], boxes, shift=1)
    with pytest.raises(ValueError, match="BOX_PLANE"):
        eng.rowstats(ps["tair"], ps["u"], ps["v"], ps["omega"], ps["geopt"], pb, tm=tm, tp=tp, tcoef=tc, per_step_boxes=True, tuning={"kernel": "box_plane"})
    st = synthetic_domain(3, 4, 20, 30, seed=1, nonuniform_lon=True)
    eng4 = _engine(st)
    f4 = [_dev(a) for a in (st.tair, st.u, st.v, st.omega, st.geopt)]
    with pytest.raises(ValueError, match="BOX_PLANE"):
        eng4.rowstats(*f4, boxes, dTdt=_dev(o.moving_dTdt(st)), per_step_boxes=True, tuning={"kernel": "box_plane"})
    big = synthetic_domain(2, 30, 12, 16, seed=2)
    e5, f5, tc5, ps5, pb5 = _series(big, [(1, 12, 1, 9)] * 2)
    assert torch.equal(_rows(e5, ps5, pb5, "box_plane", tile_j=30), _rows(e5, ps5, pb5, "box_tile", tile_j=15))     # (one walk of all 30 levels)
    with pytest.raises(ValueError, match="42"):
        _rows(e5, ps5, pb5, "box_plane", tile_j=43)


def test_a_long_series_takes_every_xcd_chunk_and_level_walk():
    """600 steps of 61 x 61 x 37 (the launch rule's longest level walk, eight XCD chunks of 75 steps): both kernels, record for record."""
    dom = synthetic_domain(600, 37, 66, 70, seed=5)
    boxes = [(int(4 + 3 * np.sin(t / 40.0)), int(64 + 3 * np.sin(t / 40.0)), int(2 + 2 * np.cos(t / 55.0)), int(62 + 2 * np.cos(t / 55.0))) for t in range(600)]
    eng, f, tc, ps, pb = _series(dom, boxes)
    ref, got = _rows(eng, ps, pb, "box_tile"), _rows(eng, ps, pb, "box_plane")
    assert torch.equal(got, ref) and torch.isfinite(got).all()
