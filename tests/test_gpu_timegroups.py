"""Time groups of the box-tile kernel (lec_boxtile.hip, template parameter TG): a workgroup of TG waves owns the same four rows of TG
consecutive time steps and hands T(t - 1) / T(t + 1) from wave to wave through LDS instead of loading them.  The waves of a group
load on the rows and columns of the UNION of their boxes, while everything that shapes a sum stays box-relative -- so the row records
must not depend on TG, on whether a group's boxes fit one union (64 columns, the launch's row blocks), on how a series is cut into
groups, shards or chunks, or on what lies just outside a box: bit for bit against TG = 1, and to rounding against the independent
one-wave-per-row kernel."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from lorenzcycletoolkit_amd.engine import LECEngine          # noqa: E402
from tests.helpers import synthetic_domain                    # noqa: E402

DEV = "cuda:0"


def _same(a, b):
    a, b = a[..., :28], b[..., :28]
    return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())


def _rows(eng, f, boxes, time_s, tg, t_begin=0, t_count=None, nyb=None):
    """Row records + terms of steps [t_begin, t_begin + t_count) with time groups of `tg` (0: the one-wave-per-row kernel).  The record
    buffer is handed over full of NaN: a row the kernel fails to write -- a padding row of a lower box, say -- shows."""
    tuning = {"kernel": "box_tile", "block_shape": tg} if tg else {"kernel": "row_sweep"}
    t_count = len(boxes) if t_count is None else t_count
    nyb = max(b[3] - b[2] + 1 for b in boxes) if nyb is None else nyb
    prep = eng.prepare_boxes(boxes, nyb_min=nyb)
    rows = torch.full((t_count, eng.level.size, nyb, 32), float("nan"), dtype=torch.float64, device=DEV)
    eng.rowstats(*f, prep, time_s=time_s, per_step_boxes=True, tuning=tuning, rows_out=rows, t_begin=t_begin, t_count=t_count)
    return eng.reduce(rows, prep, drop_any_time=False, keep_rows=True)


def _track(nt, nx, ny, w, h, rng, jumps=()):
    """Boxes of w x h points that drift by 0..2 columns / rows a step, with sudden jumps at the steps in `jumps`."""
    i, j = 5, 4
    out = []
    for t in range(nt):
        if t in jumps:
            i, j = int(rng.integers(0, nx - w)), int(rng.integers(0, ny - h))
        else:
            i = int(np.clip(i + rng.integers(-1, 3), 0, nx - w))
            j = int(np.clip(j + rng.integers(-1, 2), 0, ny - h))
        out.append((i, i + w - 1, j, j + h - 1))
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nonuniform", [False, True])
def test_records_do_not_depend_on_the_time_group(dtype, nonuniform):
    rng = np.random.default_rng(11)
    nt, nl, ny, nx = 23, 9, 80, 100
    dom = synthetic_domain(nt, nl, ny, nx, seed=3, dtype=dtype, lat0=-60, lat1=19, lon0=-100, lon1=-1, nonuniform_lon=nonuniform)
    eng = LECEngine(dom.lat, dom.lon, dom.level, device=DEV)
    cases = {
        "61x61 drifting (unions of 64 fit, some do not)": _track(nt, nx, ny, 61, 61, rng),
        "50x45 drifting with jumps (groups that cannot share)": _track(nt, nx, ny, 50, 45, rng, jumps=(6, 7, 15)),
        "one box for every step": [(10, 70, 8, 68)] * nt,
        "two-row, three-column boxes": _track(nt, nx, ny, 3, 2, rng),
        "boxes of changing size": [(4 + t % 3, 40 + 2 * (t % 11), 3 + t % 2, 30 + 3 * (t % 7)) for t in range(nt)],
        "64 columns wide, 64 rows high": [(20 + (t % 2), 83 + (t % 2), 10, 73) for t in range(nt)],
        # (found by the soak: a LOW box that starts far below the top of its group's union -- its padding rows, which stage 2 reads
        # as zeros, must still lie inside the launch's row blocks, else the group may not share)
        "a low box far down the union": [(7, 13, 41, 53) if t % 2 == 0 else (3, 9, 45, 46) for t in range(nt)],
        "boxes of very different heights": [[(13, 30, 8, 42), (0, 30, 6, 56), (16, 30, 6, 23), (4, 17, 23, 40)][t % 4] for t in range(nt)],
    }
    for what, boxes in cases.items():
        f = [torch.as_tensor(np.ascontiguousarray(x)).to(DEV) for x in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
        one = _rows(eng, f, boxes, dom.time_s, 1)
        ind = _rows(eng, f, boxes, dom.time_s, 0)                                 # the one-wave-per-row kernel: an independent formulation
        den = ind.rows[..., :28].abs().amax(dim=(0, 1, 2)).clamp_min(1e-300)
        assert float(((one.rows[..., :28] - ind.rows[..., :28]).abs().amax(dim=(0, 1, 2)) / den).max()) < 1e-10, what
        for tg in (2, 4):
            got = _rows(eng, f, boxes, dom.time_s, tg)
            assert _same(got.rows, one.rows), (what, tg)
            assert torch.equal(got.scalars, one.scalars) and torch.equal(got.levels, one.levels), (what, tg)
            # a shard that cuts the groups elsewhere (steps 3..13 of the series, its own one-step halo)
            part = _rows(eng, f, boxes[3:14], dom.time_s, tg, t_begin=3, t_count=11, nyb=int(one.rows.shape[2]))
            assert _same(part.rows, one.rows[3:14]), (what, tg, "shard")


def test_what_lies_outside_a_box_does_not_reach_its_records():
    """NaN just outside every box -- the rows above and below it and the columns either side, at the step's own time and at its
    time neighbours' -- where a sharing group loads the union's rows: the one-sided stencils give those points the coefficient 0,
    and 0 x NaN must not be formed."""
    rng = np.random.default_rng(5)
    nt, nl, ny, nx = 14, 6, 70, 90
    dom = synthetic_domain(nt, nl, ny, nx, seed=8, dtype=np.float64, lat0=-60, lat1=9, lon0=-100, lon1=-11)
    boxes = _track(nt, nx - 4, ny - 4, 58, 57, rng)
    boxes = [(a + 2, b + 2, c + 2, d + 2) for a, b, c, d in boxes]
    T = dom.tair.copy()
    for t, (iw, ie, js, jn) in enumerate(boxes):
        frame = np.zeros((ny, nx), dtype=bool)
        frame[js - 1: jn + 2, iw - 1: ie + 2] = True
        frame[js: jn + 1, iw: ie + 1] = False
        # outside THIS step's box and outside its time neighbours' (their dT/dt reads this step's T inside THEIR boxes)
        for tn in (t - 1, t + 1):
            if 0 <= tn < nt:
                a, b, c, d = boxes[tn]
                frame[c: d + 1, a: b + 1] = False
        T[t][:, frame] = np.nan
    eng = LECEngine(dom.lat, dom.lon, dom.level, device=DEV)
    clean = [torch.as_tensor(np.ascontiguousarray(x)).to(DEV) for x in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    dirty = [torch.as_tensor(np.ascontiguousarray(T)).to(DEV)] + clean[1:]
    ref = _rows(eng, clean, boxes, dom.time_s, 1)
    assert bool(torch.isfinite(ref.rows[..., :28]).all())
    for tg in (1, 2, 4):
        got = _rows(eng, dirty, boxes, dom.time_s, tg)
        assert _same(got.rows, ref.rows), tg


def test_config5_shape_in_time_groups():
    """BASELINE config 5's shape (37 levels, 61 x 61 boxes on a 0.25-degree crop, the bench's track) on 37 steps: TG = 2 and 4 against
    TG = 1, the terms against the oracle."""
    from lorenzcycletoolkit_amd.synthetic import era5_like_levels, synthetic_cube
    from oracle import lec_oracle as o
    from tests.helpers import scale_err
    level = era5_like_levels()
    lat = np.arange(-57.75, -17.5 + 1e-9, 0.25)
    lon = np.arange(-80.25, -19.75 + 1e-9, 0.25)
    nt = 37
    f = synthetic_cube(nt, level, lat, lon, device=DEV, dtype=torch.float64, seed=77)
    eng = LECEngine(lat, lon, level, device=DEV)
    tg_ = np.arange(nt) * 9                                                        # nine bench steps apart: boxes up to 7 columns apart
    clat = -37.5 + 12.0 * np.sin(2 * np.pi * tg_ / 400.0)
    clon = -50.0 + 22.0 * np.cos(2 * np.pi * tg_ / 700.0)
    slow = [eng.box_from_limits(-50.0 + 0.2 * t - 7.5, -50.0 + 0.2 * t + 7.5, -37.5 - 7.5, -37.5 + 7.5) for t in range(nt)]
    fast = [eng.box_from_limits(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(clat, clon)]
    time_s = np.arange(nt) * 3600.0
    cubes = [f["tair"], f["u"], f["v"], f["omega"], f["geopt"]]
    for boxes in (slow, fast):
        one = _rows(eng, cubes, boxes, time_s, 1)
        for tg in (2, 4):
            got = _rows(eng, cubes, boxes, time_s, tg)
            assert _same(got.rows, one.rows) and torch.equal(got.scalars, one.scalars), tg
    host = {k: v[:6].cpu().numpy() for k, v in f.items()}
    dom = o.Domain(host["tair"], host["u"], host["v"], host["omega"], host["geopt"], lat, lon, level, time_s[:6])
    limits = [(lon[b[0]], lon[b[1]], lat[b[2]], lat[b[3]]) for b in slow[:6]]
    sc, _ = o.lec_moving(dom, limits)
    got = eng.compute(*[c[:6] for c in cubes], slow[:6], time_s=time_s[:6], per_step_boxes=True, tuning={"kernel": "box_tile", "block_shape": 4})
    names = ("Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "BΦZ", "BΦE", "Gz", "Ge")
    gs = got.scalars_dict()
    for n in names:
        assert scale_err(gs[n], np.asarray(sc[n])) < 1e-9, n


def test_bad_time_group_is_refused():
    dom = synthetic_domain(4, 3, 20, 30, seed=1)
    eng = LECEngine(dom.lat, dom.lon, dom.level, device=DEV)
    f = [torch.as_tensor(np.ascontiguousarray(x)).to(DEV) for x in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    with pytest.raises(ValueError, match="block_shape"):
        eng.compute(*f, [(2, 20, 2, 15)] * 4, time_s=dom.time_s, per_step_boxes=True, tuning={"kernel": "box_tile", "block_shape": 3})
