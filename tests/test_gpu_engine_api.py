"""The round-3 additions to the engine's surface, each against the path it short-cuts (GPU): packed output records through
lec_reduce's strides, device-resident d/dt coefficients, views of prepared box tables, a streamed ingest of a time range, and the
device-side index checks of the C ABI called through ctypes."""
import argparse
import ctypes as C
import os
import shutil

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from lorenzcycletoolkit_amd import _lib
from lorenzcycletoolkit_amd.engine import LECEngine
from tests.helpers import synthetic_domain

DEV = "cuda:0"


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def _setup(nt=6, nl=5, ny=14, nx=70, seed=5, uneven=True):
    dom = synthetic_domain(nt, nl, ny, nx, seed=seed)
    if uneven:
        dom.time_s = np.cumsum(np.array([0, 3600, 3600, 7200, 3600, 10800, 3600, 3600][:nt], dtype=np.float64))
    eng = LECEngine(dom.lat, dom.lon, dom.level, device=DEV)
    f = [_dev(a) for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt)]
    return dom, eng, f


def test_reduce_writes_packed_records_where_it_is_told():
    """``out=``: a caller-owned [t, 16 + 21 nl] buffer with a LONGER row stride (the gather's send buffer carries one more column):
    the same bits as the default, scalars / levels are views of it, nothing else of the buffer is touched."""
    dom, eng, f = _setup()
    box = eng.box_from_limits(dom.lon[2], dom.lon[-3], dom.lat[1], dom.lat[-2])
    ref = eng.compute(*f, [box], time_s=dom.time_s)
    nl = len(dom.level)
    w = LECEngine.packed_width(nl)
    assert w == 16 + 21 * nl and ref.packed.shape == (6, w)
    assert ref.scalars.data_ptr() == ref.packed.data_ptr() and ref.levels.data_ptr() == ref.packed.data_ptr() + 16 * 8
    buf = torch.full((6, w + 3), -7.0, dtype=torch.float64, device=DEV)
    nan = torch.empty(6, dtype=torch.int32, device=DEV)
    res = eng.compute(*f, [box], time_s=dom.time_s, out=buf[:, :w])
    rows = eng.rowstats(*f, [box], time_s=dom.time_s)
    res2 = eng.reduce(rows, [box], out=buf[:, :w], nanflag_out=nan)
    torch.cuda.synchronize()
    assert torch.equal(buf[:, :w], ref.packed) and torch.all(buf[:, w:] == -7.0)
    assert torch.equal(res.scalars, ref.scalars) and torch.equal(res2.levels, ref.levels) and res2.nanflag.data_ptr() == nan.data_ptr()
    for bad in (buf[:, :w - 1], buf[:5, :w], buf[:, :w].float(), buf[:, :w].cpu(),
                torch.empty((w, 6), dtype=torch.float64, device=DEV).t()):          # short | fewer steps | fp32 | host | columns not contiguous
        with pytest.raises(ValueError):
            eng.reduce(rows, [box], out=bad)
    with pytest.raises(ValueError):
        eng.reduce(rows, [box], nanflag_out=nan[:5])
    # the C ABI refuses strides shorter than a record
    lib = _lib.load()
    rd = _lib.ReduceArgs(rows_d=16, t_count=1, nl=nl, n_box=1, nyb_max=4, box_d=16, boxtab2_d=16, lattab2_d=16, levtab2_d=16, am_d=16, levraw_d=16,
                         scalars_d=16, levels_d=16, nanflag_d=16, scalars_stride=8, levels_stride=0)       # never dereferenced: validation comes first
    assert lib.lec_reduce(C.byref(rd)) == 1 and b"stride" in lib.lec_last_error()


def test_device_resident_time_coefficients_equal_the_per_call_upload():
    """A chunk loop passes rows [h0, h1) of ``time_coefs_device(whole axis)`` instead of the chunk's ``time_s`` (no upload per call):
    for a cube that holds the steps [h0, h1) and processes only steps whose neighbours it holds, the records are the same bits --
    on an UNEVEN axis, where a wrong neighbour spacing would show."""
    dom, eng, f = _setup(nt=8)
    box = eng.box_from_limits(dom.lon[2], dom.lon[-3], dom.lat[1], dom.lat[-2])
    whole = eng.rowstats(*f, [box], time_s=dom.time_s)
    tc = eng.time_coefs_device(dom.time_s)
    for (h0, h1, a, b) in ((0, 4, 0, 3), (2, 7, 3, 6), (5, 8, 6, 8), (0, 8, 0, 8)):        # held [h0, h1), processed [a, b)
        g = [x[h0:h1].contiguous() for x in f]
        part = eng.rowstats(*g, [box], tcoef=tc[h0:h1], t_begin=a - h0, t_count=b - a)
        also = eng.rowstats(*g, [box], time_s=dom.time_s[h0:h1], t_begin=a - h0, t_count=b - a)
        # slots 28..31 of a record are stage-1 scratch (the first processed step of a launch keeps its backward covariance pieces there)
        assert torch.equal(part[..., :28], whole[a:b][..., :28]) and torch.equal(also, part), (h0, h1, a, b)
    with pytest.raises(ValueError):
        eng.rowstats(*f, [box], tcoef=tc[:5])
    with pytest.raises(ValueError):
        eng.rowstats(*f, [box], tcoef=tc.float())


def test_parts_of_prepared_boxes_are_views_and_give_the_same_records():
    dom, eng, f = _setup(nt=6, ny=30, nx=90, uneven=False)
    boxes = [(3 + t, 40 + 2 * t, 2 + t, 15 + t) for t in range(6)]
    prep = eng.prepare_boxes(boxes)
    whole = eng.compute(*f, prep, time_s=dom.time_s, keep_rows=True)
    fresh = eng.compute(*f, boxes, time_s=dom.time_s, keep_rows=True)
    assert torch.equal(whole.rows, fresh.rows) and torch.equal(whole.scalars, fresh.scalars)
    part = prep.part(2, 5)
    assert len(part) == 3 and part.boxes == boxes[2:5] and part.dev["box"].data_ptr() == prep.dev["box"][2:].data_ptr()      # a view: no upload
    got = eng.compute(*f, part, time_s=dom.time_s, t_begin=2, t_count=3, keep_rows=True, per_step_boxes=True)
    assert torch.equal(got.rows, whole.rows[2:5]) and torch.equal(got.scalars, whole.scalars[2:5])
    with pytest.raises(ValueError):
        prep.part(4, 4)


@pytest.fixture
def workdir(tmp_path, golden_dir, monkeypatch):
    os.makedirs(tmp_path / "inputs")
    shutil.copy(os.path.join(golden_dir, "inputs", "namelist_NCEP-R2"), tmp_path / "inputs" / "namelist")
    (tmp_path / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    monkeypatch.chdir(tmp_path)
    return tmp_path


def test_streamed_ingest_of_a_time_range_equals_that_part_of_the_whole(workdir, golden_dir):
    """``lec_streamed(t_range=...)`` is what a rank of a time-sharded run calls: only its steps (+ the T halo) are staged and copied, and
    the records are those of the whole run (the mask merge aside: this sample has no NaN)."""
    from lorenzcycletoolkit_amd import dataset as ds
    from lorenzcycletoolkit_amd import ingest
    args = argparse.Namespace(fixed=True, track=False, trackfile=None, residuals=True)
    df = ds.read_namelist("inputs/namelist")
    raw = ds.open_raw(os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), df)
    plan = ingest.make_plan(raw, args)
    limits = [(-55.0, -36.0, -35.0, -20.0)]
    st_whole, st_part = {}, {}
    whole = ingest.lec_streamed(raw, plan, df, limits, chunk_steps=5, stats=st_whole)
    w = LECEngine.packed_width(len(plan.level))
    for (a, b) in ((0, 12), (12, 13), (13, 36)):
        out = torch.zeros((b - a, w + 1), dtype=torch.float64, device=DEV)
        part = ingest.lec_streamed(raw, plan, df, limits, chunk_steps=5, t_range=(a, b), out=out[:, :w], stats=st_part)
        torch.cuda.synchronize()
        assert torch.equal(part.scalars, whole.scalars[a:b]) and torch.equal(part.levels, whole.levels[a:b]), (a, b)
        assert torch.equal(out[:, :w], whole.packed[a:b]) and part.scalars.data_ptr() == out.data_ptr()
        assert st_part["bytes_moved"] < st_whole["bytes_moved"] * ((b - a + 2) / 36 + 0.05)       # own steps + halo only
    with pytest.raises(ValueError):
        ingest.lec_streamed(raw, plan, df, limits, t_range=(5, 40))
    raw.close()


def test_device_side_index_checks_through_ctypes():
    """lec_check_boxes / lec_check_maps from Python, as a C caller would use them (tests/c_abi/lec_c_client.c does the same in C)."""
    lib = _lib.load()
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    good = torch.tensor([[0, 9, 0, 4], [3, 19, 2, 7]], dtype=torch.int32, device=DEV)
    ra = _lib.RowstatsArgs(box_d=good.data_ptr(), n_box=2, nx=20, ny=8, nxb_max=17, nyb_max=6, stream=stream)
    assert lib.lec_check_boxes(C.byref(ra), status.data_ptr()) == 0
    for bad, what in (([[0, 9, 0, 4], [3, 20, 2, 7]], b"first: box 1"), ([[-1, 9, 0, 4], [3, 19, 2, 7]], b"first: box 0"),
                      ([[0, 9, 4, 4], [3, 19, 2, 7]], b"1 of 2"), ([[0, 17, 0, 4], [3, 19, 2, 7]], b"first: box 0")):      # outside | negative | one row | too wide
        t = torch.tensor(bad, dtype=torch.int32, device=DEV)
        ra.box_d = t.data_ptr()
        assert lib.lec_check_boxes(C.byref(ra), status.data_ptr()) == 1 and what in lib.lec_last_error(), bad
    big = torch.zeros((100000, 4), dtype=torch.int32, device=DEV)
    big[:, 1] = 9; big[:, 3] = 4
    big[77777, 1] = 20
    ra.box_d, ra.n_box = big.data_ptr(), 100000
    assert lib.lec_check_boxes(C.byref(ra), status.data_ptr()) == 1 and b"first: box 77777" in lib.lec_last_error()
    maps = torch.tensor([0, 1, 2, 5, 4, 0, 1], dtype=torch.int32, device=DEV)
    ga = _lib.IngestArgs(nl_in=3, ny_in=6, nx_in=2, nl=3, ny=2, nx=2, kmap_d=maps.data_ptr(), jmap_d=maps[3:].data_ptr(), imap_d=maps[5:].data_ptr(),
                         stream=stream)
    assert lib.lec_check_maps(C.byref(ga), status.data_ptr()) == 0
    ga.ny_in = 5
    assert lib.lec_check_maps(C.byref(ga), status.data_ptr()) == 1 and b"jmap_d[0]" in lib.lec_last_error()


def test_a_per_step_gather_table_is_bounds_checked_on_the_device():
    """ABI 10: lec_ingest_args.step_d lives in device memory, where argument validation cannot see it (the reference bounds-checks its
    track against the data on the host: lec_moving_framework.py:112-154).  lec_check_maps scans it and names the first bad entry;
    lec_ingest itself never reads through a bad entry -- that output step's rows come out NaN, the others are gathered as always."""
    lib = _lib.load()
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nt_src, nl, ny_in, nx_in, ny, nx = 3, 2, 6, 7, 2, 3
    src = torch.arange(nt_src * nl * ny_in * nx_in, dtype=torch.float32, device=DEV).reshape(nt_src, nl, ny_in, nx_in)
    up = lambda a: torch.tensor(a, dtype=torch.int32, device=DEV)
    kmap, jmap, imap = up([0, 1]), up([0, 1, 2, 3, 4, 5]), up([0, 1, 2, 3, 4, 5, 6])
    good = [[10, 0, 0], [11, 4, 4], [12, 2, 1], [10, 3, 2]]              # {source step (the source starts with step 10), latitude offset, longitude offset}
    out = torch.zeros((4, nl, ny, nx), dtype=torch.float32, device=DEV)

    def args(table):
        t = up(table)
        return t, _lib.IngestArgs(src_d=src.data_ptr(), src_dtype=_lib.LEC_F32, nt=len(table), nl_in=nl, ny_in=ny_in, nx_in=nx_in, nl=nl, ny=ny, nx=nx,
                                  kmap_d=kmap.data_ptr(), jmap_d=jmap.data_ptr(), imap_d=imap.data_ptr(), unit_scale=1.0, out_dtype=_lib.LEC_F32,
                                  decode_dtype=_lib.LEC_F32, out_d=out.data_ptr(), stream=stream, step_d=t.data_ptr(), step_base=10, nt_src=nt_src,
                                  jmap_len=6, imap_len=7)

    t, ga = args(good)
    assert lib.lec_check_maps(C.byref(ga), status.data_ptr()) == 0, lib.lec_last_error()
    assert lib.lec_ingest(C.byref(ga)) == 0
    torch.cuda.synchronize()
    for i, (ts, oj, oi) in enumerate(good):
        assert torch.equal(out[i], src[ts - 10, :, oj:oj + ny, oi:oi + nx])
    for bad, what in (([[10, 0, 0], [13, 0, 0]], b"step_d[1]"),           # a source step past the source
                      ([[9, 0, 0], [10, 0, 0]], b"step_d[0]"),            # ... before it
                      ([[10, 0, 0], [10, 1, 0], [10, 5, 0]], b"step_d[2]"),      # latitude offset + ny past the map
                      ([[10, 0, 5]], b"step_d[0]"), ([[10, -1, 0]], b"step_d[0]")):
        t, ga = args(bad)
        assert lib.lec_check_maps(C.byref(ga), status.data_ptr()) == 1 and what in lib.lec_last_error() and b"maps themselves are fine" in lib.lec_last_error(), bad
        out.fill_(7.0)
        assert lib.lec_ingest(C.byref(ga)) == 0
        torch.cuda.synchronize()
        nbad = int(what[7:-1])
        for i, (ts, oj, oi) in enumerate(bad):
            if i == nbad:
                assert bool(torch.isnan(out[i]).all())
            else:
                assert torch.equal(out[i], src[ts - 10, :, oj:oj + ny, oi:oi + nx])
    # a bad MAP entry is still named before the table
    jbad = up([0, 1, 2, 3, 4, 6])
    t, ga = args(good)
    ga.jmap_d = jbad.data_ptr()
    assert lib.lec_check_maps(C.byref(ga), status.data_ptr()) == 1 and b"jmap_d[5]" in lib.lec_last_error()
