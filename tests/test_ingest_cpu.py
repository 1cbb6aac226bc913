"""The host side of the device ingest, without a GPU: the index maps of ingest.make_plan applied with NumPy to the raw
(memory-mapped, big-endian, possibly packed) file variables must reproduce what the host preparation path
(open_dataset + process_data + slice_domain) produces, element for element."""
import argparse
import os
import shutil

import numpy as np
import pytest

from lorenzcycletoolkit_amd import dataset as ds
from lorenzcycletoolkit_amd import ingest


@pytest.fixture
def workdir(tmp_path, golden_dir, monkeypatch):
    os.makedirs(tmp_path / "inputs")
    shutil.copy(os.path.join(golden_dir, "inputs", "namelist_NCEP-R2"), tmp_path / "inputs" / "namelist")
    monkeypatch.chdir(tmp_path)
    return tmp_path


def _emulate_lec_ingest(var: ds.RawVariable, plan: ingest.IngestPlan) -> np.ndarray:
    """What lec_ingest has to produce (include/lec_hip.h): the plan's gather of the file values, decoded as the reference's
    xarray does -- by the ORACLE's restatement of that decode (oracle/cf_decode.py), not by the package's own decoder."""
    from oracle import cf_decode as cf
    raw = np.asarray(var.data)[plan.tsel][:, plan.kmap][:, :, plan.jmap][:, :, :, plan.imap]
    attrs = {k: v for k, v in (("scale_factor", var.scale_factor), ("add_offset", var.add_offset), ("_FillValue", var.fill_value))
             if v is not None}
    return cf.decode_cf_variable(raw, attrs)


def _compare(infile, args):
    df = ds.read_namelist("inputs/namelist")
    host = ds.slice_domain(ds.process_data(ds.open_dataset(infile, df), args, df), args, df)
    raw = ds.open_raw(infile, df)
    plan = ingest.make_plan(raw, args)
    assert np.array_equal(plan.lat, host.lat) and np.array_equal(plan.lon, host.lon)
    assert np.array_equal(plan.level, host.level) and np.array_equal(plan.time, host.time)
    assert np.all(np.diff(plan.lat) > 0) and np.all(np.diff(plan.lon) > 0) and np.all(np.diff(plan.level) > 0)
    for name, var in raw.variables.items():
        got = _emulate_lec_ingest(var, plan)
        want = host.variables[name]
        assert got.shape == want.shape and got.dtype == want.dtype, name
        assert np.array_equal(got, want, equal_nan=True), name
    raw.close()
    return plan, host


def test_catarina_fixed_plan(workdir, golden_dir):
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    args = argparse.Namespace(fixed=True, track=False, trackfile=None)
    plan, host = _compare(os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), args)
    assert plan.tsel.tolist() == list(range(36))


def test_testdata_track_plan(workdir, golden_dir):
    """Track mode: time steps selected from the track, domain = track extent +- (half box + one grid step)."""
    shutil.copy(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), workdir / "inputs" / "track")
    args = argparse.Namespace(fixed=False, track=True, trackfile="inputs/track")
    plan, host = _compare(os.path.join(golden_dir, "testdata_NCEP-R2.nc"), args)
    assert len(plan.tsel) == len(host.time)


def test_packed_file_plan(workdir):
    from tests.helpers import write_packed_era5_style
    path = str(workdir / "packed.nc")
    write_packed_era5_style(path, nt=5)
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
        "Time;time\nVertical Level;level\n")
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;20\nmin_lat;-45\nmax_lat;30\n")
    args = argparse.Namespace(fixed=True, track=False, trackfile=None)
    plan, host = _compare(path, args)
    assert plan.level.size == 8 and plan.level[0] == 5000.0          # 5 hPa dropped, 50 hPa first
    assert plan.lon[0] == -60.0 and plan.lon[-1] == 20.0 and plan.lat[0] == -45.0 and plan.lat[-1] == 30.0
    assert np.isnan(host.variables["v"]).any()                         # the fill values became NaN on both paths


def test_open_raw_rejects_what_the_kernel_cannot_read(workdir, golden_dir):
    from scipy.io import netcdf_file
    path = str(workdir / "bad.nc")
    f = netcdf_file(path, "w")
    for n, s in (("time", 2), ("level", 2), ("lat", 3), ("lon", 3)):
        f.createDimension(n, s)
        v = f.createVariable(n, "f", (n,)); v[:] = np.arange(s)
    f.variables["time"].units = "hours since 2000-01-01"
    for n in ("t", "u", "v", "w", "z"):
        v = f.createVariable(n, "f", ("time", "lat", "level", "lon")); v[:] = 0        # wrong axis order
    f.close()
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;lon\nLatitude;lat\nTime;time\nVertical Level;level\n")
    df = ds.read_namelist("inputs/namelist")
    with pytest.raises(ValueError, match="order"):
        ds.open_raw(path, df)


def test_prepare_data_is_the_lazy_gather_and_equals_the_whole_file_path(workdir, golden_dir):
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    args = argparse.Namespace(fixed=True, track=False, trackfile=None, cdsapi=False,
                              infile=os.path.join(golden_dir, "Catarina_NCEP-R2.nc"))
    df = ds.read_namelist("inputs/namelist")
    whole = ds.slice_domain(ds.process_data(ds.open_dataset(args.infile, df), args, df), args, df)
    lazy = ds.prepare_data(args, "inputs/namelist")
    assert np.array_equal(lazy.lat, whole.lat) and np.array_equal(lazy.lon, whole.lon) and np.array_equal(lazy.time, whole.time)
    for k in whole.variables:
        assert lazy.variables[k].dtype == whole.variables[k].dtype and np.array_equal(lazy.variables[k], whole.variables[k]), k


def test_prepare_data_falls_back_for_other_axis_orders(workdir):
    """A file whose variables are (time, lat, level, lon) cannot be streamed; the whole-file path transposes it."""
    from scipy.io import netcdf_file
    rng = np.random.default_rng(2)
    path = str(workdir / "odd.nc")
    f = netcdf_file(path, "w")
    sizes = {"time": 3, "level": 4, "lat": 6, "lon": 8}
    coords = {"time": np.arange(3) * 6.0, "level": np.array([1000.0, 850.0, 500.0, 200.0]), "lat": np.linspace(-50, -25, 6), "lon": np.linspace(-60, -25, 8)}
    for n, sz in sizes.items():
        f.createDimension(n, sz)
        v = f.createVariable(n, "d", (n,)); v[:] = coords[n]
    f.variables["time"].units = "hours since 2000-01-01"
    f.variables["level"].units = "hPa"
    data = {}
    for n in ("t", "u", "v", "w", "z"):
        data[n] = rng.standard_normal((3, 6, 4, 8))
        v = f.createVariable(n, "d", ("time", "lat", "level", "lon")); v[:] = data[n]
    f.close()
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;lon\nLatitude;lat\nTime;time\nVertical Level;level\n")
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-30\nmin_lat;-45\nmax_lat;-30\n")
    args = argparse.Namespace(fixed=True, track=False, trackfile=None, cdsapi=False, infile=path)
    got = ds.prepare_data(args, "inputs/namelist")
    assert got.variables["t"].shape == (3, 4, 4, 6) and got.level.tolist() == [20000.0, 50000.0, 85000.0, 100000.0]
    want = np.transpose(data["t"], (0, 2, 1, 3))[:, ::-1][:, :, 1:5][:, :, :, 1:7]
    assert np.array_equal(got.variables["t"], want)


def test_registered_spans_bookkeeping():
    """RegisteredSpans keeps the host memory registered for direct copies as page-aligned, NON-overlapping blocks (the runtime refuses
    overlaps), registers only what a request does not find covered, cuts copies at block boundaries and releases by last use."""
    from lorenzcycletoolkit_amd import ingest

    class FakeRuntime:
        def __init__(self):
            self.reg = {}

        def lec_host_register(self, p, n):
            a = p.value
            assert a % 4096 == 0 and n % 4096 == 0
            assert all(a >= e or a + n <= b for b, e in self.reg.items()), "overlapping registration"
            self.reg[a] = a + n
            return 0

        def lec_host_unregister(self, p):
            del self.reg[p.value]
            return 0

    rt = FakeRuntime()
    sp = ingest.RegisteredSpans(rt)
    base = 1 << 30
    sp.ensure(base + 10000, base + 30000, 0)
    assert sp.blocks == [[base + 8192, base + 32768, 0]]
    sp.ensure(base + 25000, base + 50000, 1)                               # overlaps the first: only the uncovered tail is registered
    assert sp.blocks == [[base + 8192, base + 32768, 1], [base + 32768, base + 53248, 1]] and sp.calls == 2
    assert sp.pieces(base + 20000, base + 45000) == [(base + 20000, base + 32768), (base + 32768, base + 45000)]
    sp.ensure(base + 100, base + 5000, 2)
    sp.ensure(base + 4000, base + 60000, 2)                                # fills the gap between blocks and extends the end: two new blocks
    assert [b[:2] for b in sp.blocks] == [[base, base + 8192], [base + 8192, base + 32768], [base + 32768, base + 53248],
                                          [base + 53248, base + 61440]] and all(b[2] == 2 for b in sp.blocks)
    assert sp.registered_bytes == 61440
    with pytest.raises(RuntimeError):
        sp.pieces(base + 60000, base + 70000)                              # runs past what is registered
    sp.ensure(base + 200000, base + 204096, 5)
    sp.release(3)                                                          # everything last used by chunks < 3 goes
    assert [b[:2] for b in sp.blocks] == [[base + 200704 - 4096, base + 208896]] or len(sp.blocks) == 1
    sp.close()
    assert rt.reg == {} and sp.blocks == []


def test_a_registration_refused_in_mid_call_leaves_no_untracked_block():
    """ensure() over a range with a registered block in its middle registers TWO new blocks; the runtime refuses the second.  The first
    stays tracked: a later request over its pages finds it covered (no overlapping registration, which the runtime would refuse
    too -- one refusal would cascade every variable into the pinned fallback), and close() releases it."""
    from lorenzcycletoolkit_amd import _lib, ingest

    class Runtime:
        def __init__(self):
            self.reg, self.calls, self.fail_at = {}, 0, None

        def lec_host_register(self, p, n):
            self.calls += 1
            if self.calls == self.fail_at:
                return 2
            a = p.value
            assert all(a >= e or a + n <= b for b, e in self.reg.items()), "overlapping registration"
            self.reg[a] = a + n
            return 0

        def lec_host_unregister(self, p):
            del self.reg[p.value]
            return 0

    rt = Runtime()
    sp = ingest.RegisteredSpans(rt)
    base = 1 << 30
    sp.ensure(base + 8192, base + 12288, 0)
    rt.fail_at = 3                                        # the call below registers [base, base+8192) (2nd call) and [base+12288, base+20480) (3rd)
    with pytest.raises(_lib.LecLibraryError):
        sp.ensure(base, base + 20480, 1)
    assert [b[:2] for b in sp.blocks] == [[base, base + 8192], [base + 8192, base + 12288]] and set(rt.reg) == {base, base + 8192}
    rt.fail_at = None
    sp.ensure(base + 100, base + 9000, 2)                 # covered: nothing new is registered (an untracked block would be registered AGAIN here)
    assert rt.calls == 3
    sp.ensure(base, base + 20480, 2)                      # the refused tail alone
    assert rt.calls == 4 and sp.registered_bytes == 20480
    sp.close()
    assert rt.reg == {} and sp.blocks == []


def test_chunk_copy_plan_places_every_stream_and_keeps_its_alignment():
    """ingest.chunk_copy_plan on random layouts: adjacent chunks, small and large gaps, repeated chunks, any order, with and without
    the 4 checksum bytes: copying the runs reproduces every stream at its planned offset, offsets keep the file alignment modulo 16,
    runs do not overlap in the buffer, and chunks further apart than the gap get separate runs."""
    from lorenzcycletoolkit_amd.ingest import chunk_copy_plan
    rng = np.random.default_rng(0)
    multi = 0
    for trial in range(300):
        n, tail = int(rng.integers(1, 40)), int(rng.choice([0, 4]))
        size = rng.integers(1, 5000, n).astype(np.int64)
        addr = np.zeros(n, dtype=np.int64)
        cur = int(rng.integers(0, 1000))
        for i in range(n):
            if i and rng.random() < 0.15:                     # the same chunk again (a repeated time step)
                j = int(rng.integers(0, i))
                addr[i], size[i] = addr[j], size[j]
                continue
            cur += int(rng.choice([0, 0, 0, 3, 70000, 200000]))
            addr[i] = cur
            cur += int(size[i]) + tail
        perm = rng.permutation(n)
        addr, size = addr[perm], size[perm]
        blob = rng.integers(0, 256, int((addr + size + tail).max()) + 10, dtype=np.uint8)
        src_off, run_lo, run_len, base, need = chunk_copy_plan(addr, size, tail)
        multi += len(run_lo) > 1
        comp = np.zeros(need, dtype=np.uint8)
        spans = sorted(zip(base.tolist(), run_len.tolist()))
        assert all(b0 + l0 <= b1 for (b0, l0), (b1, _l1) in zip(spans, spans[1:])) and spans[-1][0] + spans[-1][1] <= need
        for lo, ln, d in zip(run_lo.tolist(), run_len.tolist(), base.tolist()):
            comp[d: d + ln] = blob[lo: lo + ln]
        for c in range(n):
            assert np.array_equal(comp[src_off[c]: src_off[c] + size[c] + tail], blob[addr[c]: addr[c] + size[c] + tail]), (trial, c)
            assert (src_off[c] - addr[c]) % 16 == 0
        a_sorted = np.sort(addr)
        far = int(np.sum(np.diff(a_sorted) > 5000 + tail + (64 << 10)))
        assert len(run_lo) >= far + 1
    assert multi > 50
