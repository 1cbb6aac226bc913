"""Device ingest: memory-mapped file bytes -> lec_ingest -> lec_rowstats per chunk -> one lec_reduce must
give the very same numbers as the host-prepared, fully resident path (open_dataset + process_data + slice_domain +
BoxData), for the reference's own float32 sample and for an ERA5-style int16-packed file whose axes need the
longitude wrap, every sort, the 10 hPa filter and a fill value."""
import argparse
import os
import shutil

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from lorenzcycletoolkit_amd import dataset as ds
from lorenzcycletoolkit_amd import ingest
from lorenzcycletoolkit_amd.frameworks import BoxData
from tests.helpers import write_packed_era5_style as _write_packed


@pytest.fixture
def workdir(tmp_path, golden_dir, monkeypatch):
    os.makedirs(tmp_path / "inputs")
    shutil.copy(os.path.join(golden_dir, "inputs", "namelist_NCEP-R2"), tmp_path / "inputs" / "namelist")
    monkeypatch.chdir(tmp_path)
    return tmp_path


def _both_paths(infile, namelist, limits, chunk_steps):
    args = argparse.Namespace(fixed=True, track=False, trackfile=None, residuals=True)
    df = ds.read_namelist(namelist)
    host = ds.slice_domain(ds.process_data(ds.open_dataset(infile, df), args, df), args, df)
    box = BoxData(host, df, *limits, args=args)
    raw = ds.open_raw(infile, df)
    plan = ingest.make_plan(raw, args)
    stats = {}
    res = ingest.lec_fixed_streamed(raw, plan, df, limits, chunk_steps=chunk_steps, stats=stats)
    torch.cuda.synchronize()
    raw.close()
    assert np.array_equal(plan.lat, host.lat) and np.array_equal(plan.lon, host.lon) and np.array_equal(plan.level, host.level)
    assert np.array_equal(plan.time, host.time)
    return box.result, res, stats


@pytest.mark.parametrize("chunk_steps", [5, 36, 1])
def test_catarina_streamed_equals_resident(workdir, golden_dir, chunk_steps):
    limits = (-55.0, -36.0, -35.0, -20.0)
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    a, b, stats = _both_paths(os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), "inputs/namelist", limits, chunk_steps)
    # a one-step first chunk fills the copy pipeline, then chunks of chunk_steps
    assert stats["storage"] == "float32" and stats["chunks"] == (1 if chunk_steps >= 36 else 1 + -(-35 // chunk_steps))
    assert torch.equal(a.scalars, b.scalars)
    assert torch.equal(a.levels, b.levels)
    assert torch.equal(a.nanflag, b.nanflag)


@pytest.mark.parametrize("chunk_steps", [3, 7])
def test_packed_int16_file_streamed_equals_resident(workdir, chunk_steps):
    path = str(workdir / "packed.nc")
    _write_packed(path)
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
        "Time;time\nVertical Level;level\n")
    limits = (-60.0, 20.0, -45.0, 30.0)      # crosses the Greenwich meridian: needs the longitude wrap + sort
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;20\nmin_lat;-45\nmax_lat;30\n")
    a, b, stats = _both_paths(path, "inputs/namelist", limits, chunk_steps)
    assert stats["storage"] == "float32" and stats["domain"][1] == 8          # int16 + fill value: float32 in the reference's xarray; 5 hPa dropped
    assert int(a.nanflag.max()) > 0                                            # the fill values reached _handle_nans
    assert torch.equal(a.scalars, b.scalars)
    assert np.array_equal(a.levels.cpu().numpy(), b.levels.cpu().numpy(), equal_nan=True)     # NaN levels stay NaN in the tables
    assert torch.isfinite(a.scalars[:, :4]).all()
    full_bytes = 5 * 7 * 9 * 49 * 144 * 2
    band = (30 + 45) / 2.5 + 1                                                 # latitudes of the box: only that band of the 8 kept levels is staged
    assert stats["bytes_moved"] <= 1.6 * full_bytes * (8 / 9) * (band / 49)    # int16 over PCIe (+ T halo), not fp64, not the whole globe


@pytest.mark.parametrize("name,storage", [("packed_chunked_tracked.nc", "float32"), ("float_chunked_latest.nc", "float32"),
                                          ("packed_chunked_earliest.nc", "float32")])
def test_netcdf4_file_streamed_equals_resident(workdir, name, storage):
    """NetCDF-4 / HDF5 input (hdf5_lite): deflated chunks are inflated on the device (lec_inflate; the contiguous-free float file too),
    then take the same device path -- the resident run inflates on the host."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", "hdf5", name)
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
        "Time;time\nVertical Level;level\n")
    limits = (-60.0, 30.0, -40.0, 30.0)
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;30\nmin_lat;-40\nmax_lat;30\n")
    a, b, stats = _both_paths(path, "inputs/namelist", limits, 2)
    assert stats["storage"] == storage and stats["chunks"] == 3 and stats["inflate"] == "device"
    assert torch.equal(a.scalars, b.scalars)
    assert np.array_equal(a.levels.cpu().numpy(), b.levels.cpu().numpy(), equal_nan=True)
    assert torch.isfinite(a.scalars[:, :4]).all()


HDF5_DEFLATED = ["packed_chunked_earliest.nc", "packed_chunked_tracked.nc", "float_chunked_latest.nc", "packed_unlimited_v18.nc",
                 "packed_unlimited_latest.nc", "packed_timechunk2_latest.nc",
                 "float_chunked_plain_latest.nc", "packed_shuffle_only_v18.nc",       # chunked without deflate: chunks copied, not inflated
                 "packed_interleaved_v18.nc"]                                         # written step by step: the variables' chunks interleave


@pytest.mark.parametrize("name", HDF5_DEFLATED)
def test_device_inflate_equals_host_inflate(workdir, name):
    """Every deflated fixture (v1 B-tree / fixed-array / extensible-array chunk indexes, shuffle, fletcher32 trailers, int16 and
    float32, chunks of 2 levels x half the longitudes): the raw cube rebuilt by lec_inflate + lec_chunk_scatter and decoded by
    lec_ingest equals the one the host reader inflates, for the whole domain and for a band of it (whole chunks cross the link
    compressed; the band is cut out on the device)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", "hdf5", name)
    (workdir / "inputs" / "namelist").write_text(ERA5_NAMELIST)
    args = argparse.Namespace(fixed=True, track=False, trackfile=None)
    df = ds.read_namelist("inputs/namelist")
    for limits in ((-60.0, 30.0, -40.0, 30.0), (-170.0, 170.0, -60.0, 60.0)):
        (workdir / "inputs" / "box_limits").write_text("min_lon;%r\nmax_lon;%r\nmin_lat;%r\nmax_lat;%r\n" % limits)
        raw = ds.open_raw(path, df)
        plan = ingest.make_plan(raw, args)
        for var in ("t", "u", "v", "w", "z"):
            v = raw.variables[var]
            assert v.data.chunk_streams() is not None
            dev = ingest.device_cube(v, plan, inflate="device")
            host = ingest.device_cube(v, plan, inflate="host")
            assert dev.dtype == host.dtype and torch.equal(torch.nan_to_num(dev, nan=-1e30), torch.nan_to_num(host, nan=-1e30)), (var, limits)
        st_d, st_h = {}, {}
        a = ingest.lec_streamed(raw, plan, df, [limits], chunk_steps=2, stats=st_d, inflate="device")
        b = ingest.lec_streamed(raw, plan, df, [limits], chunk_steps=2, stats=st_h, inflate="host")
        assert torch.equal(a.scalars, b.scalars) and np.array_equal(a.levels.cpu().numpy(), b.levels.cpu().numpy(), equal_nan=True)
        assert st_d["inflate"] == "device" and st_h["inflate"] == "host" and st_d["bytes_moved"] > 0
        raw.close()


def test_corrupt_deflated_chunk_is_reported(workdir, tmp_path):
    """A flipped byte inside a compressed chunk: the device inflate ends with a status (never a hang) and the host raises Hdf5Error
    naming the chunk -- or, if the damaged stream still decodes to the right size, the data differ and nothing crashes."""
    from lorenzcycletoolkit_amd.hdf5_lite import Hdf5Error
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = bytearray(open(os.path.join(root, "tests", "golden", "hdf5", "packed_chunked_earliest.nc"), "rb").read())
    (workdir / "inputs" / "namelist").write_text(ERA5_NAMELIST)
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;30\nmin_lat;-40\nmax_lat;30\n")
    df = ds.read_namelist("inputs/namelist")
    clean = str(tmp_path / "clean.nc")
    open(clean, "wb").write(blob)
    raw = ds.open_raw(clean, df)
    info = raw.variables["t"].data.chunk_streams()
    addr, size, _plain = info["table"][(1, 0, 0, 0)]
    raw.close()
    outcomes = set()
    for k, at in enumerate((2, 5, size // 2, size - 6)):        # block header, code lengths, the tokens, the end
        bad = bytearray(blob)
        bad[addr + at] ^= 0x5A
        path = str(tmp_path / f"bad{k}.nc")
        open(path, "wb").write(bad)
        raw = ds.open_raw(path, df)
        plan = ingest.make_plan(raw, argparse.Namespace(fixed=True, track=False, trackfile=None))
        try:
            ingest.device_cube(raw.variables["t"], plan, inflate="device")
            outcomes.add("decoded")
        except Hdf5Error as e:
            assert "(1, 0, 0, 0)" in str(e) and "did not inflate" in str(e)
            outcomes.add("reported")
        raw.close()
    assert "reported" in outcomes


def test_fletcher32_is_verified_on_the_device_path(workdir, tmp_path):
    """packed_unlimited_latest.nc carries a Fletcher-32 after every deflated chunk: lec_inflate checks it before it inflates (the
    clean file passes, so the sum is formed the way the HDF5 library forms it); one flipped bit in the LAST byte of a stream -- inside
    zlib's own adler32 trailer, which no decoder needs -- is caught by it, on the device as on the host."""
    from lorenzcycletoolkit_amd.hdf5_lite import Hdf5Error
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = bytearray(open(os.path.join(root, "tests", "golden", "hdf5", "packed_unlimited_latest.nc"), "rb").read())
    (workdir / "inputs" / "namelist").write_text(ERA5_NAMELIST)
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;30\nmin_lat;-40\nmax_lat;30\n")
    df = ds.read_namelist("inputs/namelist")
    args = argparse.Namespace(fixed=True, track=False, trackfile=None)
    clean = str(tmp_path / "clean.nc")
    open(clean, "wb").write(blob)
    raw = ds.open_raw(clean, df)
    info = raw.variables["u"].data.chunk_streams()
    assert info["fletcher32"]
    addr, size, _plain = info["table"][(2, 0, 0, 0)]
    plan = ingest.make_plan(raw, args)
    good = ingest.device_cube(raw.variables["u"], plan, inflate="device")
    assert torch.equal(torch.nan_to_num(good, nan=-1e30), torch.nan_to_num(ingest.device_cube(raw.variables["u"], plan, inflate="host"), nan=-1e30))
    raw.close()
    blob[addr + size - 1] ^= 0x04
    bad = str(tmp_path / "bad.nc")
    open(bad, "wb").write(blob)
    raw = ds.open_raw(bad, df)
    plan = ingest.make_plan(raw, args)
    with pytest.raises(Hdf5Error, match="fletcher32"):
        ingest.device_cube(raw.variables["u"], plan, inflate="device")
    with pytest.raises(Hdf5Error, match="fletcher32"):
        ingest.device_cube(raw.variables["u"], plan, inflate="host")
    raw.close()


ERA5_NAMES = {"tair": "t", "u": "u", "v": "v", "omega": "w", "geo": "z", "lat": "latitude", "lon": "longitude", "level": "level", "time": "time"}
ERA5_NAMELIST = (";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
                 "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
                 "Time;time\nVertical Level;level\n")


@pytest.mark.parametrize("fill,offset,want", [(True, True, torch.float32), (False, True, torch.float64), (False, False, torch.float32)])
def test_lec_ingest_cube_equals_the_oracle_decode(workdir, fill, offset, want):
    """``lec_ingest`` pinned independently of the package's own host decoder: the cube it writes for every variable of an
    ERA5-style int16 file (0..360 longitudes, N -> S latitudes, hPa levels incl. 5 hPa, fill values) must equal, bit for bit
    and NaN for NaN, the ORACLE's restatement of the reference's decode + process_data + slice_domain (oracle/cf_decode.py:
    xarray 2024.2.0's mask / scale-offset coders with their dtype rules) -- float32 or float64 as that decode gives it."""
    from oracle import cf_decode as cf
    path = str(workdir / "packed.nc")
    _write_packed(path, nt=5, fill=fill, offset=offset)
    (workdir / "inputs" / "namelist").write_text(ERA5_NAMELIST)
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;20\nmin_lat;-45\nmax_lat;30\n")
    args = argparse.Namespace(fixed=True, track=False, trackfile=None)
    df = ds.read_namelist("inputs/namelist")
    raw = ds.open_raw(path, df)
    plan = ingest.make_plan(raw, args)
    ref = cf.prepare(path, ERA5_NAMES, fixed_limits=(-60, 20, -45, 30))
    assert np.array_equal(plan.lat, ref.lat) and np.array_equal(plan.lon, ref.lon) and np.array_equal(plan.level, ref.level)
    assert np.array_equal(plan.time_s, ref.time_s)
    for role, name in (("tair", "t"), ("u", "u"), ("v", "v"), ("omega", "w"), ("geopt", "z")):
        cube = ingest.device_cube(raw.variables[name], plan)
        want_np = getattr(ref, role)
        assert cube.dtype == want and str(want_np.dtype) == str(want).replace("torch.", ""), (role, cube.dtype, want_np.dtype)
        assert np.array_equal(cube.cpu().numpy(), want_np, equal_nan=True), role
        wide = ingest.device_cube(raw.variables[name], plan, out_dtype=np.float64)        # storage may be wider than the decode: exact
        assert wide.dtype == torch.float64 and np.array_equal(wide.cpu().numpy(), want_np.astype(np.float64), equal_nan=True)
    assert bool(np.isnan(ref.v).any()) == fill
    raw.close()


def test_moving_framework_streamed_equals_resident(workdir, golden_dir):
    """The semi-Lagrangian framework on the streamed path: per-time-step boxes over chunked ingest, dT/dt differentiated on the
    device over the track-selected times (select_area.py:297-313, lec_moving_framework.py:639-730) -- the same bits as the
    host-prepared resident path, whatever the chunking."""
    shutil.copy(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), workdir / "inputs" / "track")
    infile = os.path.join(golden_dir, "testdata_NCEP-R2.nc")
    args = argparse.Namespace(fixed=False, track=True, trackfile="inputs/track", residuals=True, infile=infile, cdsapi=False)
    df = ds.read_namelist("inputs/namelist")
    host = ds.prepare_data(args, "inputs/namelist")
    track = ds.read_track("inputs/track")
    limits = [(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(track["Lat"], track["Lon"])]
    a = BoxData(host, df, args=args, boxes_limits=limits).result
    for chunk in (1, 2, 8):
        st = ingest.prepare_streamed(args, "inputs/namelist", chunk_steps=chunk)
        b = BoxData(st, df, args=args, boxes_limits=limits).result
        st.raw.close()
        assert torch.equal(a.scalars, b.scalars) and torch.equal(a.levels, b.levels), chunk
    assert torch.isfinite(a.scalars).all()


@pytest.mark.parametrize("fill,storage", [(False, "float64"), (True, "float32")])
def test_moving_framework_packs_each_steps_box(workdir, fill, storage):
    """The streamed moving framework hands stage 1 a BOX-PACKED series (lec_ingest gathers each step's box alone; include/lec_hip.h):
    fp64 storage with dT/dt as a cube of its own (lec_dtdt), fp32 storage with T of the two neighbouring steps.  Boxes of several
    sizes, one at the domain's southern edge, a file that needs the longitude wrap and every sort: the same bits as the same path
    with whole-crop cubes (packed=False) and as the host-prepared resident path; the 850-hPa slices kept for the diagnostics too."""
    path = str(workdir / "packed.nc")
    _write_packed(path, nt=7, fill=fill)                 # int16 + scale + offset: float64 in the reference's xarray without a fill value
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
        "Time;time\nVertical Level;level\n")
    (workdir / "inputs" / "track").write_text(
        "time;Lat;Lon;length;width\n" + "".join(f"2020-01-0{1 + t // 4}-{6 * (t % 4):02d}00;{la};{lo};{ln};{wd}\n" for t, (la, lo, ln, wd) in enumerate(
            [(-30, -20, 30, 40), (-27.5, -15, 30, 40), (-25, -10, 25, 45), (-40, 0, 40, 30), (-20, 5, 30, 40), (-20, 10, 20, 40), (-17.5, 15, 30, 50)])))
    args = argparse.Namespace(fixed=False, track=True, trackfile="inputs/track", residuals=True, infile=path, cdsapi=False)
    df = ds.read_namelist("inputs/namelist")
    host = ds.prepare_data(args, "inputs/namelist")
    track = ds.read_track("inputs/track")
    limits = [(lo - w / 2, lo + w / 2, la - ln / 2, la + ln / 2) for la, lo, ln, w in zip(track["Lat"], track["Lon"], track["length"], track["width"])]
    a = BoxData(host, df, args=args, boxes_limits=limits).result
    for chunk in (2, 7):
        got = {}
        for packed in (True, False):
            st = ingest.prepare_streamed(args, "inputs/namelist")
            stats = {}
            got[packed] = (ingest.lec_streamed(st.raw, st.plan, df, limits, per_step_boxes=True, chunk_steps=chunk, packed=packed, stats=stats,
                                               keep_level=85000.0), stats)
            torch.cuda.synchronize()
            st.raw.close()
            assert stats["storage"] == storage
        (p, ps), (c, cs) = got[True], got[False]
        for r in (p, c):
            assert torch.equal(a.scalars, r.scalars) and np.array_equal(a.levels.cpu().numpy(), r.levels.cpu().numpy(), equal_nan=True), (chunk, storage)
        for k in ("u", "v", "geopt"):
            assert torch.equal(ps["level_slices"][k], cs["level_slices"][k]), k
    assert torch.isfinite(a.scalars[:, :4]).all()


def test_mixed_dtype_file_on_both_paths(workdir):
    """A file that mixes dtypes -- float32 T, u, v, w with an int16-packed geopotential that has an add_offset but no fill value
    (float64 in the reference's decode) -- used to fail in the resident path ('all field cubes must share ... dtype').
    Both paths promote the five cubes to the widest dtype (exact) and agree bit for bit."""
    from scipy.io import netcdf_file
    rng = np.random.default_rng(9)
    nt, lev, lat, lon = 4, np.array([1000.0, 850.0, 700.0, 500.0, 300.0]), np.linspace(-50, -20, 13), np.linspace(-70, -30, 17)
    path = str(workdir / "mixed.nc")
    f = netcdf_file(path, "w", version=2)
    for n, s in (("time", nt), ("level", lev.size), ("lat", lat.size), ("lon", lon.size)):
        f.createDimension(n, s)
    tv = f.createVariable("time", "d", ("time",)); tv[:] = 6.0 * np.arange(nt); tv.units = "hours since 2001-01-01 00:00:00"
    lv = f.createVariable("level", "d", ("level",)); lv[:] = lev; lv.units = "hPa"
    f.createVariable("lat", "d", ("lat",))[:] = lat
    f.createVariable("lon", "d", ("lon",))[:] = lon
    p = lev[None, :, None, None] / 1000.0
    shp = (nt, lev.size, lat.size, lon.size)
    fields = {"t": 288 * p ** 0.19 + rng.standard_normal(shp), "u": 10 + 5 * rng.standard_normal(shp), "v": 3 * rng.standard_normal(shp),
              "w": 0.1 * rng.standard_normal(shp)}
    for name, a in fields.items():
        f.createVariable(name, "f", ("time", "level", "lat", "lon"))[:] = a.astype(np.float32)
    z = 9.80665 * 7000 * np.log(1 / p) + 80 * rng.standard_normal(shp)
    scale, off = (z.max() - z.min()) / 65000.0, 0.5 * (z.max() + z.min())
    zv = f.createVariable("z", "h", ("time", "level", "lat", "lon"))
    zv[:] = np.clip(np.round((z - off) / scale), -32000, 32000).astype(np.int16)
    zv.scale_factor = float(scale); zv.add_offset = float(off)
    f.close()
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;lon\nLatitude;lat\nTime;time\nVertical Level;level\n")
    limits = (-65.0, -35.0, -45.0, -25.0)
    (workdir / "inputs" / "box_limits").write_text("min_lon;-65\nmax_lon;-35\nmin_lat;-45\nmax_lat;-25\n")
    a, b, stats = _both_paths(path, "inputs/namelist", limits, 2)
    assert stats["storage"] == "float64" and stats["decode"]["tair"] == "float32" and stats["decode"]["geopt"] == "float64"
    assert torch.equal(a.scalars, b.scalars) and torch.equal(a.levels, b.levels) and torch.isfinite(a.scalars).all()


@pytest.mark.parametrize("chunk_steps", [4, 36])
def test_registered_file_memory_equals_staged(workdir, golden_dir, chunk_steps):
    """staging="registered": the mapped file's own pages are registered with the HIP runtime and copied from directly (no pinned
    staging copy); "staged": the thread-pool copy into pinned buffers.  Same bytes on the device, same results, bit for bit; the
    Catarina box is the file's whole latitude range after the crop?  No: the fixed box crops -- so the full-range case is built by
    giving a box that spans every latitude of the file."""
    args = argparse.Namespace(fixed=True, track=False, trackfile=None, residuals=True)
    df = ds.read_namelist("inputs/namelist")
    infile = os.path.join(golden_dir, "Catarina_NCEP-R2.nc")
    raw = ds.open_raw(infile, df)
    lat = np.sort(raw.lat)
    (workdir / "inputs" / "box_limits").write_text(f"min_lon;-55\nmax_lon;-36\nmin_lat;{lat[0]}\nmax_lat;{lat[-1]}\n")
    plan = ingest.make_plan(raw, args)
    limits = (-55.0, -36.0, float(lat[0]), float(lat[-1]))
    sa, sb = {}, {}
    a = ingest.lec_fixed_streamed(raw, plan, df, limits, chunk_steps=chunk_steps, staging="staged", stats=sa)
    b = ingest.lec_fixed_streamed(raw, plan, df, limits, chunk_steps=chunk_steps, staging="registered", stats=sb)
    c = ingest.lec_fixed_streamed(raw, plan, df, limits, chunk_steps=chunk_steps, stats={})          # auto
    torch.cuda.synchronize()
    assert sa["staging"] == "staged" and sb["staging"] == "registered" and sb["register_calls"] >= 1 and sb["bytes_moved"] == sa["bytes_moved"]
    for r in (b, c):
        assert torch.equal(a.scalars, r.scalars) and torch.equal(a.levels, r.levels) and torch.equal(a.nanflag, r.nanflag)
    raw.close()
    # a latitude BAND of a larger file is staged (registering pins -- and so reads -- whole levels): asked for by name it is refused, auto stages
    raw = ds.open_raw(os.path.join(golden_dir, "testdata_NCEP-R2.nc"), df)
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;-30\nmin_lat;-42.5\nmax_lat;-17.5\n")
    plan2 = ingest.make_plan(raw, args)
    assert plan2.lat.size < raw.lat.size
    with pytest.raises(ValueError, match="registered"):
        ingest.lec_fixed_streamed(raw, plan2, df, (-60.0, -30.0, -42.5, -17.5), staging="registered")
    st = {}
    ingest.lec_fixed_streamed(raw, plan2, df, (-60.0, -30.0, -42.5, -17.5), stats=st)
    assert st["staging"] == "staged"
    raw.close()


def test_random_file_layouts_device_decode_equals_the_oracle(workdir):
    """60 cases of tests/soak_ingest.py: lec_ingest (int8 / int16 / int32 / float32 / float64 sources, every combination of packing
    and fill attributes, rolled longitudes, reversed latitudes and levels) against the oracle's decode + process_data +
    slice_domain, dtype for dtype, bit for bit -- and the host preparation with it."""
    from tests import soak_ingest as soak
    (workdir / "inputs" / "namelist").write_text(soak.NAMELIST)
    rng = np.random.default_rng(12)
    fails = []
    for case in range(60):
        fails += soak.one_case(rng, case, str(workdir), gpu=True)
    assert not fails, fails[:3]


def test_chunk_stager_grows_for_scattered_steps_of_time_spanning_chunks():
    """Chunks of 3 time steps x 2 levels x 4 x 5 points; a plan that takes every third step needs one time-chunk PER step -- more than the
    stager sized its buffers for (consecutive steps) -- and a band of levels and latitudes: the slot's buffers grow, the rows are right."""
    import zlib
    rng = np.random.default_rng(5)
    shape, chunk = (40, 5, 9, 10), (3, 2, 4, 5)
    a = rng.integers(-30000, 30000, shape).astype("<i2")
    table, blobs, at = {}, [], 64                               # (a file does not start with a chunk)
    for t in range(0, 40, 3):
        for k in range(0, 5, 2):
            for j in range(0, 9, 4):
                for i in range(0, 10, 5):
                    blk = np.zeros(chunk, dtype="<i2")
                    part = a[t: t + 3, k: k + 2, j: j + 4, i: i + 5]
                    blk[: part.shape[0], : part.shape[1], : part.shape[2], : part.shape[3]] = part
                    z = zlib.compress(blk.reshape(-1).view(np.uint8).reshape(-1, 2).T.tobytes(), 5)
                    table[(t, k, j, i)] = (at, len(z), False)
                    blobs.append(z)
                    at += len(z)
    blob = np.frombuffer(b"\0" * 64 + b"".join(blobs), dtype=np.uint8)

    class Var:
        pass
    v = Var()
    v.shape, v.dtype = shape, np.dtype("<i2")
    info = {"chunk": chunk, "shuffle": True, "table": table, "map": blob}
    var = ds.RawVariable(v, None, None, None)
    levels, j0, j1 = np.array([1, 2, 4]), 2, 7
    st = ingest._ChunkStager(var, info, 5, "cuda:0", levels, j0, j1, slots=2)
    sized = st.max_chunks[0]
    steps = np.array([0, 3, 6, 9, 12])                          # five steps, five time-chunks (5 // 3 + 2 = 3 were planned)
    for slot, use in ((0, steps), (1, steps[:2]), (0, steps[::-1].copy()), (1, np.array([39, 0, 20]))):      # (the last: the whole file's span for three steps)
        st.stage(slot, use, 0)
        with torch.cuda.stream(st.streams[slot]):
            st.upload(slot, 0, len(use))
        st.check(slot)
        got = st.raw_dev[slot][: len(use)].cpu().numpy().reshape(len(use), 3, j1 - j0 + 1, 10)
        assert np.array_equal(got, a[use][:, levels][:, :, j0: j1 + 1]), slot
    assert st.max_chunks[0] > sized and st.max_chunks[1] == sized and st.tmap_pin[1].numel() >= 42


def test_inflate_and_slot_arguments_are_validated(workdir, golden_dir):
    """``inflate="device"`` on a variable that is not a chunked NetCDF-4 one is refused (no silent host inflate), unknown values and
    fewer than two pipeline slots too."""
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    args = argparse.Namespace(fixed=True, track=False, trackfile=None)
    df = ds.read_namelist("inputs/namelist")
    raw = ds.open_raw(os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), df)          # classic NetCDF: plain mapped memory
    plan = ingest.make_plan(raw, args)
    limits = [(-55.0, -36.0, -35.0, -20.0)]
    with pytest.raises(ValueError, match="inflate='device'"):
        ingest.lec_streamed(raw, plan, df, limits, inflate="device")
    with pytest.raises(ValueError, match="inflate must be"):
        ingest.lec_streamed(raw, plan, df, limits, inflate="gpu")
    with pytest.raises(ValueError, match="slots"):
        ingest.lec_streamed(raw, plan, df, limits, slots=1)
    a = ingest.lec_streamed(raw, plan, df, limits, slots=2)
    b = ingest.lec_streamed(raw, plan, df, limits, slots=4, chunk_steps=3)
    assert torch.equal(a.scalars, b.scalars) and torch.equal(a.levels, b.levels)
    raw.close()


def test_registration_refused_in_mid_run_falls_back_to_pinned_staging(workdir, monkeypatch):
    """The compressed chunks of a deflated NetCDF-4 file go to the GPU straight from the registered file pages; when the runtime refuses
    a registration in mid-run (a locked-memory limit), the variable goes on through pinned staging buffers: same bits, no error."""
    from lorenzcycletoolkit_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
        "Time;time\nVertical Level;level\n")
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;30\nmin_lat;-40\nmax_lat;30\n")
    args = argparse.Namespace(fixed=True, track=False, trackfile=None)
    df = ds.read_namelist("inputs/namelist")
    raw = ds.open_raw(os.path.join(root, "tests", "golden", "hdf5", "packed_interleaved_v18.nc"), df)
    plan = ingest.make_plan(raw, args)
    lim = [(-60.0, 30.0, -40.0, 30.0)]
    st = {}
    ref = ingest.lec_streamed(raw, plan, df, lim, chunk_steps=1, stats=st)
    assert st["staging"] == "registered" and st["inflate"] == "device" and st["register_calls"] >= 1
    staged = ingest.lec_streamed(raw, plan, df, lim, chunk_steps=1, staging="staged")
    real, calls = ingest.RegisteredSpans.ensure, [0]

    def flaky(self, lo, hi, use):
        calls[0] += 1
        if calls[0] > 4:
            raise _lib.LecLibraryError("lec_host_register failed (code 3): out of locked memory (simulated)")
        return real(self, lo, hi, use)
    monkeypatch.setattr(ingest.RegisteredSpans, "ensure", flaky)
    got = ingest.lec_streamed(raw, plan, df, lim, chunk_steps=1)
    assert calls[0] > 4
    same = lambda a, b: bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())      # (a dropped level is a NaN column of the tables)
    for r in (staged, got):
        assert same(r.scalars, ref.scalars) and same(r.levels, ref.levels) and torch.equal(r.nanflag, ref.nanflag)
    raw.close()
