"""CPU tests of the input side of the drop-in (namelist / box_limits / track parsing, NetCDF-3
reader, process_data / slice_domain semantics) against the oracle's loader and the reference's files."""
import argparse
import os
import shutil

import numpy as np
import pandas as pd
import pytest

from lorenzcycletoolkit_amd import dataset as ds
from oracle import lec_oracle as o


def _args(**kw):
    base = dict(infile="", fixed=False, track=False, choose=False, residuals=True, cdsapi=False, trackfile="inputs/track",
                box_limits="inputs/box_limits", mpas=False, plots=False, outname=None, verbosity=False, zeta=False)
    base.update(kw)
    return argparse.Namespace(**base)


@pytest.fixture
def workdir(tmp_path, golden_dir, monkeypatch):
    """A scratch working directory laid out like the reference expects (inputs/...)."""
    os.makedirs(tmp_path / "inputs")
    shutil.copy(os.path.join(golden_dir, "inputs", "namelist_NCEP-R2"), tmp_path / "inputs" / "namelist")
    monkeypatch.chdir(tmp_path)
    return tmp_path


def test_namelist_and_box_limits(workdir, golden_dir):
    nl = ds.read_namelist("inputs/namelist")
    assert nl.loc["Air Temperature"]["Variable"] == "TMP_2_ISBL"
    assert nl.loc["Vertical Level"]["Variable"] == "lv_ISBL3"
    assert ds.field_scale(nl, "Geopotential Height") == o.G
    assert ds.field_scale(nl, "Air Temperature") == 1.0
    assert ds.read_box_limits(os.path.join(golden_dir, "inputs", "box_limits_Reg1")) == (-60.0, -30.0, -42.5, -17.5)
    with pytest.raises(FileNotFoundError):
        ds.read_box_limits("inputs/nope")
    (workdir / "bad").write_text("min_lon;10\nmax_lon;-10\nmin_lat;0\nmax_lat;5\n")
    with pytest.raises(ValueError, match="min_lon"):
        ds.read_box_limits("bad")
    (workdir / "short").write_text("min_lon;10\nmax_lon;20\n")
    with pytest.raises(ValueError, match="missing"):
        ds.read_box_limits("short")
    (workdir / "nl2").write_text(";standard_name;Variable;Units\nAir Temperature;t;T;K\n")
    with pytest.raises(ValueError, match="missing"):
        ds.read_namelist("nl2")


def test_track_parsing(workdir, golden_dir):
    tr = ds.read_track(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"))
    assert list(tr.columns) == ["Lat", "Lon"] and len(tr) == 5
    assert tr.index[1] == pd.Timestamp("2005-08-08 06:00")
    (workdir / "t1").write_text("time,Lat,Lon\n2005-08-08-0000,-22.5,-45\n")
    assert len(ds.read_track("t1")) == 1
    (workdir / "t2").write_text("time;Lat;Lon\n2005/08/08 00:00;-22.5;-45\n")
    with pytest.raises(ValueError, match="date format"):
        ds.read_track("t2")
    (workdir / "t3").write_text("time;Latitude;Lon\n2005-08-08-0000;-22.5;-45\n")
    with pytest.raises(ValueError, match="missing required columns"):
        ds.read_track("t3")


def test_prepare_data_fixed_matches_oracle_loader(workdir, golden_dir):
    shutil.copy(os.path.join(golden_dir, "inputs", "box_limits_Reg1"), workdir / "inputs" / "box_limits")
    args = _args(infile=os.path.join(golden_dir, "testdata_NCEP-R2.nc"), fixed=True)
    data = ds.prepare_data(args, "inputs/namelist")
    ref = o.crop_domain(o.load_ncep_sample(args.infile), -60, -30, -42.5, -17.5)
    assert np.array_equal(data.lat, ref.lat) and np.array_equal(data.lon, ref.lon)
    assert np.array_equal(data.level, ref.level) and data.level[0] == 60000.0
    assert np.array_equal(data.time_s, ref.time_s)
    assert np.array_equal(data.variables["TMP_2_ISBL"], ref.tair)
    assert np.array_equal(data.variables["V_VEL_2_ISBL"], ref.omega)
    assert data.variables["TMP_2_ISBL"].dtype == np.float32
    assert str(data.time[0])[:13] == "2005-08-08T00"


def test_prepare_data_wraps_longitudes_and_drops_top_levels(workdir, golden_dir):
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    args = _args(infile=os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), fixed=True)
    data = ds.prepare_data(args, "inputs/namelist")
    assert data.lon[0] == -55.0 and data.lon[-1] == -37.5          # 305..322.5 E wrapped
    assert np.all(np.diff(data.lat) > 0) and data.lat[0] == -35.0
    assert data.level[0] == 1000.0 and data.level[-1] == 100000.0 and data.level.size == 17
    ref = o.crop_domain(o.load_ncep_sample(args.infile), -55, -36, -35, -20)
    assert np.array_equal(data.variables["HGT_2_ISBL"] * np.float32(1), ref.geopt / np.float32(o.G)) or \
        np.allclose(data.variables["HGT_2_ISBL"], ref.geopt / o.G, rtol=1e-6)


def test_prepare_data_track_selects_times_and_extent(workdir, golden_dir):
    shutil.copy(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), workdir / "inputs" / "track")
    args = _args(infile=os.path.join(golden_dir, "testdata_NCEP-R2.nc"), track=True)
    data = ds.prepare_data(args, "inputs/namelist")
    tr = ds.read_track("inputs/track")
    ref = o.crop_domain_track(o.load_ncep_sample(args.infile), tr.Lat.values, tr.Lon.values)
    assert np.array_equal(data.lat, ref.lat) and np.array_equal(data.lon, ref.lon)
    assert data.variables["TMP_2_ISBL"].shape == ref.tair.shape
    (workdir / "inputs" / "track").write_text("time;Lat;Lon\n2005-08-07-0000;-22.5;-45\n2005-08-07-0600;-22.5;-45\n")
    with pytest.raises(ValueError, match="earlier"):
        ds.prepare_data(args, "inputs/namelist")


def test_open_dataset_rejects_unknown_containers(workdir):
    (workdir / "x.nc").write_bytes(b"GRIB" + b"0" * 64)
    with pytest.raises(ValueError, match="neither"):
        ds.open_dataset("x.nc", ds.read_namelist("inputs/namelist"))
    (workdir / "y.nc").write_bytes(b"\x89HDF\r\n\x1a\n" + b"\xff" * 64)          # HDF5 signature, garbage behind it
    with pytest.raises(ValueError):
        ds.open_dataset("y.nc", ds.read_namelist("inputs/namelist"))
    with pytest.raises(FileNotFoundError):
        ds.open_dataset("missing.nc", ds.read_namelist("inputs/namelist"))


def test_shipped_input_presets_follow_the_contract(tmp_path, monkeypatch):
    """./inputs carries the preset namelists / box limits / tracks of the drop-in contract; a run copies one to
    inputs/namelist exactly as the reference's tests do (shutil.copy('inputs/namelist_NCEP-R2', 'inputs/namelist'))."""
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copytree(os.path.join(root, "inputs"), tmp_path / "inputs")
    monkeypatch.chdir(tmp_path)
    presets = sorted(f for f in os.listdir("inputs") if f.startswith("namelist_"))
    assert presets == ["namelist_ERA5", "namelist_ERA5-cdsapi", "namelist_ERA5-copernicus", "namelist_ERA5-copernicus-new",
                       "namelist_MPAS-A", "namelist_NCEP-R1", "namelist_NCEP-R2"]
    for name in presets:
        shutil.copy(os.path.join("inputs", name), "inputs/namelist")
        df = ds.read_namelist("inputs/namelist")
        geo = "Geopotential" if "Geopotential" in df.index else "Geopotential Height"
        for role in ("Air Temperature", "Omega Velocity", "Eastward Wind Component", "Northward Wind Component", geo):
            assert ds.field_scale(df, role) > 0                       # the preset's units are understood
    assert ds.read_box_limits("inputs/box_limits") == (-60.0, -30.0, -42.5, -17.5)
    assert ds.read_box_limits("inputs/box_limits-testcase") == (-53.0, -44.0, -31.0, -24.0)
    for name, n in (("track_testdata_NCEP-R2", 5), ("track_testdata_ERA5", 5)):
        tr = ds.read_track(os.path.join("inputs", name))
        assert len(tr) == n and list(tr.columns) == ["Lat", "Lon"]
    # the fixtures the tests use are these very files
    for name in ("namelist_NCEP-R2", "namelist_ERA5", "box_limits_Reg1", "track_testdata_NCEP-R2", "track_testdata_ERA5"):
        assert open(os.path.join("inputs", name)).read().split() == open(os.path.join(root, "tests", "golden", "inputs", name)).read().split()
