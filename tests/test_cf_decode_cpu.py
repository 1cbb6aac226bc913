"""The data preparation, pinned independently of the package: oracle/cf_decode.py restates the reference's decode (xarray
2024.2.0 on NumPy 2) and its process_data / slice_domain; the package's host path (dataset.py) must reproduce it bit for bit.
The GPU twin of this file is tests/test_gpu_ingest.py::test_lec_ingest_cube_equals_the_oracle_decode."""
import argparse
import os

import numpy as np
import pytest

from lorenzcycletoolkit_amd import dataset as ds
from oracle import cf_decode as cf
from oracle import lec_oracle as o
from tests.helpers import write_packed_era5_style

ERA5_NAMES = {"tair": "t", "u": "u", "v": "v", "omega": "w", "geo": "z", "lat": "latitude", "lon": "longitude", "level": "level", "time": "time"}
ERA5_NAMELIST = (";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
                 "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
                 "Time;time\nVertical Level;level\n")


@pytest.mark.parametrize("src,scale,offset,fill,want", [
    (np.int16, 0.01, 250.0, -32767, np.float32),      # ERA5: fill value present -> the mask coder has made the data float32 already
    (np.int16, 0.01, 250.0, None, np.float64),        # no fill value: "any offset at all -> float64"
    (np.int16, 0.01, None, None, np.float32),         # a scale factor alone is safe in float32
    (np.int16, 0.01, None, -32767, np.float32),
    (np.int32, 0.01, 250.0, -2147483647, np.float64),
    (np.float32, None, None, 9.96921e36, np.float32),
    (np.float32, 2.0, 1.0, None, np.float32),
    (np.float64, None, None, None, np.float64),
])
def test_decode_dtype_rules_and_values(src, scale, offset, fill, want):
    rng = np.random.default_rng(7)
    raw = (rng.integers(-32000, 32000, size=(4, 50)).astype(src) if np.dtype(src).kind == "i"
           else rng.standard_normal((4, 50)).astype(src) * 1000)
    if fill is not None:
        raw[1, 3] = raw[2, 7] = np.asarray(fill).astype(src)
    # a file holds the fill value in the variable's own type
    attrs = {k: v for k, v in (("scale_factor", scale), ("add_offset", offset),
                               ("_FillValue", None if fill is None else np.asarray(fill).astype(src)[()])) if v is not None}
    ref = cf.decode_cf_variable(raw.astype(np.dtype(src).newbyteorder(">")), attrs)      # classic NetCDF is big-endian
    got = ds.decode_values(raw, scale, offset, None if fill is None else float(np.asarray(fill).astype(src)))
    assert ref.dtype == want and got.dtype == want
    assert np.array_equal(ref, got, equal_nan=True)
    assert ds.decode_dtypes(np.dtype(src), scale, offset, fill)[1] == want
    if fill is not None:
        assert np.isnan(ref[1, 3]) and np.isnan(ref[2, 7]) and np.isnan(ref).sum() == 2
    if want == np.float32 and scale is not None and np.dtype(src).kind == "i":
        # one value by hand: float32 data, float64 attribute: computed in fp64, rounded to float32 after EACH operation
        x = np.float32(np.float64(np.float32(raw[0, 0])) * np.float64(scale))
        if offset is not None:
            x = np.float32(np.float64(x) + np.float64(offset))
        assert ref[0, 0] == x


def test_axis_handling_is_the_pinned_loader(golden_dir):
    """cf_decode.prepare on the reference's float32 sample == lec_oracle.load_ncep_sample + crop, the loader whose outputs
    reproduce the reference's committed CSVs (tests/test_oracle_golden.py)."""
    names = {"tair": "TMP_2_ISBL", "u": "U_GRD_2_ISBL", "v": "V_GRD_2_ISBL", "omega": "V_VEL_2_ISBL", "geo": "HGT_2_ISBL",
             "lat": "lat_2", "lon": "lon_2", "level": "lv_ISBL3", "time": "initial_time0_hours"}
    limits = (-55, -36, -35, -20)
    a = cf.prepare(os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), names, fixed_limits=limits, geo_is_height=True)
    b = o.crop_domain(o.load_ncep_sample(os.path.join(golden_dir, "Catarina_NCEP-R2.nc")), *limits)
    for k in ("tair", "u", "v", "omega", "geopt", "lat", "lon", "level", "time_s"):
        x, y = getattr(a, k), getattr(b, k)
        assert x.dtype == y.dtype and np.array_equal(x, y), k


@pytest.mark.parametrize("fill,offset,want", [(True, True, np.float32), (False, True, np.float64), (False, False, np.float32)])
def test_host_preparation_equals_the_oracle_on_an_era5_style_file(tmp_path, monkeypatch, fill, offset, want):
    """int16-packed, 0..360 longitudes, N -> S latitudes, hPa levels incl. 5 hPa: the package's prepare_data against the
    oracle's restatement, every element of every variable and coordinate."""
    os.makedirs(tmp_path / "inputs")
    (tmp_path / "inputs" / "namelist").write_text(ERA5_NAMELIST)
    (tmp_path / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;20\nmin_lat;-45\nmax_lat;30\n")
    monkeypatch.chdir(tmp_path)
    path = str(tmp_path / "packed.nc")
    write_packed_era5_style(path, nt=4, fill=fill, offset=offset)
    args = argparse.Namespace(infile=path, fixed=True, track=False, trackfile=None, cdsapi=False)
    got = ds.prepare_data(args, "inputs/namelist")
    ref = cf.prepare(path, ERA5_NAMES, fixed_limits=(-60, 20, -45, 30))
    assert np.array_equal(got.lat, ref.lat) and np.array_equal(got.lon, ref.lon) and np.array_equal(got.level, ref.level)
    assert np.array_equal(got.time_s, ref.time_s)
    for role, name in (("tair", "t"), ("u", "u"), ("v", "v"), ("omega", "w"), ("geopt", "z")):
        x, y = got.variables[name], getattr(ref, role)
        assert x.dtype == want and y.dtype == want, (role, x.dtype, y.dtype)
        assert np.array_equal(x, y, equal_nan=True), role
    assert np.isnan(ref.v).any() == fill


def test_random_file_layouts_host_preparation_equals_the_oracle(tmp_path, monkeypatch):
    """60 cases of tests/soak_ingest.py (random storage types int8 / int16 / int32 / float32 / float64, packing and fill attributes in
    float32 or float64, axis orders and units, boxes): dataset.prepare_data against the oracle's decode, dtype for dtype, bit for bit."""
    from tests import soak_ingest as soak
    os.makedirs(tmp_path / "inputs")
    (tmp_path / "inputs" / "namelist").write_text(soak.NAMELIST)
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(11)
    fails = []
    for case in range(60):
        fails += soak.one_case(rng, case, str(tmp_path), gpu=False)
    assert not fails, fails[:3]
