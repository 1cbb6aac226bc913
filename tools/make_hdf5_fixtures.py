#!/opt/conda/bin/python3.9
"""Writes the small NetCDF-4-style HDF5 fixtures under tests/golden/hdf5/ with h5py (available only in this
container's /opt/conda python, not on the GPU box -- hence committed fixtures).  They imitate what netCDF-C /
h5netcdf produce: dimension scales with DIMENSION_LIST references, chunked + shuffle + deflate int16 variables with
scale_factor / add_offset / _FillValue, fixed- and variable-length string attributes, creation-order tracking, old-
and new-style groups.  The data are a pure function of the seed below, so tests regenerate the expected arrays.

    /opt/conda/bin/python3.9 tools/make_hdf5_fixtures.py
"""
import os

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "hdf5")


def fields(nt, nl, ny, nx, seed=7):
    rng = np.random.default_rng(seed)
    lev = np.array([1000, 850, 700, 500, 300, 200, 100, 50, 5][:nl], dtype=np.float64)
    lat = np.linspace(60.0, -60.0, ny)
    lon = np.arange(nx) * (360.0 / nx)
    p = (lev[None, :, None, None] * 100.0) / 1e5
    shp = (nt, nl, ny, nx)
    f = {
        "t": 288.0 * p ** 0.19 + 8.0 * np.cos(np.deg2rad(2 * lat))[None, None, :, None] * p + rng.standard_normal(shp),
        "u": 20.0 * np.cos(np.deg2rad(lat))[None, None, :, None] * (1 - p / 1.2) + 5 * rng.standard_normal(shp),
        "v": 3.0 * rng.standard_normal(shp),
        "w": 0.1 * rng.standard_normal(shp),
        "z": 9.80665 * 7000.0 * np.log(1.0 / p) + 100.0 * rng.standard_normal(shp),
    }
    return lev, lat, lon, f


def pack(a):
    lo, hi = a.min(), a.max()
    scale = (hi - lo) / 65000.0
    offset = 0.5 * (hi + lo)
    return np.clip(np.round((a - offset) / scale), -32000, 32000).astype(np.int16), float(scale), float(offset)


def write(path, libver, track_order, packed, chunks, vlen_units, many_attrs, nt=5, nl=6, ny=13, nx=24, unlimited=False, time_chunk=1,
          deflate=True, shuffle=True):
    lev, lat, lon, f = fields(nt, nl, ny, nx)
    with h5py.File(path, "w", libver=libver, track_order=track_order) as h:
        h.attrs["Conventions"] = np.string_("CF-1.6")
        coords = {"time": (6 * np.arange(nt)).astype(np.int32), "level": lev.astype(np.int32),
                  "latitude": lat.astype(np.float32), "longitude": lon.astype(np.float32)}
        for name, vals in coords.items():
            if unlimited and name == "time":     # netCDF's record dimension: extendible, hence chunked
                d = h.create_dataset(name, data=vals, maxshape=(None,), chunks=(4,))
            else:
                d = h.create_dataset(name, data=vals)
            d.make_scale(name)
        units = {"time": "hours since 2020-01-01 00:00:00", "level": "millibars", "latitude": "degrees_north", "longitude": "degrees_east"}
        for name, u in units.items():
            h[name].attrs["units"] = u if vlen_units else np.string_(u)
        for name, a in f.items():
            kw = {}
            if chunks:
                kw = dict(chunks=(time_chunk, 2, ny if time_chunk == 1 else 5, nx // 2), shuffle=shuffle)
                if deflate:
                    kw.update(compression="gzip", compression_opts=4)
            if unlimited:
                kw.update(maxshape=(None, nl, ny, nx), fletcher32=True)
            if packed:
                q, scale, offset = pack(a)
                if name == "v":
                    q[1, 0, :, :] = -32767
                d = h.create_dataset(name, data=q, **kw)
                d.attrs["scale_factor"] = np.float64(scale)
                d.attrs["add_offset"] = np.float64(offset)
                d.attrs["_FillValue"] = np.int16(-32767)
                d.attrs["missing_value"] = np.int16(-32767)
            else:
                d = h.create_dataset(name, data=a.astype(np.float32), **kw)
            d.attrs["units"] = "K" if vlen_units else np.string_("K")
            d.attrs["long_name"] = np.string_("field " + name)
            if many_attrs:
                for i in range(10):
                    d.attrs["extra_%02d" % i] = np.float32(i)
            for i, dn in enumerate(("time", "level", "latitude", "longitude")):
                d.dims[i].attach_scale(h[dn])
        if many_attrs:      # more than 8 links: dense link storage in new-style groups
            for i in range(6):
                h.create_dataset("pad_%d" % i, data=np.arange(3, dtype=np.int32))


def write_interleaved(path, nt=5, nl=6, ny=13, nx=24):
    """The same fields written the way a model writes a record file: step by step, all variables of a step before the next one -- the
    chunks of one variable are then scattered through the file between the other variables' (extendible time axis, v1 B-trees)."""
    lev, lat, lon, f = fields(nt, nl, ny, nx)
    with h5py.File(path, "w", libver=("earliest", "v108")) as h:
        coords = {"level": lev.astype(np.int32), "latitude": lat.astype(np.float32), "longitude": lon.astype(np.float32)}
        d = h.create_dataset("time", (nt,), dtype=np.int32, maxshape=(None,), chunks=(4,))
        d.make_scale("time")
        for name, vals in coords.items():
            h.create_dataset(name, data=vals).make_scale(name)
        for name, u in {"time": "hours since 2020-01-01 00:00:00", "level": "millibars", "latitude": "degrees_north", "longitude": "degrees_east"}.items():
            h[name].attrs["units"] = np.string_(u)
        packed = {}
        for name, a in f.items():
            q, scale, offset = pack(a)
            if name == "v":
                q[1, 0, :, :] = -32767
            packed[name] = q
            d = h.create_dataset(name, (nt, nl, ny, nx), dtype=np.int16, maxshape=(None, nl, ny, nx), chunks=(1, 2, ny, nx // 2),
                                 compression="gzip", compression_opts=4, shuffle=True)
            d.attrs["scale_factor"], d.attrs["add_offset"] = np.float64(scale), np.float64(offset)
            d.attrs["_FillValue"] = np.int16(-32767)
            d.attrs["missing_value"] = np.int16(-32767)
            d.attrs["units"] = np.string_("K")
            d.attrs["long_name"] = np.string_("field " + name)
            for i, dn in enumerate(("time", "level", "latitude", "longitude")):
                d.dims[i].attach_scale(h[dn])
        for t in range(nt):
            h["time"][t] = 6 * t
            for name in f:
                h[name][t] = packed[name][t]


def sparse_arrays():
    """What write_sparse() writes and what a reader must return for it (pure function: the test regenerates it)."""
    rng = np.random.default_rng(11)
    a = rng.integers(-30000, 30000, (6, 4, 10, 12)).astype(np.int16)          # written only in [0:4, 0:3, 0:7, 0:9]
    want_a = np.full(a.shape, -99, dtype=np.int16); want_a[:4, :3, :7, :9] = a[:4, :3, :7, :9]
    b = rng.integers(0, 60000, (1500, 2, 3)).astype(np.uint16)                 # 1500 one-step chunks, the second half never written
    want_b = np.zeros(b.shape, dtype=np.uint16); want_b[:750] = b[:750]
    c = rng.standard_normal((5, 8)).astype(np.float32)                         # a chunked variable nobody ever wrote to
    return a, want_a, b, want_b, np.zeros_like(c)


def write_sparse(path):
    """Chunks that were never written read as the dataset's HDF5 fill value (zeros if none is defined) -- NOT as its _FillValue
    attribute: fixed-array chunk indexes with undefined entries, a paged one (1500 chunks > 1024 per page) whose second page was
    never initialised, and a single-chunk dataset without storage.  (The round-3 soak against h5py, tools/soak_hdf5.py, found all
    three unread; netCDF-C writes every chunk of a variable it defines, h5py / h5netcdf writers need not.)"""
    a, _wa, b, _wb, c = sparse_arrays()
    with h5py.File(path, "w", libver="latest") as h:
        d = h.create_dataset("a", a.shape, dtype=np.int16, chunks=(2, 2, 4, 5), compression="gzip", shuffle=True, fillvalue=np.int16(-99))
        d[:4, :3, :7, :9] = a[:4, :3, :7, :9]
        d.attrs["_FillValue"] = np.int16(-32767)          # the CF attribute says one thing, the HDF5 fill value another: the latter fills
        d = h.create_dataset("b", b.shape, dtype=np.uint16, chunks=(1, 2, 3))
        d[:750] = b[:750]
        d.attrs["_FillValue"] = np.uint16(65535)
        h.create_dataset("c", c.shape, dtype=np.float32, chunks=c.shape)


def write_cds_new(path, nt=6, nl=6, ny=13, nx=24):
    """The layout the Copernicus CDS delivers since its 2024 relaunch (the reference's inputs/namelist_ERA5-copernicus-new:1-10):
    ``valid_time`` int64 "seconds since 1970-01-01", ``pressure_level`` float64 in hPa, DESCENDING, latitude / longitude float64, a
    scalar int64 ``number`` and a per-time variable-length STRING ``expver`` coordinate (both dropped by the reference,
    src/utils/preprocessing.py:291-296), float32 fields with shuffle + deflate, a NaN ``_FillValue``, GRIB_* attributes and a
    ``coordinates`` attribute.  The fields are those of fields() (same seed) rounded to float32; v carries a NaN level at step 1."""
    lev, lat, lon, f = fields(nt, nl, ny, nx)
    lon = -90.0 + 7.5 * np.arange(nx)          # an "area" request: -180..180 convention
    with h5py.File(path, "w", libver=("earliest", "v110")) as h:
        h.attrs["GRIB_centre"] = np.string_("ecmf")
        h.attrs["Conventions"] = np.string_("CF-1.7")
        h.attrs["institution"] = "European Centre for Medium-Range Weather Forecasts"
        d = h.create_dataset("number", data=np.int64(0))
        d.attrs["long_name"] = "ensemble member numerical id"
        d.attrs["units"] = "1"
        d.attrs["standard_name"] = "realization"
        vt = h.create_dataset("valid_time", data=(1577836800 + 3600 * np.arange(nt)).astype(np.int64))      # 2020-01-01 00:00 UTC, hourly
        vt.make_scale("valid_time")
        vt.attrs["long_name"] = "time"
        vt.attrs["standard_name"] = "time"
        vt.attrs["units"] = "seconds since 1970-01-01"
        vt.attrs["calendar"] = "proleptic_gregorian"
        pl = h.create_dataset("pressure_level", data=lev[:nl].astype(np.float64))        # 1000 ... 200 (5): descending
        pl.make_scale("pressure_level")
        pl.attrs["long_name"] = "pressure"
        pl.attrs["units"] = "hPa"
        pl.attrs["positive"] = "down"
        pl.attrs["stored_direction"] = "decreasing"
        pl.attrs["standard_name"] = "air_pressure"
        la = h.create_dataset("latitude", data=lat.astype(np.float64))
        la.make_scale("latitude")
        la.attrs["units"] = "degrees_north"
        la.attrs["stored_direction"] = "decreasing"
        lo = h.create_dataset("longitude", data=lon.astype(np.float64))
        lo.make_scale("longitude")
        lo.attrs["units"] = "degrees_east"
        ev = h.create_dataset("expver", data=np.array(["0001"] * (nt - 2) + ["0005"] * 2, dtype=object), dtype=h5py.string_dtype())
        ev.dims[0].attach_scale(vt)
        for name, a in f.items():
            a = a.astype(np.float32)
            if name == "v":
                a[1, 0, :, :] = np.nan
            if name == "w":
                a[3, 2, 4, 5] = np.nan              # an interior point: _handle_nans repairs the level
            d = h.create_dataset(name, data=a, chunks=(1, 1, ny, nx), shuffle=True, compression="gzip", compression_opts=1,
                                 fillvalue=np.float32(np.nan))
            d.attrs["_FillValue"] = np.float32(np.nan)
            d.attrs["GRIB_paramId"] = np.int64(130)
            d.attrs["GRIB_dataType"] = "an"
            d.attrs["GRIB_missingValue"] = np.float64(3.4028234663852886e+38)
            d.attrs["units"] = "K"
            d.attrs["long_name"] = "field " + name
            d.attrs["coordinates"] = "number expver"
            for i, dn in enumerate(("valid_time", "pressure_level", "latitude", "longitude")):
                d.dims[i].attach_scale(h[dn])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    write_cds_new(os.path.join(OUT, "cds_new_layout.nc"))
    write_sparse(os.path.join(OUT, "sparse_latest.h5"))
    write(os.path.join(OUT, "packed_chunked_earliest.nc"), "earliest", False, True, True, False, False)
    write(os.path.join(OUT, "packed_chunked_tracked.nc"), ("earliest", "v110"), True, True, True, True, True)
    write(os.path.join(OUT, "float_contiguous_latest.nc"), "latest", True, False, False, True, True)
    write(os.path.join(OUT, "float_chunked_latest.nc"), "latest", False, False, True, False, False)
    write(os.path.join(OUT, "packed_unlimited_v18.nc"), ("earliest", "v108"), True, True, True, False, False, unlimited=True)
    write(os.path.join(OUT, "packed_unlimited_latest.nc"), "latest", False, True, True, True, False, unlimited=True)    # extensible-array chunk index
    # chunks that span two time steps and tile the latitudes unevenly (5 + 5 + 3 rows): what a writer with its own chunk cache leaves
    write(os.path.join(OUT, "packed_timechunk2_latest.nc"), "latest", False, True, True, False, False, time_chunk=2)
    # chunked WITHOUT deflate: what every record variable of an uncompressed NetCDF-4 file is (the device path copies such chunks into place)
    write(os.path.join(OUT, "float_chunked_plain_latest.nc"), "latest", False, False, True, False, False, deflate=False, shuffle=False)
    write(os.path.join(OUT, "packed_shuffle_only_v18.nc"), ("earliest", "v108"), True, True, True, False, False, unlimited=True, deflate=False)
    write_interleaved(os.path.join(OUT, "packed_interleaved_v18.nc"))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
