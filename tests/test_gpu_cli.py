"""End-to-end drop-in runs (GPU): the reference's own smoke tests (tests/test_R2_fixed.py,
tests/test_R2_track.py of the reference) re-stated -- copy a preset namelist and box_limits / track
into inputs/, run the CLI with -r -f / -r -t -- plus what the reference never asserts: the numbers."""
import os
import shutil

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu

from lorenzcycletoolkit_amd.constants import LEVEL_TERMS
from oracle import lec_oracle as o


@pytest.fixture
def workdir(tmp_path, golden_dir, monkeypatch):
    os.makedirs(tmp_path / "inputs")
    shutil.copy(os.path.join(golden_dir, "inputs", "namelist_NCEP-R2"), tmp_path / "inputs" / "namelist")
    monkeypatch.chdir(tmp_path)
    return tmp_path


def _main(argv):
    import lorenzcycletoolkit
    parser = lorenzcycletoolkit.create_arg_parser()
    args = parser.parse_args(argv)
    lorenzcycletoolkit.main(argv)
    return args


def test_catarina_fixed_cli_matches_committed_outputs(workdir, golden_dir):
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    infile = os.path.join(golden_dir, "Catarina_NCEP-R2.nc")
    args = _main([infile, "-r", "-f", "-v"])
    assert args.fixed and args.residuals and args.verbosity and not args.track
    out = workdir / "LEC_Results" / "Catarina_NCEP-R2_fixed"
    got = pd.read_csv(out / "Catarina_NCEP-R2_fixed_results.csv", index_col=0)
    ref = pd.read_csv(os.path.join(golden_dir, "Catarina_NCEP-R2_fixed", "Catarina_NCEP-R2_fixed_results.csv"), index_col=0)
    assert list(got.columns) == list(ref.columns)                 # same schema, same order
    assert list(got.index) == list(ref.index)                     # same datetime formatting
    for c in ref.columns:
        a, r = got[c].values, ref[c].values
        assert np.all(np.abs(a - r) <= 5e-4 * np.abs(r) + 2e-4 * np.max(np.abs(r))), c   # float32 noise of the reference
    assert (out / "log.Catarina_NCEP-R2").exists()
    lvdir = out / "results_vertical_levels"
    assert sorted(os.listdir(lvdir)) == sorted(f"{t}_lv_ISBL3.csv" for t in LEVEL_TERMS)
    az = pd.read_csv(lvdir / "Az_lv_ISBL3.csv", index_col=0)
    refaz = pd.read_csv(os.path.join(golden_dir, "Catarina_NCEP-R2_fixed", "Az_lv_ISBL3.csv"), index_col=0)
    assert az.shape == refaz.shape and list(az.index) == list(refaz.index)
    assert [float(c) for c in az.columns] == [100 * float(c) for c in refaz.columns]   # Pa headers (CHANGELOG 1.0.0) vs hPa in the old sample
    assert np.max(np.abs(az.values - refaz.values)) / np.max(np.abs(refaz.values)) < 1e-4
    cz1 = pd.read_csv(lvdir / "Cz_1_lv_ISBL3.csv", index_col=0)
    assert list(cz1.index) == ["lv_ISBL3", "Cz_1"]                 # level-only term written transposed
    assert np.allclose(cz1.loc["Cz_1"].values * cz1.loc["lv_ISBL3"].values * o.G, o.RD)


def test_testdata_track_cli(workdir, golden_dir):
    shutil.copy(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), workdir / "inputs" / "track")
    infile = os.path.join(golden_dir, "testdata_NCEP-R2.nc")
    _main([infile, "-r", "-t"])
    out = workdir / "LEC_Results" / "testdata_NCEP-R2_track"
    got = pd.read_csv(out / "testdata_NCEP-R2_track_results.csv", index_col=0)
    want = ["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "BΦZ", "BΦE", "Gz", "Ge",
            "∂Az/∂t (finite diff.)", "∂Ae/∂t (finite diff.)", "∂Kz/∂t (finite diff.)", "∂Ke/∂t (finite diff.)",
            "RGz", "RKz", "RGe", "RKe"]
    assert list(got.columns) == want and len(got) == 5
    # numbers: oracle on the same float32 data upcast to fp64
    dom = o.load_ncep_sample(infile, dtype=np.float64)
    tr = pd.read_csv(workdir / "inputs" / "track", sep=";")
    domt = o.crop_domain_track(dom, tr.Lat.values, tr.Lon.values)
    limits = [(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(tr.Lat, tr.Lon)]
    ref, reflv = o.lec_moving(domt, limits)
    for c in want:
        r = np.asarray(ref[c], dtype=np.float64)
        assert np.max(np.abs(got[c].values - r)) <= 1e-9 * np.max(np.abs(r)), c
    # per-level tables: one row per time step; the committed Reg1 track sample pins Kz / Ke / Ce rows 1-3
    kz = pd.read_csv(out / "results_vertical_levels" / "Kz_lv_ISBL3.csv", index_col=0)
    assert kz.shape == (5, 5) and kz.index[1] == "2005-08-08 06:00:00"
    gold = pd.read_csv(os.path.join(golden_dir, "Reg1_track", "Kz_lv_ISBL3.csv"), index_col=0)
    cols = ["60000.0", "70000.0", "85000.0", "92500.0", "100000.0"]
    assert np.max(np.abs(kz.values[:3] / gold[cols].values - 1)) < 5e-7
    trk = pd.read_csv(out / "testdata_NCEP-R2_track_trackfile", sep=";")
    assert list(trk.columns) == ["time", "Lat", "Lon", "length", "width", "min_lon", "max_lon", "min_lat", "max_lat",
                                 "min_max_zeta_850_lat", "min_max_zeta_850_lon", "min_max_zeta_850",
                                 "min_hgt_850_lat", "min_hgt_850_lon", "min_hgt_850",
                                 "max_wind_850_lat", "max_wind_850_lon", "max_wind_850"]
    assert trk["time"][0] == "2005-08-08-0000" and trk["width"][0] == 15
    assert (trk["min_max_zeta_850"] < 0).all() and (trk["max_wind_850"] > 0).all()       # southern-hemisphere box
    assert trk["min_hgt_850_lat"].between(-30, -15).all() and trk["max_wind_850_lon"].between(-52.5, -37.5).all()


def test_track_with_width_and_length_columns(workdir, golden_dir):
    """A track file that carries its own box sizes (lec_moving_framework.py:224-225: width / length per time step; select_area.py:
    305-313: the pre-crop uses their maxima).  Numbers against the oracle with those boxes; -o is ignored by the moving framework
    (lec_moving_framework.py:525-528: the results file keeps its name)."""
    (workdir / "inputs" / "track").write_text(
        "time;Lat;Lon;length;width\n2005-08-08-0000;-22.5;-45;15;15\n2005-08-08-0600;-22.5;-45;10;12.5\n2005-08-08-1200;-25;-42.5;12.5;10\n"
        "2005-08-08-1800;-22.5;-45;10;10\n2005-08-09-0000;-20;-47.5;15;12.5\n")
    infile = os.path.join(golden_dir, "testdata_NCEP-R2.nc")
    _main([infile, "-r", "-t", "-o", "ignored_name"])
    out = workdir / "LEC_Results" / "testdata_NCEP-R2_track"
    assert not (out / "ignored_name.csv").exists()
    got = pd.read_csv(out / "testdata_NCEP-R2_track_results.csv", index_col=0)
    tr = pd.read_csv(workdir / "inputs" / "track", sep=";")
    dom = o.load_ncep_sample(infile, dtype=np.float64)
    domt = o.crop_domain_track(dom, tr.Lat.values, tr.Lon.values, max_width=tr.width.max(), max_length=tr.length.max())
    limits = [(lo - w / 2, lo + w / 2, la - l / 2, la + l / 2) for la, lo, l, w in zip(tr.Lat, tr.Lon, tr.length, tr.width)]
    ref, _ = o.lec_moving(domt, limits)
    for c in ("Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "BΦZ", "BΦE", "Gz", "Ge"):
        r = np.asarray(ref[c], dtype=np.float64)
        assert np.max(np.abs(got[c].values - r)) <= 1e-9 * np.max(np.abs(r)), c
    trk = pd.read_csv(out / "testdata_NCEP-R2_track_trackfile", sep=";")
    assert trk["length"].tolist() == [15, 10, 12.5, 10, 15] and trk["width"].tolist() == [15, 12.5, 10, 10, 12.5]
    assert np.allclose(trk["max_lon"] - trk["min_lon"], trk["width"]) and np.allclose(trk["max_lat"] - trk["min_lat"], trk["length"])


def test_custom_box_limits_file_meets_the_crop_by_inputs_box_limits(workdir, golden_dir):
    """SURVEY B-6: slice_domain always crops by the hard-coded inputs/box_limits (select_area.py:273-275) while lec_fixed takes its
    box from --box_limits (lec_fixed_framework.py:59-63): with a custom file the data are cropped first and the box is then the
    nearest grid points INSIDE that crop.  A custom box reaching beyond the crop therefore collapses onto it."""
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    (workdir / "inputs" / "wide").write_text("min_lon;-70\nmax_lon;-40\nmin_lat;-50\nmax_lat;-25\n")
    infile = os.path.join(golden_dir, "Catarina_NCEP-R2.nc")
    _main([infile, "-r", "-f", "--box_limits", "inputs/wide", "-o", "custom"])
    got = pd.read_csv(workdir / "LEC_Results" / "Catarina_NCEP-R2_fixed" / "custom.csv", index_col=0)
    dom = o.crop_domain(o.load_ncep_sample(infile, dtype=np.float64), -55, -36, -35, -20)
    ref, _ = o.lec_fixed(dom, -70, -40, -50, -25)           # nearest points of the CROPPED axes: west and south edges of the crop
    assert dom.lon[o.select_nearest(dom.lon, -70)] == dom.lon[0] and dom.lat[o.select_nearest(dom.lat, -50)] == dom.lat[0]
    for c in ("Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "Gz", "Ge"):
        r = np.asarray(ref[c], dtype=np.float64)
        assert np.max(np.abs(got[c].values - r)) <= 1e-9 * np.max(np.abs(r)), c


def test_non_residual_mode_fails_like_the_reference(workdir, golden_dir):
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    with pytest.raises(KeyError, match="Friction Velocity"):       # SURVEY B-8: only -r works in the reference
        _main([os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), "-f"])


def test_device_ingest_flag_writes_identical_csvs(workdir, golden_dir):
    """--device-ingest streams the file bytes to the GPU instead of preparing the data on the host: same CSVs, byte for byte."""
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    infile = os.path.join(golden_dir, "Catarina_NCEP-R2.nc")
    _main([infile, "-r", "-f", "-o", "host"])
    _main([infile, "-r", "-f", "--device-ingest", "-o", "device"])
    out = workdir / "LEC_Results" / "Catarina_NCEP-R2_fixed"
    assert (out / "host.csv").read_bytes() == (out / "device.csv").read_bytes()
    a = pd.read_csv(out / "device.csv", index_col=0)
    assert len(a) == 36 and np.isfinite(a.values).all()


@pytest.mark.parametrize("ingest", ["host", "device"])
@pytest.mark.parametrize("kind", ["fixed", "track"])
def test_reg1_sample_tables_pin_the_engine(workdir, golden_dir, kind, ingest):
    """The HIP path against the reference's OWN numbers on a second data set, both frameworks, both ways of preparing the data:
    the level tables the CLI writes for testdata_NCEP-R2.nc against every cell of the committed Reg1 sample tables that the 5-level /
    5-step subset reproduces (tests/helpers.py REG1_TERMS) -- sigma, the Q stencil (the moving framework's with the dT/dt it forms
    over the track's times), Ca's two gradients, Ck.  The reference computed in float32, the engine computes in fp64 from the same
    float32 file: tolerance policy (ii) of SURVEY appendix D.  The track sample's Ck predates the current second piece
    (tests/golden/README.md; accounted for on the CPU in tests/test_oracle_golden.py) and is left out."""
    from tests.helpers import REG1_TERMS, reg1_table
    shutil.copy(os.path.join(golden_dir, "inputs", "box_limits_Reg1"), workdir / "inputs" / "box_limits")
    shutil.copy(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), workdir / "inputs" / "track")
    infile = os.path.join(golden_dir, "testdata_NCEP-R2.nc")
    _main([infile, "-r", "-f" if kind == "fixed" else "-t", "--ingest", ingest])
    lvdir = workdir / "LEC_Results" / f"testdata_NCEP-R2_{kind}" / "results_vertical_levels"
    worst = {}
    for term in REG1_TERMS:
        if kind == "track" and term == "Ck":
            continue
        r, lev, rows, sign = reg1_table(golden_dir, kind, term)
        got = pd.read_csv(lvdir / f"{term}_lv_ISBL3.csv", index_col=0)
        assert [float(c) for c in got.columns] == [60000.0, 70000.0, 85000.0, 92500.0, 100000.0]
        a = sign * got.values[:rows, lev]
        assert np.all(np.abs(a - r) <= 2e-4 * np.abs(r) + 1e-4 * np.max(np.abs(r))), term
        worst[term] = float(np.max(np.abs(a - r)) / np.max(np.abs(r)))
    print(kind, ingest, worst)


@pytest.mark.parametrize("ingest", ["host", "device"])
def test_reg1_choose_sample_tables_pin_the_moving_engine(workdir, golden_dir, ingest):
    """The reference's `-c` sample (samples/Reg1-Representative_NCEP-R2_choose: the moving framework with a box picked per step) has
    three steps with three DIFFERENT boxes; its boxes were recovered from its own Kz table (tests/helpers.REG1_CHOOSE_BOXES).  The same
    boxes as a track with width / length columns (two more steps appended so that the third step's dT/dt is centred, as it was over the
    chooser's whole file): the engine's tables -- host-prepared and box-packed device ingest -- against the reference's own numbers,
    float32-noise policy of SURVEY appendix D; Ck left out (the sample predates today's second piece: tests/test_oracle_golden.py)."""
    from tests.helpers import REG1_CHOOSE_BOXES, REG1_TERMS, reg1_table
    boxes = REG1_CHOOSE_BOXES + [REG1_CHOOSE_BOXES[-1]] * 2
    stamps = ["2005-08-08-0000", "2005-08-08-0600", "2005-08-08-1200", "2005-08-08-1800", "2005-08-09-0000"]
    (workdir / "inputs" / "track").write_text("time;Lat;Lon;length;width\n" + "".join(
        f"{ts};{(s + n) / 2};{(w + e) / 2};{n - s};{e - w}\n" for ts, (w, e, s, n) in zip(stamps, boxes)))
    infile = os.path.join(golden_dir, "testdata_NCEP-R2.nc")
    _main([infile, "-r", "-t", "--ingest", ingest])
    lvdir = workdir / "LEC_Results" / "testdata_NCEP-R2_track" / "results_vertical_levels"
    trk = pd.read_csv(workdir / "LEC_Results" / "testdata_NCEP-R2_track" / "testdata_NCEP-R2_track_trackfile", sep=";")
    assert [tuple(x) for x in trk[["min_lon", "max_lon", "min_lat", "max_lat"]].values[:3]] == [tuple(b) for b in REG1_CHOOSE_BOXES]
    for term in REG1_TERMS:
        if term == "Ck":
            continue
        r, lev, rows, sign = reg1_table(golden_dir, "choose", term)
        got = pd.read_csv(lvdir / f"{term}_lv_ISBL3.csv", index_col=0)
        a = sign * got.values[:rows, lev]
        assert np.all(np.abs(a - r) <= 2e-4 * np.abs(r) + 1e-4 * np.max(np.abs(r))), term


def test_ingest_auto_falls_back_to_the_host_preparation(workdir, monkeypatch):
    """`--ingest auto` (the default) sends a deflated NetCDF-4 file to the streamed device path by itself.  If that path then refuses
    the input (ValueError / NotImplementedError), the run must not fail where the host preparation works: the file is closed, the
    reason logged, the data prepared on the host -- same files as `--ingest host`.  Asked for by name, the refusal stands."""
    from lorenzcycletoolkit_amd import ingest
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(ROOT, "tests", "golden", "hdf5", "cds_new_layout.nc")
    shutil.copy(os.path.join(ROOT, "inputs", "namelist_ERA5-copernicus-new"), workdir / "inputs" / "namelist")
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;30\nmin_lat;-40\nmax_lat;30\n")
    out = workdir / "LEC_Results" / "cds_new_layout_fixed"
    tree = lambda: {os.path.relpath(os.path.join(d, f), out): open(os.path.join(d, f), "rb").read()
                    for d, _, fs in os.walk(out) for f in fs if not f.startswith("log.")}
    _main([src, "-r", "-f", "--ingest", "host"])
    host = tree()
    shutil.rmtree(out)
    closed = []
    real_close = ingest.ds.RawDataset.close
    monkeypatch.setattr(ingest.ds.RawDataset, "close", lambda self: (closed.append(1), real_close(self))[1])

    real_streamed = ingest.lec_streamed

    def refuse(*a, **k):
        raise ValueError("simulated refusal of the streamed path")
    monkeypatch.setattr(ingest, "lec_streamed", refuse)
    _main([src, "-r", "-f"])                                       # auto: device chosen, refused in mid-run, host preparation takes over
    assert tree() == host and closed
    log = open(out / "log.cds_new_layout").read()
    assert "whose chunks the GPU can inflate" in log and "simulated refusal" in log and "preparing the data on the host instead" in log
    with pytest.raises(ValueError, match="simulated refusal"):
        _main([src, "-r", "-f", "--ingest", "device"])
    # ... but only a REFUSAL of the streamed path falls back: an error later in the run (here: while the tables are written, after the
    # engine has returned) is an error -- no second analysis on the host
    monkeypatch.setattr(ingest, "lec_streamed", real_streamed)
    import lorenzcycletoolkit as cli
    from lorenzcycletoolkit_amd import frameworks
    host_runs = []
    real_prepare = cli.prepare_data
    monkeypatch.setattr(cli, "prepare_data", lambda *a, **k: (host_runs.append(1), real_prepare(*a, **k))[1])

    def broken(*a, **k):
        raise ValueError("simulated failure while writing the tables")
    monkeypatch.setattr(frameworks, "format_level_table", broken)
    shutil.rmtree(out)
    with pytest.raises(ValueError, match="while writing the tables"):
        _main([src, "-r", "-f"])
    assert not host_runs


def _write_era5_style(path, nt=6):
    """What the reference's missing samples/testdata_ERA5.nc looks like to the toolkit: namelist_ERA5 names (T, Z, W, U, V; time,
    level, latitude, longitude), hourly from 2005-08-09, int16-packed, latitudes N -> S, levels in millibars incl. 850 hPa."""
    from scipy.io import netcdf_file
    rng = np.random.default_rng(11)
    lon = np.arange(-80.0, -9.0, 1.25)
    lat = np.arange(0.0, -61.25, -1.25)
    lev = np.array([1000, 925, 850, 700, 500, 300, 200, 100], dtype=np.int32)
    nl, ny, nx = lev.size, lat.size, lon.size
    f = netcdf_file(path, "w", version=2)
    for n, s in (("time", nt), ("level", nl), ("latitude", ny), ("longitude", nx)):
        f.createDimension(n, s)
    tv = f.createVariable("time", "i", ("time",)); tv[:] = 925704 + np.arange(nt); tv.units = "hours since 1900-01-01 00:00:00.0"
    lv = f.createVariable("level", "i", ("level",)); lv[:] = lev; lv.units = "millibars"
    la = f.createVariable("latitude", "f", ("latitude",)); la[:] = lat
    lo = f.createVariable("longitude", "f", ("longitude",)); lo[:] = lon
    p = (lev[None, :, None, None] * 100.0) / 1e5
    tt = np.arange(nt)[:, None, None, None]
    lam, phi = np.deg2rad(lon)[None, None, None, :], np.deg2rad(lat)[None, None, :, None]
    fields = {
        "T": 288.0 * p ** 0.19 + 8.0 * np.cos(2 * phi) * p + 2.0 * np.sin(3 * lam + 0.1 * tt) + 0.3 * rng.standard_normal((nt, nl, ny, nx)),
        "U": 20.0 * np.cos(phi) * (1 - p / 1.2) + 3 * rng.standard_normal((nt, nl, ny, nx)),
        "V": 3.0 * np.sin(2 * lam) + 2 * rng.standard_normal((nt, nl, ny, nx)),
        "W": 0.1 * rng.standard_normal((nt, nl, ny, nx)),
        "Z": 9.80665 * 7000.0 * np.log(1.0 / p) + 50.0 * rng.standard_normal((nt, nl, ny, nx)),
    }
    for name, a in fields.items():
        lo_, hi_ = a.min(), a.max()
        scale, offset = (hi_ - lo_) / 65000.0, 0.5 * (hi_ + lo_)
        v = f.createVariable(name, "h", ("time", "level", "latitude", "longitude"))
        v[:] = np.clip(np.round((a - offset) / scale), -32000, 32000).astype(np.int16)
        v.scale_factor = float(scale); v.add_offset = float(offset); v._FillValue = np.int16(-32767); v.missing_value = np.int16(-32767)
    f.close()


def test_era5_style_fixed_and_track_cli(workdir, golden_dir):
    """The reference's tests/test_ERA5_fixed.py and tests/test_ERA5_track.py re-stated (same namelist, box and track inputs, flags
    -r -f -p -v and -r -t -p -v); numbers against the oracle on the host-decoded data."""
    import lorenzcycletoolkit
    shutil.copy(os.path.join(golden_dir, "inputs", "namelist_ERA5"), workdir / "inputs" / "namelist")
    shutil.copy(os.path.join(golden_dir, "inputs", "box_limits_Reg1"), workdir / "inputs" / "box_limits")
    shutil.copy(os.path.join(golden_dir, "inputs", "track_testdata_ERA5"), workdir / "inputs" / "track")
    _write_era5_style("testdata_ERA5.nc")
    # fixed, host path and device ingest
    args = _main(["testdata_ERA5.nc", "-r", "-f", "-p", "-v"])
    assert args.residuals and args.fixed and args.plots and args.verbosity
    out = workdir / "LEC_Results" / "testdata_ERA5_fixed"
    got = pd.read_csv(out / "testdata_ERA5_fixed_results.csv", index_col=0)
    assert len(got) == 6 and np.isfinite(got.values).all() and str(got.index[0]) == "2005-08-09 00:00:00"
    # the oracle's OWN preparation of the file (oracle/cf_decode.py: xarray 2024.2.0's decode, process_data, slice_domain), not the
    # package's: int16 + add_offset + _FillValue decode to float32 there; the engine computes in fp64 from those float32 values
    from oracle import cf_decode as cf
    from tests.helpers import as_f64
    names = {"tair": "T", "u": "U", "v": "V", "omega": "W", "geo": "Z", "lat": "latitude", "lon": "longitude", "level": "level", "time": "time"}
    prepared = cf.prepare("testdata_ERA5.nc", names, fixed_limits=(-60.0, -30.0, -42.5, -17.5))
    assert prepared.tair.dtype == np.float32
    dom = as_f64(prepared)
    ref, _ = o.lec_fixed(dom, -60.0, -30.0, -42.5, -17.5)
    for c in ("Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "Gz", "Ge"):
        r = np.asarray(ref[c], dtype=np.float64)
        assert np.max(np.abs(got[c].values - r)) <= 1e-9 * np.max(np.abs(r)), c
    _main(["testdata_ERA5.nc", "-r", "-f", "--device-ingest", "-o", "device"])
    assert (out / "device.csv").read_bytes() == (out / "testdata_ERA5_fixed_results.csv").read_bytes()
    # track
    args = _main(["testdata_ERA5.nc", "-r", "-t", "-p", "-v"])
    assert args.track and args.plots
    tr = pd.read_csv(workdir / "LEC_Results" / "testdata_ERA5_track" / "testdata_ERA5_track_results.csv", index_col=0)
    assert len(tr) == 5 and np.isfinite(tr[["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "Gz", "Ge"]].values).all()
    tdir = workdir / "LEC_Results" / "testdata_ERA5_track"
    assert (tdir / "testdata_ERA5_track_trackfile").exists()
    # numbers: the oracle's moving framework on the oracle's preparation of the file (track-time selection, track-extent crop)
    trk = pd.read_csv(workdir / "inputs" / "track", sep=";")
    times = pd.to_datetime(trk["time"], format="%Y-%m-%d-%H%M").values.astype("datetime64[ns]")
    domt = as_f64(cf.prepare("testdata_ERA5.nc", names, track=(times, trk["Lat"].values, trk["Lon"].values)))
    limits = [(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(trk["Lat"], trk["Lon"])]
    mref, _ = o.lec_moving(domt, limits)
    for c in ("Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "BΦZ", "BΦE", "Gz", "Ge"):
        r = np.asarray(mref[c], dtype=np.float64)
        assert np.max(np.abs(tr[c].values - r)) <= 1e-9 * np.max(np.abs(r)), c
    # the same run with the data streamed to the GPU (per-step boxes over chunked ingest): byte-identical outputs
    before = {f: (tdir / f).read_bytes() for f in ("testdata_ERA5_track_results.csv", "testdata_ERA5_track_trackfile")}
    _main(["testdata_ERA5.nc", "-r", "-t", "--device-ingest"])
    for f, b in before.items():
        assert (tdir / f).read_bytes() == b, f


def test_outputs_feed_the_reference_plot_readers(workdir, golden_dir):
    """the reference's plot scripts consume only the CSVs, through four readers (src/plots/utils.py:79-193 of the
    reference: read_results, read_track, read_box_limits, get_data_vertical_levels).  The readers are restated here as the pandas
    calls they make and run over the -f and -t outputs and the *_trackfile: index types, column sets and headers are what the plot
    code indexes by."""
    import re
    from glob import glob

    def read_results(path):                                   # plots/utils.py:79-94
        df = pd.read_csv(path, index_col=[0])
        df["Datetime"] = pd.to_datetime(df.index)
        return df.set_index("Datetime")

    def read_track(path):                                     # plots/utils.py:110-121
        return pd.read_csv(path, parse_dates=[0], delimiter=";", index_col="time")

    def read_box_limits(path):                                # plots/utils.py:138-153
        df = pd.read_csv(path, sep=";", index_col=0, header=None)
        return df.loc[["min_lon", "max_lon", "min_lat", "max_lat"]]

    def get_data_vertical_levels(results_subdirectory):      # plots/utils.py:156-193: split terms (Cz_1, Ck_3, ...) are skipped
        files = [f for f in glob(os.path.join(results_subdirectory, "results_vertical_levels", "*.csv"))
                 if not re.search(r"_[0-9]", os.path.basename(f))]
        data = {}
        for f in files:
            term = os.path.splitext(os.path.basename(f))[0].split("_")[0]
            data[term] = pd.read_csv(f, header=0, index_col=0, parse_dates=True)
            data[term].columns = [float(x) for x in data[term].columns]
        return data

    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    _main([os.path.join(golden_dir, "Catarina_NCEP-R2.nc"), "-r", "-f"])
    shutil.copy(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), workdir / "inputs" / "track")
    _main([os.path.join(golden_dir, "testdata_NCEP-R2.nc"), "-r", "-t"])
    fixed, track = workdir / "LEC_Results" / "Catarina_NCEP-R2_fixed", workdir / "LEC_Results" / "testdata_NCEP-R2_track"

    box = read_box_limits(workdir / "inputs" / "box_limits")
    assert box.loc["min_lon"].iloc[0] == -55 and box.loc["max_lat"].iloc[0] == -20

    ten = {"Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "Gz", "Ge"}
    for out, name, nt, nl, extra in ((fixed, "Catarina_NCEP-R2_fixed_results.csv", 36, 17, set()),
                                      (track, "testdata_NCEP-R2_track_results.csv", 5, 5, {"BΦZ", "BΦE"})):
        df = read_results(out / name)
        assert isinstance(df.index, pd.DatetimeIndex) and df.index.is_monotonic_increasing and len(df) == nt
        # what timeseries / LEC-diagram / LPS plots index by (plots/plot_timeseries.py, plot_LEC.py, plot_LPS.py of the reference)
        need = ten | {"BAz", "BAe", "BKz", "BKe", "RGz", "RKz", "RGe", "RKe"} | {f"∂{t}/∂t (finite diff.)" for t in ("Az", "Ae", "Kz", "Ke")} | extra
        assert need <= set(df.columns) and all(df[c].dtype == np.float64 for c in df.columns)
        lv = get_data_vertical_levels(str(out))
        assert set(lv) == ten                               # one table per un-split term
        for term, tab in lv.items():
            assert tab.shape == (nt, nl) and tab.columns[0] < tab.columns[-1] and tab.columns[-1] == 100000.0, term      # Pa, top -> bottom
            assert isinstance(tab.index, pd.DatetimeIndex) and list(tab.index) == list(df.index), term                # Hovmoller x-axis
            assert np.isfinite(tab.values).all(), term
        # integrating a level table over pressure reproduces the results column (what plot_hovmoller's users cross-check)
        p = np.array(lv["Kz"].columns)
        kz = (getattr(np, "trapezoid", None) or np.trapz)(lv["Kz"].values, p, axis=1) / (2 * o.G)
        assert np.allclose(kz, df["Kz"].values, rtol=1e-12)

    trk = read_track(track / "testdata_NCEP-R2_track_trackfile")
    assert trk.index.name == "time" and len(trk) == 5
    for col in ("Lat", "Lon", "length", "width", "min_lon", "max_lon", "min_lat", "max_lat", "min_max_zeta_850", "min_hgt_850", "max_wind_850"):
        assert col in trk.columns and np.issubdtype(trk[col].dtype, np.number), col      # plot_track / plot_box_limits read these
