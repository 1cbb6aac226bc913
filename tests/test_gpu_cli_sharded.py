"""The time shard behind the drop-in command line (SURVEY 8e / VERDICT r2 row e'): ``lorenzcycletoolkit.py ... --gpus N`` and
``python -m torch.distributed.run --nproc-per-node N lorenzcycletoolkit.py ...`` shard the time steps over N rank processes, gather
the per-step results on rank 0 and must write the SAME BYTES as the one-process run -- every results CSV, every per-level table and
the trackfile.  On this box the ranks share the one GPU and rendezvous over gloo (LEC_DIST_BACKEND=gloo); RCCL needs a GPU per rank
(the nccl variant below runs where there are at least two)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from tests.helpers import write_packed_era5_style

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "lorenzcycletoolkit.py")


@pytest.fixture
def workdir(tmp_path, golden_dir, monkeypatch):
    os.makedirs(tmp_path / "inputs")
    shutil.copy(os.path.join(golden_dir, "inputs", "namelist_NCEP-R2"), tmp_path / "inputs" / "namelist")
    monkeypatch.chdir(tmp_path)
    return tmp_path


def _run(argv, backend="gloo", launcher=None, timeout=600, extra_env=None):
    env = dict(os.environ, LEC_DIST_BACKEND=backend)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LEC_FORCE_SHARD"):
        env.pop(k, None)
    env.update(extra_env or {})
    cmd = [sys.executable, CLI] + argv if launcher is None else launcher + [CLI] + argv
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r


def _tree(d):
    """{relative path: bytes} of every result file (the log carries time stamps and the command line)."""
    out = {}
    for base, _, files in os.walk(d):
        for f in files:
            if not f.startswith("log."):
                p = os.path.join(base, f)
                out[os.path.relpath(p, d)] = open(p, "rb").read()
    return out


def _same_tree(one, many, what):
    assert sorted(one) == sorted(many), what
    for k in one:
        assert one[k] == many[k], f"{what}: {k} differs from the one-process run"


def _fresh(results):
    if os.path.isdir(results):
        shutil.rmtree(results)


@pytest.mark.parametrize("ingest", [[], ["--device-ingest"]])
def test_fixed_framework_sharded_writes_identical_files(workdir, golden_dir, ingest):
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    infile = os.path.join(golden_dir, "Catarina_NCEP-R2.nc")
    results = workdir / "LEC_Results" / "Catarina_NCEP-R2_fixed"
    _run([infile, "-r", "-f"] + ingest)
    one = _tree(results)
    assert len(one) == 22 and "Catarina_NCEP-R2_fixed_results.csv" in one        # results + 21 per-level tables
    for n in (2, 3):
        _fresh(results)
        r = _run([infile, "-r", "-f", "--gpus", str(n)] + ingest)
        _same_tree(one, _tree(results), f"-f {ingest} on {n} ranks")
        assert f"Time-sharded run: {n} ranks" in open(results / "log.Catarina_NCEP-R2").read()


@pytest.mark.parametrize("ingest", [[], ["--device-ingest"]])
def test_moving_framework_sharded_writes_identical_files(workdir, golden_dir, ingest):
    shutil.copy(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), workdir / "inputs" / "track")
    infile = os.path.join(golden_dir, "testdata_NCEP-R2.nc")
    results = workdir / "LEC_Results" / "testdata_NCEP-R2_track"
    _run([infile, "-r", "-t"] + ingest)
    one = _tree(results)
    assert "testdata_NCEP-R2_track_results.csv" in one and "testdata_NCEP-R2_track_trackfile" in one
    for n in (2, 3):
        _fresh(results)
        _run([infile, "-r", "-t", "--gpus", str(n)] + ingest)
        _same_tree(one, _tree(results), f"-t {ingest} on {n} ranks")


@pytest.mark.parametrize("ingest", [[], ["--device-ingest"]])
def test_nan_levels_are_dropped_across_shards(workdir, ingest):
    """An int16-packed file whose fill values make the TOP kept level NaN at ONE time step (step 2) and one interior point NaN at
    another: the reference's dropna(dim=level) on the [time, level] arrays drops that level for EVERY time step, so the rank that
    holds step 2 must tell the others (the [28, L] mask all_reduce); bytes equal to the one-process run, on 2 and 3 ranks."""
    path = str(workdir / "packed.nc")
    write_packed_era5_style(path)
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
        "Time;time\nVertical Level;level\n")
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;20\nmin_lat;-45\nmax_lat;30\n")
    results = workdir / "LEC_Results" / "packed_fixed"
    _run([path, "-r", "-f"] + ingest)
    one = _tree(results)
    df = pd.read_csv(results / "packed_fixed_results.csv", index_col=0)
    assert len(df) == 7 and np.isfinite(df[["Az", "Ae", "Kz", "Ke"]].values).all()
    assert "NaN level values" in open(results / "log.packed").read()
    for n in (2, 3):
        _fresh(results)
        _run([path, "-r", "-f", "--gpus", str(n)] + ingest)
        _same_tree(one, _tree(results), f"NaN sample {ingest} on {n} ranks")


def test_under_the_torch_launcher(workdir, golden_dir):
    """python -m torch.distributed.run --nproc-per-node 2 lorenzcycletoolkit.py ...: the launcher's environment names the ranks."""
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    infile = os.path.join(golden_dir, "Catarina_NCEP-R2.nc")
    results = workdir / "LEC_Results" / "Catarina_NCEP-R2_fixed"
    _run([infile, "-r", "-f", "-o", "one"])
    one = open(results / "one.csv", "rb").read()
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    _run([infile, "-r", "-f", "-o", "two"], launcher=launcher)
    assert open(results / "two.csv", "rb").read() == one


def test_more_ranks_than_time_steps_is_refused(workdir, golden_dir):
    (workdir / "inputs" / "track").write_text("time;Lat;Lon\n2005-08-08-0000;-22.5;-45\n2005-08-08-0600;-22.5;-45\n")      # two time steps
    env = dict(os.environ, LEC_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, CLI, os.path.join(golden_dir, "testdata_NCEP-R2.nc"), "-r", "-t", "--gpus", "3"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "cannot be sharded over 3 ranks" in r.stderr


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: runs on a node with >= 2 GPUs")
def test_sharded_over_rccl(workdir, golden_dir):
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    infile = os.path.join(golden_dir, "Catarina_NCEP-R2.nc")
    results = workdir / "LEC_Results" / "Catarina_NCEP-R2_fixed"
    _run([infile, "-r", "-f"])
    one = _tree(results)
    _fresh(results)
    _run([infile, "-r", "-f", "--gpus", str(min(torch.cuda.device_count(), 4))], backend="nccl")
    _same_tree(one, _tree(results), "RCCL")


@pytest.mark.parametrize("ingest", [[], ["--device-ingest"]])
def test_the_rccl_branch_of_the_cli_with_one_rank(workdir, golden_dir, ingest):
    """LEC_FORCE_SHARD=1: the command line builds the world-1 shard context over backend "nccl" (= RCCL) -- process group,
    ``dist.gather`` into the unbound receive views, the mask all_reduce on DEVICE tensors, the nine-number diagnostics gather,
    ``barrier(device_ids)`` -- i.e. every line of the N > 1 branch that a one-GPU box can execute, for -f and -t, and must write
    the bytes of the plain run (more RCCL ranks need more GPUs: test_sharded_over_rccl)."""
    (workdir / "inputs" / "box_limits").write_text("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
    shutil.copy(os.path.join(golden_dir, "inputs", "track_testdata_NCEP-R2"), workdir / "inputs" / "track")
    for flag, infile, name in (("-f", "Catarina_NCEP-R2.nc", "Catarina_NCEP-R2_fixed"), ("-t", "testdata_NCEP-R2.nc", "testdata_NCEP-R2_track")):
        infile = os.path.join(golden_dir, infile)
        results = workdir / "LEC_Results" / name
        _fresh(results)
        _run([infile, "-r", flag] + ingest)
        one = _tree(results)
        _fresh(results)
        _run([infile, "-r", flag] + ingest, backend="nccl", extra_env={"LEC_FORCE_SHARD": "1"})
        _same_tree(one, _tree(results), f"{flag} {ingest} through the world-1 RCCL shard context")
        log = [f for f in os.listdir(results) if f.startswith("log.")][0]
        assert "Time-sharded run: 1 ranks (backend nccl)" in open(results / log).read()


def test_nan_mask_all_reduce_on_device_tensors_with_one_rccl_rank(workdir):
    """The NaN sample (a level that is NaN at one step only) through the world-1 RCCL context: ``merge_dropmask`` all_reduces the
    [28, L] int32 mask where it lies -- in device memory -- and the files equal the plain run's."""
    path = str(workdir / "packed.nc")
    write_packed_era5_style(path)
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
        "Time;time\nVertical Level;level\n")
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;20\nmin_lat;-45\nmax_lat;30\n")
    results = workdir / "LEC_Results" / "packed_fixed"
    _run([path, "-r", "-f"])
    one = _tree(results)
    _fresh(results)
    _run([path, "-r", "-f"], backend="nccl", extra_env={"LEC_FORCE_SHARD": "1"})
    _same_tree(one, _tree(results), "NaN sample through the world-1 RCCL shard context")


def test_deflated_netcdf4_file_through_the_cli(workdir):
    """A shuffled + deflated NetCDF-4 file (tests/golden/hdf5/packed_chunked_tracked.nc, int16 with fill values) through the command
    line: host-prepared resident run, ``--device-ingest`` (the chunks inflate on the GPU: lec_inflate) and the same on two ranks, where
    every rank inflates only its own steps' chunks -- three times the same bytes, for the fixed box and for a track."""
    src = os.path.join(ROOT, "tests", "golden", "hdf5", "packed_chunked_tracked.nc")
    (workdir / "inputs" / "namelist").write_text(
        ";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\n"
        "Eastward Wind Component;u;m/s\nNorthward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\n"
        "Time;time\nVertical Level;level\n")
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;30\nmin_lat;-40\nmax_lat;30\n")
    (workdir / "inputs" / "track").write_text(
        "time;Lat;Lon;width;length\n" + "".join(f"2020-01-01-{6 * t:02d}00;{-10 + 5 * t};{-30 + 10 * t};60;50\n" for t in range(4)))
    for flag, name in (("-f", "fixed"), ("-t", "track")):
        results = workdir / "LEC_Results" / f"packed_chunked_tracked_{name}"
        _fresh(results)
        _run([src, "-r", flag, "--ingest", "host"])
        one = _tree(results)
        assert "being inflated on the host" in open(results / "log.packed_chunked_tracked").read()
        df = pd.read_csv(results / f"packed_chunked_tracked_{name}_results.csv", index_col=0)
        assert len(df) == (5 if name == "fixed" else 4) and np.isfinite(df[["Az", "Ae", "Kz", "Ke"]].values).all()
        _fresh(results)
        _run([src, "-r", flag, "--device-ingest"])
        _same_tree(one, _tree(results), f"{name}: --device-ingest (device inflate)")
        assert "inflate device" in open(results / "log.packed_chunked_tracked").read()
        _fresh(results)
        _run([src, "-r", flag, "--device-ingest", "--inflate", "host"])
        _same_tree(one, _tree(results), f"{name}: --device-ingest --inflate host")
        assert "inflate host" in open(results / "log.packed_chunked_tracked").read()
        _fresh(results)
        _run([src, "-r", flag, "--device-ingest", "--gpus", "2"])
        _same_tree(one, _tree(results), f"{name}: --device-ingest on 2 ranks")
        _fresh(results)
        _run([src, "-r", flag])                                    # --ingest auto (the default): a deflated file goes to the device by itself
        _same_tree(one, _tree(results), f"{name}: default ingest")
        log = open(results / "log.packed_chunked_tracked").read()
        assert "whose chunks the GPU can inflate" in log and "inflate device" in log


def test_cds_new_layout_through_the_cli(workdir):
    """The layout the Copernicus CDS delivers today (tests/golden/hdf5/cds_new_layout.nc: `valid_time` int64 seconds since 1970,
    `pressure_level` float64 hPa descending, scalar `number`, per-time string `expver`, float32 + shuffle + deflate, NaN _FillValue) with
    the reference's own preset inputs/namelist_ERA5-copernicus-new: host-prepared resident run, `--device-ingest` (chunks inflated on
    the GPU) and the same on two ranks write the same bytes, for the fixed box and for a track; the fixed-box numbers agree with the
    oracle on the oracle's preparation of the arrays the fixture's writer wrote (not of what the package's reader read)."""
    src = os.path.join(ROOT, "tests", "golden", "hdf5", "cds_new_layout.nc")
    shutil.copy(os.path.join(ROOT, "inputs", "namelist_ERA5-copernicus-new"), workdir / "inputs" / "namelist")
    (workdir / "inputs" / "box_limits").write_text("min_lon;-60\nmax_lon;30\nmin_lat;-40\nmax_lat;30\n")
    (workdir / "inputs" / "track").write_text(
        "time;Lat;Lon;width;length\n" + "".join(f"2020-01-01-{t:02d}00;{-10 + 5 * t};{-30 + 10 * t};60;50\n" for t in range(5)))
    for flag, name in (("-f", "fixed"), ("-t", "track")):
        results = workdir / "LEC_Results" / f"cds_new_layout_{name}"
        _fresh(results)
        _run([src, "-r", flag, "--ingest", "host"])
        one = _tree(results)
        df = pd.read_csv(results / f"cds_new_layout_{name}_results.csv", index_col=0)
        assert len(df) == (6 if name == "fixed" else 5) and np.isfinite(df[["Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce"]].values).all()
        assert str(df.index[1]) == "2020-01-01 01:00:00"
        if name == "fixed":
            assert "NaN level values" in open(results / "log.cds_new_layout").read()      # v's 1000-hPa level is NaN at step 1: dropped for every step
            fixed = df
        _fresh(results)
        _run([src, "-r", flag, "--device-ingest"])
        _same_tree(one, _tree(results), f"{name}: --device-ingest")
        assert "inflate device" in open(results / "log.cds_new_layout").read()
        _fresh(results)
        _run([src, "-r", flag, "--device-ingest", "--gpus", "2"])
        _same_tree(one, _tree(results), f"{name}: --device-ingest on 2 ranks")
        _fresh(results)
        _run([src, "-r", flag, "--gpus", "2", "-m", "--ingest", "host"])      # --mpas: nothing on `standard_height` here, same files
        _same_tree(one, _tree(results), f"{name}: resident on 2 ranks with --mpas")
        _fresh(results)
        _run([src, "-r", flag])                                    # the default: --ingest auto takes the device for this file
        _same_tree(one, _tree(results), f"{name}: default ingest")
        assert "inflate device" in open(results / "log.cds_new_layout").read()
    from oracle import cf_decode as cf
    from oracle import lec_oracle as o
    from tests.helpers import as_f64
    from tests.test_hdf5_cpu import _cds_expected
    lev, lat, lon, f = _cds_expected()
    var = dict(f, valid_time=cf.decode_cf_time(1577836800 + 3600 * np.arange(6, dtype=np.int64), "seconds since 1970-01-01"),
               pressure_level=lev[:6].astype(np.float64), latitude=lat, longitude=lon)
    dims = {k: ("valid_time", "pressure_level", "latitude", "longitude") for k in f}
    names = {"tair": "t", "u": "u", "v": "v", "omega": "w", "geo": "z", "lat": "latitude", "lon": "longitude", "level": "pressure_level",
             "time": "valid_time"}
    dom = as_f64(cf.prepare_opened((var, dims, {"pressure_level": {"units": "hPa"}}), names, fixed_limits=(-60.0, 30.0, -40.0, 30.0)))
    ref, _ = o.lec_fixed(dom, -60.0, 30.0, -40.0, 30.0)
    for c in ("Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "Gz", "Ge"):
        r = np.asarray(ref[c], dtype=np.float64)
        assert np.array_equal(np.isnan(fixed[c].values), np.isnan(r)) and np.nanmax(np.abs(fixed[c].values - r)) <= 1e-9 * np.nanmax(np.abs(r)), c
