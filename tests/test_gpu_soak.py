"""Bounded slices of the randomised soaks, with fixed seeds, under the driver's `pytest -m gpu`.

The soak scripts (tests/soak_gpu.py, soak_streamed.py, soak_diag.py, soak_repeat.py) draw random grids, boxes, storage types, axis
orders, NaN patches and chunk lengths and found seven real defects in round 3 (each pinned by a regression test since); run by hand they
take hundreds of cases.  Here ~50 cases of each run in seconds, so the randomised coverage is part of the recorded GPU test tier and
not a claim in the notes.  The seeds differ from the ones the by-hand runs use (1, 2): different cases, same generators."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_soak_slice_resident_kernels_oracle_and_shards():
    """50 random grids / boxes / dtypes / time axes / NaN patches: every kernel family that can run a case agrees record by record,
    the terms agree with the oracle (every other case), a shard of the series reproduces the whole bit for bit."""
    from tests import soak_gpu as soak
    rng = np.random.default_rng(20260401)
    fails = []
    for c in range(50):
        fails += soak.one_case(rng, c, with_oracle=(c % 2 == 0))
    assert not fails, fails[:5]


@pytest.mark.parametrize("fill_rate", [0.0, 0.002])
def test_soak_slice_streamed_pipeline(tmp_path, monkeypatch, fill_rate):
    """24 random classic NetCDF files (+ 12 with scattered fill values, where nearly every level is NaN somewhere): the resident framework
    run, `lec_streamed` with a random chunk length (staged / registered), two time ranges with the NaN-level mask merged by hand, and
    the oracle on ITS preparation of the file; every third case is a track (one box per step) case."""
    from tests import soak_ingest as si
    from tests import soak_streamed as soak
    os.makedirs(tmp_path / "inputs")
    (tmp_path / "inputs" / "namelist").write_text(si.NAMELIST)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(soak, "FILL_RATE", fill_rate)
    rng = np.random.default_rng(20260402 + int(fill_rate * 1e4))
    fails = []
    for c in range(24 if fill_rate == 0.0 else 12):
        fails += soak.one_case(rng, c, str(tmp_path)) if c % 3 else soak.one_track_case(rng, c, str(tmp_path))
    assert not fails, [f[:600] for f in fails[:3]]


def test_soak_slice_track_diagnostics():
    """60 random grids in both hemispheres, even and uneven axes, random boxes, both vorticity formulations, NaN patches, with and
    without track columns: values to 1e-11, positions exactly, against oracle/track_diagnostics.py."""
    from tests import soak_diag as soak
    rng = np.random.default_rng(20260403)
    fails = []
    for c in range(60):
        fails += soak.one_case(rng, c)
    assert not fails, [f[:400] for f in fails[:5]]


def test_soak_slice_repeatability():
    """The same inputs again and again give the same bits: ~12 s of tests/soak_repeat.py (resident passes with fixed and per-step boxes
    in both storage dtypes, streamed passes over two deflated fixtures with alternating chunk lengths, 2000 zlib streams through
    lec_inflate in all four launch forms)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_repeat.py"), "--seconds", "12"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "repeatability soak: 0 failures" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.skipif(not os.path.exists("/opt/conda/bin/python3.9"), reason="the NetCDF-4 writer of this soak needs the image's conda interpreter (h5py)")
def test_soak_slice_streamed_netcdf4_layouts():
    """40 random files, each as classic NetCDF and rewritten as NetCDF-4 in a random layout (chunk shapes, shuffle, deflate, fletcher32,
    contiguous / uncompressed variables mixed in): the NetCDF-4 file through the host preparation and through `lec_streamed` (device or
    host inflate, chunks from the registered file pages or pinned staging, 2-3 slots, random chunk lengths, time ranges, tracks with the
    850-hPa slices kept from the pass) gives the bits of the classic file's resident run."""
    import subprocess
    import sys
    if subprocess.run(["/opt/conda/bin/python3.9", "-c", "import h5py, scipy"], capture_output=True).returncode != 0:
        pytest.skip("the conda interpreter of this box has no h5py / scipy for the writer")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_deflated.py"), "--cases", "40", "--seed", "20260404"], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "seed 20260404: 0 failures" in r.stdout, (r.stdout[-3000:], r.stderr[-1500:])
    assert "'device_inflate': 0" not in r.stdout and "'registered': 0" not in r.stdout          # the device paths were really exercised
