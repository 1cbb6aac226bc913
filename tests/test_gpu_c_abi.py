"""The boundary really is a C ABI: a plain-C program (tests/c_abi/lec_c_client.c: gcc, HIP runtime C API, no Python / torch /
C++) links liblec_hip.so, feeds it the same fields and tables, and must get the engine's numbers bit for bit."""
import os
import shutil
import subprocess

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from lorenzcycletoolkit_amd import _lib, tables
from lorenzcycletoolkit_amd.engine import LECEngine
from tests.helpers import synthetic_domain

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client_reproduces_the_engine(tmp_path):
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    exe = str(tmp_path / "lec_c_client")
    libdir = os.path.join(ROOT, "lorenzcycletoolkit_amd")
    subprocess.run(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "lec_c_client.c"), "-o", exe,
                    "-L/opt/rocm/lib", "-lamdhip64", "-L" + libdir, "-llec_hip",
                    "-Wl,-rpath," + libdir + ":/opt/rocm/lib"], check=True)
    dom = synthetic_domain(5, 6, 11, 130, seed=31)
    eng = LECEngine(dom.lat, dom.lon, dom.level, device="cuda:0")
    box = eng.box_from_limits(dom.lon[3], dom.lon[-4], dom.lat[1], dom.lat[-2])
    bt = tables.build_box_tables(dom.lat, dom.lon, [box])
    levtab, levtab2 = tables.level_tables(dom.level)
    tcoef = tables.time_coefs(dom.time_s)
    phi_scale = 1.0
    with open(tmp_path / "bundle.bin", "wb") as f:
        np.array([5, 6, 11, 130, bt.nxb_max, bt.nyb_max, int(bt.lon_uniform), 0], dtype=np.int32).tofile(f)
        np.array([phi_scale], dtype=np.float64).tofile(f)
        for a in (dom.tair, dom.u, dom.v, dom.omega, dom.geopt):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
        np.ascontiguousarray(bt.box, dtype=np.int32).tofile(f)
        for a in (bt.boxtab, bt.wlon, bt.glon, bt.lattab, levtab, tcoef, bt.boxtab2, bt.lattab2, levtab2):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    r = subprocess.run([exe, str(tmp_path / "bundle.bin"), str(tmp_path / "out.bin"), str(tmp_path / "packed.bin"), str(tmp_path / "az.csv")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float64)
    nt, nl = 5, 6
    scal = out[: nt * _lib.LEC_NSCALAR].reshape(nt, _lib.LEC_NSCALAR)
    lev = out[nt * _lib.LEC_NSCALAR:].reshape(nt, _lib.LEC_NLEVTAB, nl)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")
    res = eng.compute(dev(dom.tair), dev(dom.u), dev(dom.v), dev(dom.omega), dev(dom.geopt), [box], time_s=dom.time_s)
    assert np.array_equal(scal, res.scalars.cpu().numpy())
    assert np.array_equal(lev, res.levels.cpu().numpy(), equal_nan=True)
    assert "ok: 5 time steps" in r.stdout
    # ABI 9 from plain C.  The same box as a BOX-PACKED moving series (hipMemcpy2D crops, lec_dtdt, box_per_step): the moving
    # framework's numbers for five identical boxes, bit for bit ...
    mv = eng.compute(dev(dom.tair), dev(dom.u), dev(dom.v), dev(dom.omega), dev(dom.geopt), [box] * nt, time_s=dom.time_s, per_step_boxes=True)
    pk = np.fromfile(tmp_path / "packed.bin", dtype=np.float64)
    assert np.array_equal(pk[: nt * _lib.LEC_NSCALAR].reshape(nt, -1), mv.scalars.cpu().numpy())
    assert np.array_equal(pk[nt * _lib.LEC_NSCALAR:].reshape(nt, _lib.LEC_NLEVTAB, nl), mv.levels.cpu().numpy(), equal_nan=True)
    assert "box-packed series" in r.stdout
    # ... and the Az table as the text pandas writes for it (lec_format_csv_rows)
    import io
    import pandas as pd
    want = io.StringIO()
    idx = pd.date_range("2005-08-08", periods=nt, freq="6h").strftime("%Y-%m-%d %H:%M:%S")
    pd.DataFrame(lev[:, 0, :], index=idx).to_csv(want, mode="a", header=None)
    assert open(tmp_path / "az.csv", "rb").read() == want.getvalue().encode()
