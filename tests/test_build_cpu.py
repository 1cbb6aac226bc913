"""Build-time properties of the kernels that can be read off the ISA without a GPU (hipcc cross-compiles gfx950 here)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_inflate_kernels_use_no_flat_memory_instructions(tmp_path):
    """lec_inflate_kernel keeps its history ring in LDS and its output in HBM and orders the two only with wave_sync() (a
    wavefront-scope fence: no s_waitcnt).  That is sound while LDS traffic is DS instructions and HBM traffic GLOBAL instructions; a
    FLAT load that resolves to LDS at run time is NOT ordered with the wave's earlier DS writes.  The compiler builds such a load when a
    pointer is selected between the two address spaces (round 3's far-match form did: wrong bytes in the timing builds), so the ISA of
    both instantiations is checked for flat_load / flat_store / flat_atomic."""
    asm = tmp_path / "lec_inflate.s"
    src = os.path.join(ROOT, "lorenzcycletoolkit_amd", "csrc", "lec_inflate.hip")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", str(asm)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()
    kernels = re.findall(r"^(_ZN\S*lec_inflate_kernel\S*):", text, re.M)
    assert len(kernels) == 2, kernels
    for k in kernels:
        body = text[text.index(k + ":"): text.index(".Lfunc_end", text.index(k + ":"))]
        flat = re.findall(r"^\s*(flat_(?:load|store|atomic)\w*)", body, re.M)
        assert not flat, f"{k}: {sorted(set(flat))}"
        assert "ds_read" in body and "global_load" in body and "global_store" in body
