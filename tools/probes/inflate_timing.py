#!/usr/bin/env python3
"""Cycles per round of lec_inflate's phases (GPU box; needs the debug builds: `for k in 1 2 3 4 6 7 8; do tools/build_variant.sh t$k
-DLEC_INFLATE_TIMING=$k; done`): window build / token decode / chain walk / write + flush (and, within the last, offsets + literals / matches / sync + flush), on ERA5-like chunks."""
import os
import subprocess
import sys
import zlib

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))


def measure():
    import inflate_check as ic
    from lorenzcycletoolkit_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    out = []
    for noise in (0.5, 0.02):
        y, x = np.mgrid[0:361, 0:720]
        distinct, raw = [], []
        for i in range(16):
            f = 250.0 + 30.0 * np.sin(x * 0.01 + i) * np.cos(y * 0.02) + noise * rng.standard_normal((361, 720))
            q = np.round((f - 250.0) / 0.002).clip(-32000, 32000).astype(np.int16)
            sh = q.view(np.uint8).reshape(-1, 2).T.copy().reshape(-1).tobytes()
            distinct.append(zlib.compress(sh, 4)); raw.append(sh)
        for n in (256, 4608):
            streams = [distinct[i % 16] for i in range(n)]
            sizes = [len(raw[i % 16]) for i in range(n)]
            desc, _o, status, dt = ic.run(lib, streams, sizes, "cuda:0", flags=2)
            st = status.astype(np.int64)
            out.append((noise, n, float((st[:, 1] & 0xffffffff).sum()) / float(st[:, 3].sum()), float(np.mean(sizes)) * n / float(st[:, 3].sum()), float(st[:, 2].sum()) / float(st[:, 3].sum())))
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        for r in measure():
            print("RES %g %d %.1f %.2f %.2f" % r)
        sys.exit(0)
    names = {1: "window", 2: "decode", 3: "walk", 4: "write + flush", 6: "offsets + literals", 7: "matches", 8: "sync + flush"}
    table = {}
    for k in (1, 2, 3, 4, 6, 7, 8):
        env = dict(os.environ, LEC_LIB=os.path.join(ROOT, "tools", "probes", f"liblec_t{k}.so"))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], capture_output=True, text=True, env=env, timeout=300)
        for ln in r.stdout.splitlines():
            if ln.startswith("RES"):
                _, noise, n, cyc, bpr, ms = ln.split()
                table.setdefault((float(noise), int(n)), {})[k] = (float(cyc), float(bpr), float(ms))
    for key, v in sorted(table.items()):
        print(f"noise {key[0]}, {key[1]} streams: {v[1][1]:.1f} bytes and {v[1][2]:.2f} matches per round, {sum(v[k][0] for k in (1, 2, 3, 4)):.0f} cycles:", "; ".join(f"{names[k]} {v[k][0]:.0f}" for k in sorted(v)), "cycles per round")
