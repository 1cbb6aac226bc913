#!/usr/bin/env python3
"""How large should the streamed pipeline's slots be?  (profiles/r05_notes.md section 4)

Writes the ERA5-size deflated file of tools/bench_cli.py once, then runs the global-box fixed framework through
``ingest.lec_fixed_streamed`` with several (chunk_steps, slots) pairs -- each in a process of its own, as the product runs -- and
reports set-up (device buffers allocated), chunk loop, drain, bytes of device buffers and a digest of the results (they must not
depend on the pair).

    python tools/bench_streamed_buffers.py [--timesteps 96] [--pairs 21x3,12x3,8x3,21x2,12x2] [--out gpurun_out/r05_streamed_buffers.json]
    python tools/bench_streamed_buffers.py --one <file> <chunk> <slots>         (the child)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONDA = "/opt/conda/bin/python3.9"
NAMELIST = (";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\nEastward Wind Component;u;m/s\n"
            "Northward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\nTime;time\nVertical Level;level\n")
LIMITS = (-180.0, 179.75, -89.75, 89.75)


def one(path, chunk, slots):
    t_start = time.perf_counter()
    import torch
    from lorenzcycletoolkit_amd import dataset as ds
    from lorenzcycletoolkit_amd import ingest
    torch.zeros(1, device="cuda:0")
    torch.cuda.synchronize()
    t_ready = time.perf_counter()
    args = argparse.Namespace(infile=path, fixed=True, track=False, trackfile=None, residuals=True, cdsapi=False, mpas=False, inflate="auto",
                              box_limits="inputs/box_limits")
    df = ds.read_namelist("inputs/namelist")
    data = ingest.prepare_streamed(args, "inputs/namelist")
    stats = {}
    t0 = time.perf_counter()
    res = ingest.lec_fixed_streamed(data.raw, data.plan, df, LIMITS, chunk_steps=chunk or None, slots=slots or None, stats=stats)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    digest = hashlib.sha256(res.scalars.cpu().numpy().tobytes() + res.levels.cpu().numpy().tobytes()).hexdigest()[:16]
    out = {"chunk_steps": stats["chunk_steps"], "chunks": stats["chunks"], "slots": slots or "default", "seconds": stats["seconds"],
           "device_buffer_gb": stats["device_buffer_bytes"] / 1e9, "call_s": t1 - t0, "imports_and_init_s": t_ready - t_start, "digest": digest}
    print("RESULT " + json.dumps(out), flush=True)
    os._exit(0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        return one(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--timesteps", type=int, default=96)
    ap.add_argument("--pairs", default="0x0,21x3,12x3,8x3,21x2,12x2,0x0")
    ap.add_argument("--pause", type=float, default=0.0, help="seconds between the runs (the driver wipes a process's memory after it exits)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_streamed_buffers.json"))
    a = ap.parse_args()
    big = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"era5_like_T{a.timesteps}.nc")
    r = subprocess.run([CONDA, os.path.join(ROOT, "tools", "make_big_nc4.py"), "--out", big, "--timesteps", str(a.timesteps)], capture_output=True, text=True)
    print(r.stdout.strip()[-300:], r.stderr[-300:], flush=True)
    results = []
    try:
        with tempfile.TemporaryDirectory() as wd:
            os.makedirs(os.path.join(wd, "inputs"))
            open(os.path.join(wd, "inputs", "namelist"), "w").write(NAMELIST)
            open(os.path.join(wd, "inputs", "box_limits"), "w").write("min_lon;-180\nmax_lon;179.75\nmin_lat;-89.75\nmax_lat;89.75\n")
            for pair in a.pairs.split(","):
                c, s = (int(x) for x in pair.split("x"))
                t0 = time.time()
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", big, str(c), str(s)], cwd=wd, capture_output=True, text=True, timeout=300)
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
                rec = json.loads(line[0][7:]) if line else {"error": p.stderr[-800:]}
                rec.update(asked=pair, wall_s=time.time() - t0)
                results.append(rec)
                print(json.dumps(rec), flush=True)
                json.dump(results, open(a.out, "w"), indent=1)
                time.sleep(a.pause)
    finally:
        if os.path.exists(big):
            os.remove(big)
    assert len({r.get("digest") for r in results if "digest" in r}) <= 1, "the results depend on the buffer sizes"


if __name__ == "__main__":
    main()
