#!/usr/bin/env python3
"""Wall clock of the PRODUCT -- ``python lorenzcycletoolkit.py <file> -r -f|-t [--device-ingest]`` from a file on disk to the CSVs --
and where it goes.  The reference times the same span: its log says "Fixed framework ran in 2.73 seconds" for the
36-step Catarina sample (samples/Catarina_NCEP-R2_fixed/log.txt:3-8, lorenzcycletoolkit.py:173,180,199: the framework call alone,
imports and data preparation not counted).

Runs on the GPU box:  python tools/bench_cli.py [--big-steps 96] [--out gpurun_out/r04_cli_end_to_end.json]

Cases: (i) the reference's own samples (tests/golden: Catarina -f, testdata -t), resident and --device-ingest, twice each (the first
run pays the page-in of torch on a fresh box); (ii) an ERA5-size shuffle + deflate NetCDF-4 file written to local disk by
tools/make_big_nc4.py (needs h5py: the image's /opt/conda interpreter), regional box / global box / track, cold (its pages dropped
with posix_fadvise) and warm page cache.  Phases come from the program itself (LEC_PHASES, lorenzcycletoolkit_amd/phases.py)."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "lorenzcycletoolkit.py")
CONDA = "/opt/conda/bin/python3.9"
NAMELIST_ERA5 = (";Variable;Units\nAir Temperature;t;K\nGeopotential;z;m**2/s**2\nOmega Velocity;w;Pa/s\nEastward Wind Component;u;m/s\n"
                 "Northward Wind Component;v;m/s\nLongitude;longitude\nLatitude;latitude\nTime;time\nVertical Level;level\n")


PAUSE = [0.0]


def run_case(workdir, argv, label, env_extra=None, timeout=1500):
    # On some boxes the driver makes a process wait while the device memory the PREVIOUS process freed is being wiped: a run that
    # allocates tens of GB milliseconds after such a process exited sits 1-1.5 s in its first large allocation (profiles/r05_notes.md
    # section 4: 0.07 s of `setup` after a 5-s pause, 0.9-1.4 s back to back, on the same box).  A user does not start the product in
    # the millisecond another 50-GB job ends; the big cases are therefore measured after a pause (recorded in the output).
    if PAUSE[0] > 0 and not label.startswith(("catarina", "testdata")):
        time.sleep(PAUSE[0])
    ph = os.path.join(workdir, "phases.json")
    if os.path.exists(ph):
        os.remove(ph)
    env = dict(os.environ, LEC_PHASES=ph, **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    shutil.rmtree(os.path.join(workdir, "LEC_Results"), ignore_errors=True)
    t0 = time.time()
    r = subprocess.run([sys.executable, CLI] + argv, cwd=workdir, env=env, capture_output=True, text=True, timeout=timeout)
    t1 = time.time()
    out = {"case": label, "argv": argv, "returncode": r.returncode, "wall_s": t1 - t0}
    if r.returncode != 0:
        out["stderr"] = r.stderr[-1500:]
        return out
    try:
        p = json.load(open(ph))
        at = t0
        phases = {}
        for name, t in p["marks"]:
            phases[name] = t - at
            at = t
        phases["exit"] = t1 - at
        out["phases_s"] = phases
        out["phases_frac"] = {k: v / (t1 - t0) for k, v in phases.items()}
    except Exception as e:       # noqa: BLE001
        out["phases_error"] = repr(e)
    logs = [ln for ln in (r.stdout + r.stderr).splitlines() if "framework ran in" in ln or "Device ingest" in ln]
    out["log"] = [ln.split(" - ", 3)[-1] for ln in logs]
    n_csv = sum(len(f) for _, _, f in os.walk(os.path.join(workdir, "LEC_Results")))
    out["files_written"] = n_csv
    return out


def drop_cache(path):
    """Drops the file's pages from the page cache without root: posix_fadvise(DONTNEED) on a clean file."""
    fd = os.open(path, os.O_RDONLY)
    try:
        os.fsync(fd)
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
    finally:
        os.close(fd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_cli_end_to_end.json"))
    ap.add_argument("--big-steps", type=int, default=96)
    ap.add_argument("--big-dir", default=os.environ.get("TMPDIR", "/tmp"))
    ap.add_argument("--skip-big", action="store_true")
    ap.add_argument("--layout", default="era5_int16")
    ap.add_argument("--long-steps", type=int, default=744, help="also time a LONG series on a small grid (a month of hourly steps, 37 x 41 x 80, "
                    "shuffle + deflate: the host phases -- chunk index, CSV writing -- dominate there); 0 = skip")
    ap.add_argument("--classic-steps", type=int, default=24, help="also time an uncompressed CLASSIC NetCDF file of this many ERA5-size steps (0 = skip)")
    ap.add_argument("--pause", type=float, default=5.0, help="seconds before every big case (the driver wipes the previous process's device memory)")
    a = ap.parse_args()
    PAUSE[0] = a.pause
    results = {"host": {"cpus": os.cpu_count()}, "pause_before_big_cases_s": a.pause, "cases": []}
    golden = os.path.join(ROOT, "tests", "golden")

    def save():
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(results, open(a.out, "w"), indent=1)

    with tempfile.TemporaryDirectory() as wd:
        os.makedirs(os.path.join(wd, "inputs"))
        shutil.copy(os.path.join(golden, "inputs", "namelist_NCEP-R2"), os.path.join(wd, "inputs", "namelist"))
        open(os.path.join(wd, "inputs", "box_limits"), "w").write("min_lon;-55\nmax_lon;-36\nmin_lat;-35\nmax_lat;-20\n")
        shutil.copy(os.path.join(golden, "inputs", "track_testdata_NCEP-R2"), os.path.join(wd, "inputs", "track"))
        cat, tst = os.path.join(golden, "Catarina_NCEP-R2.nc"), os.path.join(golden, "testdata_NCEP-R2.nc")
        for rep in ("first", "second"):
            for label, argv in (("catarina_fixed_resident", [cat, "-r", "-f", "--ingest", "host"]), ("catarina_fixed_device_ingest", [cat, "-r", "-f", "--device-ingest"]),
                                ("testdata_track_resident", [tst, "-r", "-t", "--ingest", "host"]), ("testdata_track_device_ingest", [tst, "-r", "-t", "--device-ingest"])):
                results["cases"].append(run_case(wd, argv, f"{label}:{rep}"))
                print(json.dumps(results["cases"][-1])[:400], flush=True)
                save()

    if not a.skip_big:
        have = os.path.exists(CONDA) and subprocess.run([CONDA, "-c", "import h5py"], capture_output=True).returncode == 0
        results["h5py_on_this_box"] = have
        st = os.statvfs(a.big_dir)
        results["big_dir_free_gb"] = st.f_bavail * st.f_frsize / 1e9
        if not have:
            results["big_file"] = "skipped: no h5py on this box to write the file with"
        else:
            big = os.path.join(a.big_dir, f"era5_like_T{a.big_steps}.nc")
            t0 = time.time()
            r = subprocess.run([CONDA, os.path.join(ROOT, "tools", "make_big_nc4.py"), "--out", big, "--timesteps", str(a.big_steps), "--layout", a.layout],
                               capture_output=True, text=True)
            results["big_file"] = {"path": big, "write_s": time.time() - t0, "note": r.stdout.strip()[-600:], "returncode": r.returncode,
                                   "stderr": r.stderr[-600:], "bytes": os.path.getsize(big) if os.path.exists(big) else 0}
            print(results["big_file"], flush=True)
            save()
            if r.returncode == 0:
                try:
                    with tempfile.TemporaryDirectory() as wd:
                        os.makedirs(os.path.join(wd, "inputs"))
                        new = a.layout == "cds_new"
                        nl = NAMELIST_ERA5.replace("Time;time", "Time;valid_time").replace("Vertical Level;level", "Vertical Level;pressure_level") if new else NAMELIST_ERA5
                        open(os.path.join(wd, "inputs", "namelist"), "w").write(nl)
                        T = a.big_steps
                        trk = "time;Lat;Lon\n" + "".join("2020-01-%02d-%02d00;%.2f;%.2f\n" % (1 + t // 24, t % 24, -35.0 + 0.1 * t, -50.0 + 0.15 * t)
                                                         for t in range(T))
                        open(os.path.join(wd, "inputs", "track"), "w").write(trk)
                        plans = [("regional_box_fixed", "min_lon;-80\nmax_lon;-20\nmin_lat;-60\nmax_lat;-10\n", ["-r", "-f"]),
                                 ("global_box_fixed", "min_lon;-180\nmax_lon;179.75\nmin_lat;-89.75\nmax_lat;89.75\n", ["-r", "-f"]),
                                 ("track_15deg_boxes", None, ["-r", "-t"])]
                        for label, box, flags in plans:
                            if box:
                                open(os.path.join(wd, "inputs", "box_limits"), "w").write(box)
                            for cache in ("cold", "warm"):
                                if cache == "cold":
                                    drop_cache(big)
                                results["cases"].append(run_case(wd, [big] + flags + ["--device-ingest"], f"big:{label}:device_ingest:{cache}"))
                                print(json.dumps(results["cases"][-1])[:600], flush=True)
                                save()
                        # the host-prepared path on the regional box (inflates the band's chunks on 16 host threads), warm cache
                        open(os.path.join(wd, "inputs", "box_limits"), "w").write(plans[0][1])
                        results["cases"].append(run_case(wd, [big, "-r", "-f", "--ingest", "host"], "big:regional_box_fixed:resident_host_prepared:warm", timeout=900))
                        print(json.dumps(results["cases"][-1])[:600], flush=True)
                        results["cases"].append(run_case(wd, [big, "-r", "-f"], "big:regional_box_fixed:default_flags:warm", timeout=900))
                        print(json.dumps(results["cases"][-1])[:600], flush=True)
                        save()
                finally:
                    if os.path.exists(big):
                        os.remove(big)
    if not a.skip_big and a.long_steps > 0 and results.get("h5py_on_this_box"):
        big = os.path.join(a.big_dir, f"era5_long_T{a.long_steps}.nc")
        t0 = time.time()
        r = subprocess.run([CONDA, os.path.join(ROOT, "tools", "make_big_nc4.py"), "--out", big, "--timesteps", str(a.long_steps), "--ny", "41", "--nx", "80"],
                           capture_output=True, text=True)
        results["long_file"] = {"path": big, "write_s": time.time() - t0, "note": r.stdout.strip()[-400:], "returncode": r.returncode, "stderr": r.stderr[-400:],
                                "bytes": os.path.getsize(big) if os.path.exists(big) else 0}
        print(results["long_file"], flush=True)
        save()
        if r.returncode == 0:
            try:
                with tempfile.TemporaryDirectory() as wd:
                    os.makedirs(os.path.join(wd, "inputs"))
                    open(os.path.join(wd, "inputs", "namelist"), "w").write(NAMELIST_ERA5)
                    open(os.path.join(wd, "inputs", "box_limits"), "w").write("min_lon;-80\nmax_lon;-20\nmin_lat;-60\nmax_lat;-10\n")
                    for rep in ("first", "second", "third"):
                        results["cases"].append(run_case(wd, [big, "-r", "-f"], f"long:regional_box_fixed:default_flags:{rep}", timeout=600))
                        print(json.dumps(results["cases"][-1])[:600], flush=True)
                        save()
            finally:
                if os.path.exists(big):
                    os.remove(big)
    if not a.skip_big and a.classic_steps > 0:
        big = os.path.join(a.big_dir, f"era5_classic_T{a.classic_steps}.nc")
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_big_classic.py"), "--out", big, "--timesteps", str(a.classic_steps)],
                           capture_output=True, text=True)
        results["classic_file"] = {"path": big, "write_s": time.time() - t0, "note": r.stdout.strip()[-400:], "returncode": r.returncode, "stderr": r.stderr[-400:]}
        print(results["classic_file"], flush=True)
        save()
        if r.returncode == 0:
            try:
                with tempfile.TemporaryDirectory() as wd:
                    os.makedirs(os.path.join(wd, "inputs"))
                    open(os.path.join(wd, "inputs", "namelist"), "w").write(NAMELIST_ERA5)
                    open(os.path.join(wd, "inputs", "box_limits"), "w").write("min_lon;-80\nmax_lon;-20\nmin_lat;-60\nmax_lat;-10\n")
                    for label, extra in (("default_flags", []), ("ingest_host", ["--ingest", "host"])):
                        for cache in ("cold", "warm"):
                            if cache == "cold":
                                drop_cache(big)
                            results["cases"].append(run_case(wd, [big, "-r", "-f"] + extra, f"classic:regional_box_fixed:{label}:{cache}", timeout=900))
                            print(json.dumps(results["cases"][-1])[:600], flush=True)
                            save()
            finally:
                if os.path.exists(big):
                    os.remove(big)
    save()
    print("written", a.out)


if __name__ == "__main__":
    main()
