#!/usr/bin/env python3
"""bench.py -- LEC time steps per second on synthetic 37 x 721 x 1440 fields (BASELINE.json metric).

One "step" = one pass of the whole hot path (lec_rowstats + lec_reduce, every energy, conversion,
boundary and generation term) over the job's batch of time steps.

Launch.  ``python bench.py --gpus N``: with WORLD_SIZE unset and N > 1 this process only STARTS N rank
processes (one per GPU, children created before anything here touches a GPU) and waits for them; under
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` each process is one rank
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  Ranks talk over RCCL (backend "nccl").
A --gpus that contradicts WORLD_SIZE, or more ranks than GPUs, is an error (non-zero exit), never a
silently smaller job.

Scaling modes.
  weak  (default)                 --timesteps T per GPU resident in HBM before the timed region (BASELINE config 3:
                                  T = 64); the global series has N * T steps, sharded contiguously.
  strong (--timesteps-global T)   the global series is fixed (BASELINE configs 4 / 5: T = 2048 / 4096) and sharded over
                                  the ranks; a shard that does not fit in HBM is generated and consumed in chunks of
                                  --chunk steps (one-step T halo per chunk, generated locally -- no exchange); the timed
                                  region is the kernels + collectives (synthetic generation is excluded and reported
                                  beside it).

Either way the data path has no collective: one all_reduce of the [28, L] NaN-level mask and one
gather of the per-time-step results per pass.  Prints ONE JSON line on rank 0.

The default N > 1 run (``--gpus N`` and nothing else: what the driver passes) also carries BASELINE configs 4 and 5: after the
weak-scaling headline and before the CPU leg the same process group runs two short strong-scaling legs -- the fixed box at
T = 2048 (chunked) and the moving box at T = 4096 -- and attaches them as ``config.strong_scaling`` (``--no-strong-legs`` for A/B runs).

What a ``--moving`` line times.  The reference clocks its whole call, dT/dt and the per-step slice included
(lorenzcycletoolkit.py:173-199, box_data.py:297-310).  With ``--moving-layout packed`` (the layout every ``-t`` path of the
product hands stage 1) those two are the PRODUCER's work: ``value`` is the consumer's rate (stage 1 + stage 2 + collectives on the
packed series), ``config.producer_ms`` the gathers and ``lec_dtdt`` timed on HIP events of their own, ``config.value_incl_producer``
the rate with both inside.  ``--moving-layout cube`` (rounds 1-4) slices by index and forms dT/dt per point inside the kernel.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--timesteps", type=int, default=64, help="weak scaling: time steps resident per GPU (BASELINE config 3: T=64)")
    ap.add_argument("--timesteps-global", type=int, default=0, help="strong scaling: length of the global series, sharded over the GPUs "
                    "(BASELINE configs 4 / 5: 2048 / 4096)")
    ap.add_argument("--chunk", type=int, default=0, help="strong scaling: time steps generated + processed at a time per GPU; 0 = the whole "
                    "shard if it fits in HBM, else as many steps as half of the free HBM holds")
    ap.add_argument("--storage", choices=["f64", "f32"], default="f64", help="storage dtype of the field cubes")
    ap.add_argument("--no-q", action="store_true", help="conversion-terms configuration: T,u,v,omega only (no Q, no Phi)")
    ap.add_argument("--cpu-baseline", choices=["full", "quick", "none"], default="full",
                    help="full: the bounded sample described in DESIGN.md section 6 (~1 min of host time); quick: a tiny one (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="same as --cpu-baseline none")
    ap.add_argument("--moving", action="store_true", help="semi-Lagrangian configuration: one 15x15 degree box per time step "
                    "on a track-extent crop of the 0.25 degree grid (BASELINE config 5)")
    ap.add_argument("--moving-layout", choices=["packed", "cube"], default="packed", help="--moving: how the series lies in HBM when the timed "
                    "region starts -- 'packed' (default): every time step's box alone, at the origin of its slab, with dT/dt as the series' own "
                    "data (what the streamed moving framework's ingest writes; include/lec_hip.h 'box-packed series'); 'cube': the whole "
                    "track-extent crop, boxes as index quadruples into it (rounds 1-4).  Same records, bit for bit")
    ap.add_argument("--tuning", type=str, default="", help="A/B runs: lec_tuning fields, e.g. kernel=row_sweep,tile_t=4 (default: the library's choice)")
    ap.add_argument("--force-dist", action="store_true", help="with one rank: still create the process group and run the collectives "
                    "(the N > 1 code path -- RCCL init, barrier, mask all_reduce, gather -- on a one-GPU box)")
    ap.add_argument("--digest-file", type=str, default=os.path.join(ROOT, "profiles", "series_digests.json"),
                    help="per-step checksums of the packed series of one-GPU runs: what config.series_equals_n1 compares an N-GPU run with")
    ap.add_argument("--write-digest", action="store_true", help="N = 1 only: add this run's series digest to --digest-file")
    ap.add_argument("--nonuniform-lon", action="store_true", help="stretched longitudes (a Gaussian / regridded-MPAS style axis): the kernels "
                    "then carry per-column trapezoid weights and d/dlon coefficient tables instead of the even-spacing fast path")
    ap.add_argument("--no-strong-legs", action="store_true", help="N > 1 default run: leave out the two strong-scaling legs (BASELINE configs 4 / 5) "
                    "that otherwise ride in config.strong_scaling")
    ap.add_argument("--leg-timesteps", type=str, default="2048,4096", help="global series lengths of the two strong-scaling legs (fixed box, "
                    "moving box); BASELINE configs 4 / 5 are 2048 / 4096 (rehearsals on one GPU use shorter ones)")
    ap.add_argument("--leg-steps", type=int, default=2, help="timed passes per strong-scaling leg (after one warm-up pass)")
    ap.add_argument("--n1-file", type=str, default=os.path.join(ROOT, "profiles", "strong_scaling_n1.json"),
                    help="N = 1 values of the strong-scaling configurations (by layout, with the kernel sources' digest): what speedup_vs_n1 divides by")
    ap.add_argument("--ny", type=int, default=721)
    ap.add_argument("--nx", type=int, default=1440)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N  without a torchrun environment
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """Starts args.gpus rank processes of this script and returns the job's exit code.  Nothing here initialises a GPU
    (torch.cuda.device_count() does not), so the children are ordinary child processes of a GPU-free parent."""
    import torch
    ndev = torch.cuda.device_count()
    backend = os.environ.get("LEC_DIST_BACKEND", "nccl")
    if backend == "nccl" and ndev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} asked but this node shows {ndev} GPU(s); one rank per GPU is required "
              f"(launch on a node with {args.gpus} GPUs, or rehearse the N > 1 path with LEC_DIST_BACKEND=gloo)", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for pr in list(pending):
                code = pr.poll()
                if code is None:
                    continue
                pending.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in pending:        # one rank failed: the others would wait in a collective for ever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline + full-size parity: the NumPy oracle (the reference's eager op sequence) on the GPU box's host cores
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_run(fields, lat, lon, level, time_s, limits):
    """(seconds, scalars, levels) of the oracle's fixed framework (all 16 terms) on host cubes [nt, nl, ny, nx]."""
    from oracle import lec_oracle as o
    dom = o.Domain(fields["tair"], fields["u"], fields["v"], fields["omega"], fields["geopt"], lat, lon, level, time_s)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):          # a box that holds the polar rows divides by cos(90 deg) in the reference too
        sc, lv = o.lec_fixed(dom, *limits)
    return time.perf_counter() - t0, sc, lv


def _synthetic_band_host(nt, ny, seed):
    """Host copy of the synthetic recipe on a 37 x ny x 1440 latitude band (the all-cores figure: every worker makes its own)."""
    from lorenzcycletoolkit_amd.synthetic import era5_like_levels
    from oracle import lec_oracle as o
    rng = np.random.default_rng(seed)
    nx = 1440
    level = era5_like_levels()
    lat = -45.0 + 0.25 * np.arange(ny)
    lon = np.linspace(-180.0, 179.75, nx)
    p = level[None, :, None, None]
    phi, lam = np.deg2rad(lat)[None, None, :, None], np.deg2rad(lon)[None, None, None, :]
    shp = (nt, level.size, ny, nx)
    f = {"tair": 288.0 * (p / 1e5) ** 0.19 + 10.0 * np.cos(2 * phi) * (p / 1e5) + rng.standard_normal(shp),
         "u": 25.0 * np.cos(phi) * (1 - p / 1.2e5) + 5.0 * rng.standard_normal(shp),
         "v": 3.0 * np.sin(2 * lam) * np.cos(phi) + 3.0 * rng.standard_normal(shp),
         "omega": 0.05 * np.sin(3 * lam) * np.cos(phi) + 0.1 * rng.standard_normal(shp),
         "geopt": o.G * 7000.0 * np.log(1e5 / p) + 100.0 * rng.standard_normal(shp)}
    return f, lat, lon, level


def _oracle_band_worker(a):
    """(start, end) of the oracle call on the worker's own band, on the machine-wide monotonic clock."""
    nt, ny, seed = a
    f, lat, lon, level = _synthetic_band_host(nt, ny, seed)
    t0 = time.monotonic()
    _oracle_run(f, lat, lon, level, np.arange(nt) * 3600.0, (lon[0], lon[-1], lat[0], lat[-1]))
    return t0, time.monotonic()


def _scale_err(a, r):
    a, r = np.asarray(a, dtype=np.float64), np.asarray(r, dtype=np.float64)
    if not np.array_equal(np.isnan(a), np.isnan(r)):
        return float("inf")
    ok = ~np.isnan(r)
    if not ok.any():
        return 0.0
    den = float(np.max(np.abs(r[ok])))
    return float(np.max(np.abs(a[ok] - r[ok])) / den) if den > 0 else float(np.max(np.abs(a[ok] - r[ok])))


_TERMS16 = ("Az", "Ae", "Kz", "Ke", "Cz", "Ca", "Ck", "Ce", "BAz", "BAe", "BKz", "BKe", "BΦZ", "BΦE", "Gz", "Ge")      # all 16 (the results
                                                                                                                       # CSV of lec_fixed drops the two BPhi)

def _compare_with_oracle(got_s, got_l, ref_s, ref_l, n):
    """Worst |engine - oracle| / max |oracle| over the 16 terms and over the level tables, first `n` steps of both."""
    worst, worst_term = 0.0, ""
    for name in _TERMS16:
        e = _scale_err(np.asarray(got_s[name])[:n], np.asarray(ref_s[name])[:n])
        if e > worst:
            worst, worst_term = e, name
    lworst, lworst_term = 0.0, ""
    for name, ref in ref_l.items():
        ref = np.asarray(ref, dtype=np.float64)
        got = np.asarray(got_l[name])[:n]
        if ref.ndim == 1:
            ref = np.broadcast_to(ref, got.shape)
        e = _scale_err(got, ref[:n])
        if e > lworst:
            lworst, lworst_term = e, name
    return {"worst_rel_to_scale": worst, "term": worst_term, "levels_worst_rel_to_scale": lworst, "levels_term": lworst_term,
            "steps": int(n), "terms_compared": len(_TERMS16), "level_tables_compared": len(ref_l),
            "checker": "oracle/lec_oracle.py (NumPy fp64 restatement of the reference's un-factored formulas)", "tolerance": 1e-9,
            "ok": bool(worst <= 1e-9 and lworst <= 1e-9)}


def _host_counts():
    ncpu = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = ncpu
    return ncpu, usable


def cpu_leg_steps(kind):
    """Time steps of the fixed-box CPU leg (each 5.4 s of one host thread at 37 x 721 x 1440)."""
    return 3 if kind in ("full", "single") else 2


def cpu_baseline_and_parity(kind, eng, sub, lat, lon, level, ts, device, data_note):
    """The CPU leg of a fixed-box line.  The oracle evaluates `sub` -- the FIRST TIME STEPS OF THE SYNTHETIC CUBE THE GPU HAS JUST BEEN
    TIMED ON (device tensors [nt, nl, ny, nx], kept back from the resident cube or the first chunk) -- copied to the host, on the box
    without the polar rows (the reference divides by cos 90 deg there); the engine runs the same sub-cube and every one of
    the 16 terms and 21 level tables is compared (`parity`).  The oracle's wall time on exactly that call is the one-thread CPU
    baseline (`cpu_baseline`, kind "port").
    kind "full": nt = 3 at the benchmark's size, two repetitions (the first also yields the parity), plus a best-effort all-cores figure;
    kind "quick" (tests): nt = 2, one repetition;
    kind "single" (what "full" becomes on rank 0 of an N > 1 run, the peers waiting in the closing barrier): nt = 3, one repetition, no
    all-cores figure -- the other ranks' processes occupy cores and the N = 1 line of the same scaling run carries it."""
    import multiprocessing as mp
    import torch
    ncpu, usable = _host_counts()
    nt = int(sub["tair"].shape[0])
    nx = lon.size
    south, north = (lat[1], lat[-2]) if abs(lat[0]) >= 90.0 - 1e-9 else (lat[0], lat[-1])
    limits = (lon[0], lon[-1], south, north)
    box = eng.box_from_limits(*limits)
    res = eng.compute(sub["tair"], sub["u"], sub["v"], sub["omega"], sub["geopt"], [box], time_s=ts)
    torch.cuda.synchronize(device)
    got_s, got_l = res.scalars_dict(), res.levels_dict()
    host = {k: np.ascontiguousarray(v.double().cpu().numpy()) for k, v in sub.items()}
    reps = []
    for rep in range(2 if kind == "full" else 1):
        dt, ref_s, ref_l = _oracle_run(host, lat, lon, level, ts, limits)
        reps.append(dt)
    parity = _compare_with_oracle(got_s, got_l, ref_s, ref_l, nt)
    parity.update(box=f"lon [{limits[0]}, {limits[1]}] lat [{limits[2]}, {limits[3]}] ({box[3] - box[2] + 1} x {box[1] - box[0] + 1} points)",
                  data=data_note % nt)
    best = float(np.median(reps))
    out = {"value": nt / best, "unit": "timesteps/s", "cores": 1, "kind": "port", "host_cpu_count": ncpu, "usable_cores": usable,
           "seconds_per_timestep": best / nt,
           "sample": f"NumPy fp64 oracle (the reference's eager op order), 1 thread, {nt} time steps of the cube the GPU was timed on at 37x{box[3] - box[2] + 1}x{nx} "
                     f"(box without the polar rows), {len(reps)} repetition(s): {', '.join(f'{r:.1f}' for r in reps)} s, median; the same call gives `parity`"}
    if kind == "full":
        workers = max(1, min(usable, 32))
        try:
            ctx = mp.get_context("spawn")
            with ctx.Pool(workers) as pool:
                spans = pool.map(_oracle_band_worker, [(2, 91, 100 + i) for i in range(workers)])
            wall = max(b for _, b in spans) - min(a for a, _ in spans)      # first oracle call starts .. last one ends (process start and data generation excluded)
            out["all_cores"] = {"value": workers * 2 * (91.0 / 721.0) / wall, "unit": "timesteps/s", "cores": workers,
                                "sample": f"{workers} worker processes (one per usable core, at most 32: each holds ~2.5 GB of 4-D temporaries like the "
                                          f"reference), each 2 time steps of its own 37x91x1440 band; {wall:.1f} s from the first worker's oracle call "
                                          f"starting to the last one's ending, scaled by 91/721"}
        except Exception as e:      # a host that cannot fork workers still reports the one-thread figure
            out["all_cores"] = {"value": None, "error": repr(e)}
    return out, parity


def cpu_leg_steps_moving(kind, t_local):
    """Boxes of the moving CPU leg: a 61 x 61 x 37 box takes the oracle tens of milliseconds, so the full leg takes 32 of them."""
    return max(1, min(32 if kind == "full" else 3, t_local))


def cpu_baseline_and_parity_moving(kind, crop, limits, lat, lon, level, ts, got_s, got_l, n):
    """The CPU leg of a --moving line.  `crop`: the first n (+ 1: the time neighbour of the last one) steps of the track-extent crop
    rank 0's series was produced from (device tensors), `limits`: the boxes' (west, east, south, north) per held step.  The oracle
    runs the reference's moving framework on it -- dT/dt over the crop's time axis INSIDE the clock, as the reference's own clock has
    it (lorenzcycletoolkit.py:173,184-186), then one BoxData per step (lec_moving_framework.py:639-745) -- on one thread; `got_s` /
    `got_l`: the 16 terms and 21 tables of the same steps as the TIMED GPU pass delivered them (the packed or the cube layout,
    whichever was timed).  The held steps beyond `n` exist only so that step n - 1 is differentiated as the series differentiates it."""
    from oracle import lec_oracle as o
    ncpu, usable = _host_counts()
    held = int(crop["tair"].shape[0])
    host = {k: np.ascontiguousarray(v.double().cpu().numpy()) for k, v in crop.items()}
    dom = o.Domain(host["tair"], host["u"], host["v"], host["omega"], host["geopt"], lat, lon, level, np.asarray(ts[:held], dtype=np.float64))
    reps = []
    for rep in range(2 if kind == "full" else 1):
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            ref_s, ref_l = o.lec_moving(dom, limits[:held], residuals=False)
        reps.append(time.perf_counter() - t0)
    parity = _compare_with_oracle(got_s, got_l, ref_s, ref_l, n)
    parity.update(box="one 15 x 15 degree box (61 x 61 points) per time step, on the synthetic track",
                  data=f"the first {n} time steps of the track-extent crop the timed series was produced from, copied to the host "
                       f"({held} steps held: the last one's time neighbour)", compared_with="the records of the timed GPU pass")
    best = float(np.median(reps))
    out = {"value": held / best, "unit": "timesteps/s", "cores": 1, "kind": "port", "host_cpu_count": ncpu, "usable_cores": usable,
           "seconds_per_timestep": best / held,
           "sample": f"NumPy fp64 oracle of the moving framework (dT/dt over the crop's time axis, then one box per step: the reference's eager op "
                     f"order and what its own clock spans), 1 thread, {held} time steps of the 37x{lat.size}x{lon.size} crop with 61x61 boxes, "
                     f"{len(reps)} repetition(s): {', '.join(f'{r:.2f}' for r in reps)} s, median; the same call gives `parity`"}
    return out, parity


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
class RankContext:
    """This process's place in the job: rank, GPU, process group (created once; the headline and the strong-scaling legs share it)."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("LEC_DIST_BACKEND", "nccl")   # nccl == RCCL on ROCm; gloo only rehearses the N > 1 path on one GPU
        ndev = torch.cuda.device_count()
        if ndev < 1:
            raise SystemExit("bench.py: no GPU visible: the HIP path is the only path")
        if self.backend == "nccl" and self.world > ndev:
            raise SystemExit(f"bench.py: {self.world} ranks but {ndev} GPU(s): RCCL needs one GPU per rank")
        self.local_rank = local_rank % ndev
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.force_dist = bool(args.force_dist)
        self.use_dist = self.world > 1 or self.force_dist      # --force-dist: the N > 1 code path (process group, collectives) with ONE rank
        self.stdout_fd = None
        if self.use_dist:
            # RCCL and gloo print their banners ("RCCL version : ...", "[Gloo] Rank 0 is connected to ...") on STDOUT: the contract is ONE
            # JSON line there, so everything the libraries write goes to stderr until the line itself is printed
            sys.stdout.flush()
            self.stdout_fd = os.dup(1)
            os.dup2(2, 1)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1:
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
            dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world)      # the rank's GPU is already current
        # who is really there: ranks the backend connected, the GPU each one drives (N > 1 runs happen on nodes nobody watches)
        from lorenzcycletoolkit_amd.parallel import ranks_and_devices
        self.who = ranks_and_devices(self.device) if self.use_dist else None
        if self.who is not None and (self.who["ranks_seen"] != self.world or (self.backend == "nccl" and not self.who["devices_distinct"])):
            raise SystemExit(f"bench.py: rank {self.rank}: the process group is not what was asked for: {json.dumps(self.who)}")

    def barrier(self):
        import torch.distributed as dist
        if self.backend == "nccl":
            dist.barrier(device_ids=[self.local_rank])
        else:
            dist.barrier()

    def sync(self, with_barrier=True):
        import torch
        if self.use_dist and with_barrier:
            self.barrier()
        torch.cuda.synchronize()

    def reduce_max(self, values):
        """Element-wise maximum over the ranks of a short list of floats."""
        import torch
        import torch.distributed as dist
        if not self.use_dist:
            return [float(v) for v in values]
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device)
        if self.backend == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def print_line(self, out):
        if self.stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(self.stdout_fd, 1)
        print(json.dumps(out, ensure_ascii=False), flush=True)
        if self.stdout_fd is not None:
            os.dup2(2, 1)


def n1_key(args, T_global, packed):
    """Key of a strong-scaling configuration in --n1-file: the series' layout is part of it (a packed and a cube series are different work)."""
    kind = ("moving_" + ("packed" if packed else "cube")) if args.moving else "fixed"
    return f"{kind}_{args.storage}_{'noq' if args.no_q else 'all'}_T{T_global}"


def measure(args, ctx, cpu_kind="none"):
    """One bench configuration on the job's process group: generation, warm-up, the timed passes, the instrumented passes and the
    self-checks.  Returns (out, cpu_leg): rank 0's JSON object (None elsewhere) and -- rank 0, cpu_kind != "none" -- a callable
    that runs the CPU leg and fills out["cpu_baseline"] / out["parity"]; it holds only the few time steps it needs, so the caller
    may run other configurations in between (the strong-scaling legs of the default N > 1 line)."""
    import torch
    import torch.distributed as dist

    from lorenzcycletoolkit_amd.engine import LECEngine
    from lorenzcycletoolkit_amd.parallel import (SeriesGatherer, halo_range, merge_dropmask, record_checksums, shard_range, verify_gather)
    from lorenzcycletoolkit_amd.synthetic import era5_like_levels, synthetic_cube

    world, rank, device, backend, use_dist, who = ctx.world, ctx.rank, ctx.device, ctx.backend, ctx.use_dist, ctx.who
    barrier, sync = ctx.barrier, ctx.sync

    level = era5_like_levels()
    lat = np.linspace(-90.0, 90.0, args.ny)
    lon = np.linspace(-180.0, 180.0 - 360.0 / args.nx, args.nx)
    if args.nonuniform_lon:      # strictly ascending, spacing modulated by +-30 %
        lon = np.sort(lon + 0.3 * (360.0 / args.nx) * np.sin(np.linspace(0.0, 7.0, args.nx)))
    if args.moving:
        # the reference crops the data to the track extent +- (half box + one grid step) first
        # (select_area.py:297-313); the synthetic track wanders over 25 x 45 degrees
        lat = np.arange(-57.75, -17.5 + 1e-9, 0.25)
        lon = np.arange(-80.25, -19.75 + 1e-9, 0.25)
    nl = level.size
    tdtype = torch.float64 if args.storage == "f64" else torch.float32
    esz = 8 if args.storage == "f64" else 4
    with_q = not args.no_q
    nfields = 4 if args.no_q else 5

    strong = args.timesteps_global > 0
    T_global = args.timesteps_global if strong else args.timesteps * world
    if T_global < max(world, 2 if with_q else 1):
        raise SystemExit("bench.py: need at least one time step per GPU, and 2 in all for dT/dt (or --no-q)")
    t0, t1 = shard_range(T_global, world, rank)
    T_local = t1 - t0
    time_s = np.arange(T_global) * 3600.0
    eng = LECEngine(lat, lon, level, device=device)
    box = eng.box_from_limits(lon[0], lon[-1], lat[0], lat[-1])

    def limits_of(a, b):
        """(west, east, south, north) of the moving box at the global steps [a, b): 15 x 15 degrees about the synthetic track."""
        tg = np.arange(a, b)
        clat = -37.5 + 12.0 * np.sin(2 * np.pi * tg / 400.0)
        clon = -50.0 + 22.0 * np.cos(2 * np.pi * tg / 700.0)
        return [(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(clat, clon)]

    def boxes_of(a, b):
        if not args.moving:
            return [box]
        return [eng.box_from_limits(*lim) for lim in limits_of(a, b)]

    # chunks of the shard: resident (one chunk, generated before the timed region) or streamed through HBM
    step_bytes = 5 * nl * lat.size * lon.size * esz
    free_b, _total_b = torch.cuda.mem_get_info(device)
    chunk = args.chunk
    if chunk <= 0:          # the whole shard if it fits (fields + generation temporaries), else as many steps as half of the free HBM holds
        chunk = T_local if (T_local + 2) * step_bytes * 1.25 < 0.8 * free_b else int(0.5 * free_b / (1.25 * step_bytes)) - 2
    chunk = max(1, min(chunk, T_local))
    chunks = [(a, min(a + chunk, t1)) for a in range(t0, t1, chunk)]
    resident = len(chunks) == 1
    chunk_boxes = {c: boxes_of(*c) for c in chunks}      # host work (nearest-grid-point boxes of a chunk's steps): once, outside the timed region
    all_boxes = [bx for c in chunks for bx in chunk_boxes[c]] if args.moving else [box]
    nyb_max = max(b[3] - b[2] + 1 for b in all_boxes) if args.moving else lat.size
    nxb_max = max(b[1] - b[0] + 1 for b in all_boxes) if args.moving else lon.size
    packed = bool(args.moving and args.moving_layout == "packed" and with_q)
    if args.moving:        # thousands of boxes: build and upload their tables once (PreparedBoxes), not at every call
        chunk_boxes = {c: eng.prepare_boxes(bx, nyb_min=nyb_max, packed=packed) for c, bx in chunk_boxes.items()}
        all_boxes = eng.prepare_boxes(all_boxes, nyb_min=nyb_max, packed=packed)
    tc_all = eng.time_coefs_device(time_s) if with_q else None      # d/dt coefficients of the whole axis on the device: no upload per call
    crop = {}                      # a packed series: the crop it was produced from (resident runs keep it: the producer is timed on it)
    crop_head = {}                 # ... its first steps, always kept by rank 0: the kernel cross-check and the CPU leg read them
    # (ranks that SHARE a GPU -- the gloo rehearsals of the N > 1 path on a one-GPU box -- do not hold a crop each beside their series)
    keep_whole_crop = not use_dist or world == 1 or bool(who["devices_distinct"])
    producer_ms = {"pack": [], "dtdt": []}

    def pack(f, h0, h1, timing=None):
        """The box-packed series of the held steps [h0, h1) of the crop `f`: every step's box gathered out of it (the reference's
        per-step slice, box_data.py:297-310 -- done by whoever produces the data: here pack_series, in the product lec_ingest), dT/dt over
        the series' time axis formed on each step's own box (fp64 storage: as a cube, lec_dtdt; fp32: the two neighbours' T).  Slabs of the
        tallest / widest box of the WHOLE series: every chunk's cubes have one shape."""
        return eng.pack_series(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], boxes_of(h0, h1), tc_all[h0:h1], ny=nyb_max, nx=nxb_max, timing=timing)

    def note_producer(tm):
        torch.cuda.synchronize()
        producer_ms["pack"].append(tm["pack"][0].elapsed_time(tm["pack"][1]))
        producer_ms["dtdt"].append(tm["dtdt"][0].elapsed_time(tm["dtdt"][1]) if "dtdt" in tm else 0.0)

    def generate(a, b, keep_crop=False):
        """(first held step, end, fields, the call's dT/dt arguments) of the steps [a, b) as they lie in HBM for the timed passes."""
        h0, h1 = halo_range(a, b, T_global)                  # one-step halo for dT/dt (thermodynamics.py:109-110)
        f = synthetic_cube(h1 - h0, level, lat, lon, device=device, dtype=tdtype, seed=1234, t0_global=h0)
        if args.no_q:
            f["geopt"] = None
        if not packed:
            return h0, h1, f, ({"tcoef": tc_all[h0:h1]} if with_q else {})
        tm = {}
        ps = pack(f, h0, h1, timing=tm)
        note_producer(tm)
        if keep_crop and not keep_whole_crop:      # the crop will not be there to time the producer on afterwards: a second, warmed-up production now
            del ps
            tm = {}
            ps = pack(f, h0, h1, timing=tm)
            note_producer(tm)
        if keep_crop:
            if rank == 0:
                n = min(b - a, 32) + 2
                crop_head.update(h0=h0, f={k: v[:n].clone() for k, v in f.items()})
            if keep_whole_crop:
                crop.update(h0=h0, h1=h1, f=f)
        del f
        cut = lambda x: None if x is None else x[a - h0: b - h0].contiguous()
        kw = {k: cut(ps[k]) for k in ("dTdt", "tm", "tp") if k in ps}
        if "tm" in kw:
            kw["tcoef"] = tc_all[a:b]
        return a, b, {k: cut(ps[k]) for k in ("tair", "u", "v", "omega", "geopt")}, kw

    held = generate(*chunks[0], keep_crop=True) if resident else None
    produced_once = {k: list(v) for k, v in producer_ms.items()}      # (a first call: it may pay one-time costs -- used only where the crop is not kept)
    producer_ms["pack"].clear(); producer_ms["dtdt"].clear()
    # Everything a pass writes is allocated ONCE, here: the row records, the NaN counters, and (SeriesGatherer) the send / receive
    # buffers of the gather in two pipeline slots.  lec_reduce writes its packed [T_local, 16 + 21 nl] records straight into the
    # slot's send buffer; the gather of pass i is waited for when its slot comes round again, so it overlaps the kernels of pass i + 1.
    rows = torch.empty((T_local, nl, nyb_max, 32), dtype=torch.float64, device=device)
    nanflag = torch.empty((T_local,), dtype=torch.int32, device=device)
    gat = SeriesGatherer(T_global, LECEngine.packed_width(nl), device, dst=0, slots=2, force=ctx.force_dist)
    merge = (lambda m: merge_dropmask(m)) if (use_dist and not args.moving) else None
    kernel_ms = []
    gen_s = [0.0]
    tuning = None
    if args.tuning:
        tuning = {k: (v if k in ("kernel", "order") else int(v)) for k, v in (kv.split("=") for kv in args.tuning.split(","))}
    stage1 = dict(with_q=with_q, tuning=tuning, per_step_boxes=bool(args.moving))

    pass_no = [0]
    last = {}

    def one_pass(record, seg=None):
        """One pass over this rank's shard: stage 1 (per chunk), stage 2 into the gather's send buffer, gather started.
        Returns the seconds inside the timed region for the streamed mode (None when resident: the caller's wall clock counts).
        `seg`: a dict that receives the seconds of every segment of this pass, each closed by a device synchronisation
        (the instrumented passes after the timed loop; the timed passes never synchronise inside)."""
        slot = pass_no[0] % gat.slots
        pass_no[0] += 1
        last["slot"] = slot
        timed = 0.0
        tick = [time.perf_counter()]

        def mark(name):
            if seg is not None:
                torch.cuda.synchronize()
                now = time.perf_counter()
                seg[name] = seg.get(name, 0.0) + (now - tick[0])
                tick[0] = now

        if gat._pending[slot]:
            last["series"] = gat.finish(slot)         # the exchange of the pass that used this slot two passes ago
        out = gat.send(slot)
        mark("gather_finish_previous")
        if resident:
            h0, h1, f, dkw = held
            timing = [] if record else None
            eng.rowstats(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], all_boxes, t_begin=t0 - h0, t_count=T_local, timing=timing,
                         rows_out=rows, **dkw, **stage1)
            if record:
                kernel_ms.extend(timing)
            mark("stage1")
        else:
            for (a, b) in chunks:
                g0 = time.perf_counter()
                h0, h1, f, dkw = generate(a, b)
                torch.cuda.synchronize()
                gen_s[0] += time.perf_counter() - g0
                timing = [] if record else None
                tic = time.perf_counter()
                eng.rowstats(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], chunk_boxes[(a, b)], t_begin=a - h0, t_count=b - a, timing=timing,
                             rows_out=rows[a - t0:b - t0], **dkw, **stage1)
                torch.cuda.synchronize()
                timed += time.perf_counter() - tic
                if record:
                    kernel_ms.extend(timing)
                del f, dkw
            sync()
            tick[0] = time.perf_counter()
        tic = time.perf_counter()
        mask_s = [0.0]
        mg = merge
        if seg is not None and merge is not None:
            def mg(m):
                torch.cuda.synchronize()
                a = time.perf_counter()
                merge(m)
                torch.cuda.synchronize()
                mask_s[0] = time.perf_counter() - a
        res = eng.reduce(rows, all_boxes, merge_dropmask=mg, drop_any_time=not args.moving, out=out, nanflag_out=nanflag)
        last["res"] = res
        mark("stage2")
        if seg is not None and merge is not None:
            seg["stage2"] -= mask_s[0]
            seg["mask_all_reduce"] = seg.get("mask_all_reduce", 0.0) + mask_s[0]
        gat.profile = {} if seg is not None else None
        gat.start(slot)                               # the job's only collective besides the mask: RCCL gather to rank 0 over xGMI
        if seg is not None:
            last["series"] = gat.finish(slot)
            mark("gather")
            for k, v in gat.profile.items():
                seg["gather." + k] = seg.get("gather." + k, 0.0) + v
            gat.profile = None
        if resident:
            return None
        sync()
        timed += time.perf_counter() - tic
        return timed

    def drain():
        """Completes the exchanges still in flight, oldest first: last["series"] ends as the series of the LAST pass (as last["res"])."""
        newest = last.get("slot", 0)
        for sl in [x for x in range(gat.slots) if x != newest] + [newest]:
            if gat._pending[sl]:
                last["series"] = gat.finish(sl)

    for _ in range(args.warmup):
        one_pass(False)
    drain()
    sync()
    gen_s[0] = 0.0
    producer_ms["pack"].clear(); producer_ms["dtdt"].clear()
    tic = time.perf_counter()
    timed_total = 0.0
    for _ in range(args.steps):
        timed = one_pass(True)
        if timed is not None:
            timed_total += timed
    drain()                                           # every pass's series has arrived on rank 0 inside the timed region
    sync()
    wall = time.perf_counter() - tic
    elapsed = wall if resident else timed_total
    elapsed_own = elapsed
    elapsed, wall = ctx.reduce_max([elapsed, wall])
    res = last["res"]
    series = last.get("series")
    # every rank's block of the gathered series against the checksums of what that rank sent (not only rank 0's own block)
    peers = verify_gather(series, res.packed, T_global) if use_dist else None
    per_rank = None
    if use_dist:      # each rank's own clock for the timed region (the line's ms_per_step is their maximum)
        mine = torch.tensor([elapsed_own], dtype=torch.float64, device=device)
        from lorenzcycletoolkit_amd.parallel import _all_gather_small
        per_rank = [float(x) / args.steps * 1e3 for x in _all_gather_small(mine).view(-1)]

    # The PRODUCER of a packed series (the per-step slice and dT/dt, which the reference's clock spans: lorenzcycletoolkit.py:173-199,
    # box_data.py:297-310), per pass over this rank's shard, on HIP events of its own.  Streamed runs produce every chunk inside every
    # pass (generate(): the events were taken there); a resident run re-produces its series from the crop it kept, three times.
    producer = None
    if packed:
        repack_same = None
        if resident and not crop:          # the crop was not kept (ranks share a GPU): what producing the series took, once, before the timed passes
            per_pass = [float(produced_once["pack"][-1]), float(produced_once["dtdt"][-1])]
        elif resident:
            for i in range(3):
                tm = {}
                ps = pack(crop["f"], crop["h0"], crop["h1"], timing=tm)
                note_producer(tm)
                if i == 0:      # and it is the series the timed passes read
                    a0 = t0 - crop["h0"]
                    repack_same = bool(torch.equal(ps["tair"][a0:a0 + T_local], held[2]["tair"]) and
                                       all(torch.equal(ps[k][a0:a0 + T_local], held[3][k]) for k in ("dTdt", "tm", "tp") if k in held[3]))
                del ps
            per_pass = [float(np.median(producer_ms["pack"])), float(np.median(producer_ms["dtdt"]))]
        else:       # one entry per chunk and pass: the sum over a pass's chunks
            per_pass = [float(np.sum(producer_ms["pack"])) / args.steps, float(np.sum(producer_ms["dtdt"])) / args.steps]
        per_pass = ctx.reduce_max(per_pass)
        producer = {"pack": per_pass[0], "dtdt": per_pass[1], "total": per_pass[0] + per_pass[1],
                    "unit": "ms per pass over a rank's shard, HIP events, max over ranks",
                    "pack_is": "the per-step gathers out of the crop (5 fields + T of the two time neighbours): LECEngine.pack_boxes = lec_ingest with "
                               "per-step origins, the gather the product's streamed path runs",
                    "dtdt_is": ("lec_dtdt: dT/dt of the packed series as an fp64 cube" if args.storage == "f64" else "none: fp32 storage hands T of the two neighbours over"),
                    "how": ("the second of two productions before the timed passes (ranks share a GPU: the crop is not kept beside the series; the other ranks' work runs on the same GPU meanwhile)"
                            if (resident and not crop) else "re-produced 3 times from the resident crop after the timed passes (median)" if resident else
                            "taken inside every timed pass's generation of every chunk (sum over a pass's chunks, mean over passes)")}
        if repack_same is not None:
            producer["reproduced_series_is_the_timed_one"] = repack_same

    # Where a pass's time goes: three instrumented passes (every segment closed by a device synchronisation, the gather completed
    # inside its pass), median per segment.  Reported beside the timed figure, never inside it.
    seg_runs = []
    if resident:
        for _ in range(3):
            sync()
            sg = {}
            a = time.perf_counter()
            one_pass(False, seg=sg)
            sg["pass_total_synchronised"] = time.perf_counter() - a
            seg_runs.append(sg)
        sync()
    segments = None
    if seg_runs:
        # one key list on every rank (the receiving rank has legs the others lack): the all_reduce below needs equal shapes
        keys = ["gather_finish_previous", "stage1", "stage2", "mask_all_reduce", "gather", "gather.staging_d2h", "gather.collective",
                "gather.staging_h2d", "gather.unpack", "pass_total_synchronised"]
        assert all(k in keys for sg in seg_runs for k in sg)
        med = {k: float(np.median([sg.get(k, 0.0) for sg in seg_runs])) * 1e3 for k in keys}
        segments_min = None
        if use_dist:      # the slowest and the fastest rank's figure per segment: a straggler (link, GPU) shows as a gap
            tt = torch.tensor([med[k] for k in keys], dtype=torch.float64, device=device)
            if backend == "gloo":
                tt = tt.cpu()
            lo = tt.clone()
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            med = {k: float(v) for k, v in zip(keys, tt)}
            segments_min = {k: float(v) for k, v in zip(keys, lo)}
        segments = med

    # the BASELINE target configuration (conversion terms: T, u, v, omega) on the same resident fields, rank 0
    conv = None
    if rank == 0 and resident and with_q and not args.moving:
        h0, h1, f = held[:3]
        ev = []
        for i in range(2 + 5):
            tm = [] if i >= 2 else None
            eng.rowstats(f["tair"], f["u"], f["v"], f["omega"], None, [box], t_begin=t0 - h0, t_count=T_local, with_q=False, timing=tm, rows_out=rows)
            if tm:
                ev.extend(tm)
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        cbytes = 4 * nl * lat.size * lon.size * esz * T_local
        conv = {"achieved": cbytes / (ms * 1e-3) / 1e9, "frac": cbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": ms,
                "algorithmic_bytes_per_launch": cbytes, "timesteps_per_s": T_local / (ms * 1e-3),
                "kernel": "lec_rowsweep_kernel (T, u, v, omega; the fused conversion-terms configuration of BASELINE.json's target)"}

    # moving boxes: the shipped kernel family (box tiles) against the independent one-wave-per-row formulation on the first steps of the
    # resident shard -- record by record and term by term (finite is not a check: wrong coefficients give finite numbers too)
    moving_check = None
    if rank == 0 and resident and args.moving:
        # (a box-packed series: the check runs on the CROP the series was produced from, so it also says that packing changed no
        # value: the timed pass read the packed series, `a` reads the crop)
        h0, f = (crop_head["h0"], crop_head["f"]) if packed else held[:3:2]
        n = min(T_local, 8)
        bx = [all_boxes.boxes[i] for i in range(n)]
        nheld = f["tair"].shape[0]
        kw = dict(time_s=time_s[h0:h0 + nheld] if with_q else None, t_begin=t0 - h0, t_count=n, with_q=with_q, per_step_boxes=True, keep_rows=True)
        a = eng.compute(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], bx, **kw)
        b = eng.compute(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], bx, tuning={"kernel": "row_sweep"}, **kw)
        torch.cuda.synchronize()
        ra, rb = a.rows[..., :28], b.rows[..., :28]
        rec = float(((ra - rb).abs().amax(dim=(0, 1, 2)) / rb.abs().amax(dim=(0, 1, 2)).clamp_min(1e-300)).max())
        sc = float(((a.scalars - b.scalars).abs().amax(dim=0) / b.scalars.abs().amax(dim=0).clamp_min(1e-300)).max())
        same = bool(torch.equal(a.scalars, res.scalars[:n]))       # and the timed pass produced exactly these numbers
        worst_vs_timed = float(((a.scalars - res.scalars[:n]).abs().amax(dim=0) / a.scalars.abs().amax(dim=0).clamp_min(1e-300)).max())
        moving_check = {"steps": n, "row_records_max_rel_diff_vs_row_sweep": rec, "terms_max_rel_diff_vs_row_sweep": sc,
                        "timed_pass_bit_identical": same, "timed_pass_max_rel_diff": worst_vs_timed,
                        "ok": bool(rec <= 1e-10 and sc <= 1e-9 and worst_vs_timed <= 1e-11),
                        "checked_on": ("the track-extent crop the packed series was gathered from" if packed else "the resident crop")}

    out, cpu_leg = None, None
    if rank == 0:
        from lorenzcycletoolkit_amd._lib import source_digest
        finite = bool(torch.isfinite(res.scalars).all().item())
        gathered_ok = None
        if series is not None and gat.active:       # rank 0's block of the gathered series is what its own stage 2 wrote
            gathered_ok = bool(torch.equal(series[t0:t1], res.packed)) and tuple(series.shape) == (T_global, LECEngine.packed_width(nl))
        bytes_per_step_t = nfields * nl * lat.size * lon.size * esz        # algorithmic bytes per time step
        if args.moving:
            bytes_per_step_t = nfields * nl * 61 * 61 * esz                # only the box is read
        launch_ms = [a.elapsed_time(b) for (a, b) in kernel_ms]
        avg_launch_ms = float(np.mean(launch_ms))
        steps_per_launch = T_local if resident else float(np.mean([b - a for a, b in chunks]))
        achieved = bytes_per_step_t * steps_per_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic, traffic_src, traffic_sha = None, None, None
        csrc_sha = source_digest()
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc) and (args.ny, args.nx) == (721, 1440):
            try:
                key = (("rowstats_moving_packed_hbm_bytes_per_timestep" if packed else "rowstats_moving_hbm_bytes_per_timestep") if args.moving else
                       f"rowstats_{args.storage}_{'noq' if args.no_q else 'all'}_hbm_bytes_per_timestep")
                summ = json.load(open(pmc))
                per_t = summ.get(key)
                traffic = None if per_t is None else per_t * steps_per_launch
                traffic_src = summ.get(key.replace("_hbm_bytes_per_timestep", "_source"))
                traffic_sha = summ.get(key.replace("_hbm_bytes_per_timestep", "_csrc_sha"))
            except Exception:
                traffic = None
        if args.moving:
            kname = (("lec_boxplane_kernel (a box-packed series: one wave per four box rows x a chunk of levels, the planes' rows straight into the "
                      "layout the sums are taken in)" if (not args.nonuniform_lon and "box_tile" not in args.tuning)
                      else "lec_boxtile_kernel on a box-packed series") if packed else
                     "lec_boxtile_kernel (one wave per four box rows x a chunk of levels of a time step; six values per point transposed through LDS)")
        elif args.no_q or args.storage == "f32":
            kname = "lec_rowsweep_kernel (one wave per row)" + ("" if args.no_q else " + lec_qtime_kernel")
        else:
            kname = "one lec_rowstats call = lec_rowblock_kernel + lec_qtime_kernel"
        terms = "Az Ae Kz Ke Cz Ca Ck Ce BAz BAe BKz BKe (T,u,v,omega only)" if args.no_q else "all 16 (incl. BPhi, Gz, Ge)"
        if args.moving:
            workload = (f"synthetic 0.25-degree {nl} lev x {lat.size} x {lon.size} track-extent crop, moving 15x15-degree box "
                        f"(61 x 61 points) per time step, storage {args.storage}, terms = {terms}"
                        + ("; the series lies in HBM BOX-PACKED (every step's box gathered out of the crop and dT/dt formed by the PRODUCER, outside the "
                           "timed region -- config.producer_ms --, as the streamed moving framework's ingest writes it: 5 field slabs + "
                           + ("dT/dt as an fp64 cube" if args.storage == "f64" else "T of the two time neighbours")
                           + " per step; --moving-layout cube = the whole crop, rounds 1-4)" if packed else "; the series lies in HBM as the whole crop"))
        else:
            workload = (f"synthetic ERA5-res {nl} lev x {args.ny} x {args.nx}" + (" on STRETCHED longitudes" if args.nonuniform_lon else "") +
                        f", fixed box = whole grid, storage {args.storage}, terms = {terms}")
        workload += (f"; strong scaling: global series T={T_global} sharded over {world} GPU(s), "
                     + ("shard resident in HBM" if resident else f"streamed through HBM in chunks of {chunk} steps (+ one-step T halo)")
                     if strong else f"; T={T_local} per GPU resident in HBM")
        # which configuration of BASELINE.json this line is (configs are numbered from 1 as the prompt's "configs[1]" numbers from 0: 3 = T=64 on
        # one GPU, 4 = T=2048 sharded over 2/4/8 GPUs, 5 = moving box, T=4096 on 8 GPUs)
        full_grid = (args.ny, args.nx) == (721, 1440) and not args.nonuniform_lon and args.storage == "f64" and not args.no_q
        if args.moving:
            bcfg = {"id": 5, "exact": bool(T_global == 4096 and world == 8 and args.storage == "f64" and not args.no_q),
                    "note": f"moving box, T={T_global} on {world} GPU(s); BASELINE config 5 is T=4096 on 8 GPUs (--gpus 8 --moving --timesteps-global 4096)"}
        elif not full_grid:
            bcfg = {"id": None, "exact": False, "note": "a variant of the headline workload (grid, storage or term set differ): not a BASELINE configuration"}
        elif world == 1 and not strong:
            bcfg = {"id": 3, "exact": bool(T_global == 64), "note": f"T={T_global} resident on one GPU; BASELINE config 3 is T=64"}
        else:
            bcfg = {"id": 4, "exact": bool(T_global == 2048 and world in (2, 4, 8)),
                    "note": (f"T={T_global} time-sharded over {world} GPU(s)" + (" (strong scaling)" if strong else f" (weak scaling: T={T_local} per GPU)")
                             + "; BASELINE config 4 is T=2048 over 2/4/8 GPUs (--timesteps-global 2048)")}
        if resident:
            timed_region = "wall clock around all passes (stage 1 + stage 2 + collectives, every pass's series delivered to rank 0), inputs resident in HBM"
        else:
            timed_region = "lec_rowstats per chunk + lec_reduce + collectives (synchronised segments); synthetic generation excluded"
        if packed:
            timed_region += ("; the series is resident BOX-PACKED: the per-step slice (box_data.py:297-310) and dT/dt (lorenzcycletoolkit.py:184-186) -- both "
                             "inside the reference's own clock -- are the producer's work and OUTSIDE this region: config.producer_ms times them, "
                             "config.value_incl_producer is the rate with them inside")
        elif args.moving:
            timed_region += "; the kernel slices every step's box by index and forms dT/dt per point: both inside this region"
        out = {
            "metric": ("LEC timesteps/sec (all terms), moving 61x61x37 box per time step" if args.moving else
                       "LEC timesteps/sec (all energy+conversion+boundary+generation terms) at 37x721x1440"),
            "value": T_global * args.steps / elapsed,
            "unit": "timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": workload, "baseline_config": bcfg, **({"moving_layout": "packed" if packed else "cube"} if args.moving else {}),
                "timesteps_per_gpu": T_local, "timesteps_global": T_global, "world_size": world,
                "backend": (dist.get_backend() if use_dist else "none"),
                "parallelism": f"time-sharded x{world}, no data-path collective; RCCL all_reduce of the NaN-level mask (fixed box) + one gather of the "
                               "packed per-time-step records to rank 0 (a send per peer, each over its own xGMI link), double-buffered: "
                               "the gather of pass i overlaps the kernels of pass i + 1",
                "timed_region": timed_region,
                "results_finite": finite, "csrc_sha": csrc_sha,
                **({"gathered_series_ok": gathered_ok} if gathered_ok is not None else {}),
                **({"peer_blocks_ok": peers["peer_blocks_ok"], "peer_blocks": peers["blocks_ok"],
                    "rccl_ranks_seen": who["ranks_seen"], "devices_distinct": who["devices_distinct"], "rank_devices": who["devices"],
                    "ms_per_step_per_rank": per_rank} if peers is not None else {}),
                **({"tuning": args.tuning} if args.tuning else {}),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": (None if traffic is None else f"{traffic_src}: rocprofv3 --pmc FETCH_SIZE pass of this command on an MI355X box, "
                                   "corrected per MI355X_MICROARCH.md, per launch; a stored measurement, not re-measured in this run"),
                # the stored counter pass was taken on kernel sources with this digest; stale = the kernels have changed since
                "traffic_csrc_sha": traffic_sha, "traffic_stale": (None if traffic is None else bool(traffic_sha != csrc_sha)),
                "kernel": kname, "avg_launch_ms": avg_launch_ms,
                "algorithmic_bytes_per_launch": bytes_per_step_t * steps_per_launch,
            },
        }
        if producer is not None:
            out["config"]["producer_ms"] = producer
            out["config"]["value_incl_producer"] = T_global * args.steps / (elapsed + args.steps * producer["total"] * 1e-3)
            out["config"]["ms_per_step_incl_producer"] = elapsed / args.steps * 1e3 + producer["total"]
        if segments is not None:
            k1 = avg_launch_ms
            pass_ms = elapsed / args.steps * 1e3
            segments["pass_timed_unsynchronised"] = pass_ms
            segments["stage1_kernels_hip_events"] = k1
            # what a pass costs beyond its kernels, in the timed (pipelined) loop: host gaps + whatever of the collective is not hidden
            segments["fixed_cost_per_pass"] = pass_ms - k1 - segments.get("stage2", 0.0)
            out["config"]["segments_ms"] = segments
            if segments_min is not None:
                out["config"]["segments_ms_min_over_ranks"] = segments_min
            out["config"]["segments_note"] = ("median of 3 instrumented passes (a device synchronisation closes every segment; max over ranks); "
                                              "stage2 includes its launch and host time; fixed_cost_per_pass = pass_timed_unsynchronised - "
                                              "stage1_kernels_hip_events - stage2")
        if moving_check is not None:
            out["config"]["moving_check"] = moving_check
        if not resident:
            out["config"]["chunk"] = chunk
            out["config"]["wall_ms_per_step_incl_generation"] = wall / args.steps * 1e3
            out["config"]["generation_ms_per_step"] = gen_s[0] / args.steps * 1e3
        if conv is not None:
            out["roofline"]["conversion_terms"] = conv
        # the synthetic fields are seeded per GLOBAL time step and the kernels are bitwise reproducible under sharding and chunking,
        # so the series does not depend on N: compare it with the digest a one-GPU run stored (profiles/series_digests.json)
        dkey = (f"{'moving' if args.moving else 'fixed'}_{args.storage}_{'noq' if args.no_q else 'all'}_{nl}x{lat.size}x{lon.size}_T{T_global}"
                + ("_stretched" if args.nonuniform_lon else ""))
        full = series if (series is not None and gat.active) else res.packed
        sums = record_checksums(full).cpu().numpy() if full.shape[0] == T_global else None
        if sums is not None:
            import hashlib
            digest = {"sha256": hashlib.sha256(sums.tobytes()).hexdigest(), "steps": int(T_global),
                      "per_step": "".join("%016x" % (int(x) & 0xFFFFFFFFFFFFFFFF) for x in sums)}
            try:
                stored = json.load(open(args.digest_file)).get(dkey)
            except Exception:
                stored = None
            out["config"]["series_digest_key"] = dkey
            out["config"]["series_sha256"] = digest["sha256"]
            out["config"]["series_equals_n1"] = None if stored is None else bool(stored["sha256"] == digest["sha256"])
            if stored is not None and stored["sha256"] != digest["sha256"] and len(stored.get("per_step", "")) == 16 * T_global:
                bad = [i for i in range(T_global) if stored["per_step"][16 * i:16 * i + 16] != digest["per_step"][16 * i:16 * i + 16]]
                out["config"]["series_steps_differing_from_n1"] = {"count": len(bad), "first": bad[:8]}
            if args.write_digest and world == 1:
                try:
                    book = json.load(open(args.digest_file))
                except Exception:
                    book = {"_note": "per-step checksums (parallel.record_checksums) of the packed series of one-GPU bench.py runs, keyed by "
                                     "configuration and global series length; bench.py compares an N-GPU run's gathered series with them "
                                     "(config.series_equals_n1)"}
                book[dkey] = digest
                os.makedirs(os.path.dirname(os.path.abspath(args.digest_file)), exist_ok=True)
                json.dump(book, open(args.digest_file, "w"), indent=1)
        if strong:
            # the N = 1 value of this configuration (profiles/strong_scaling_n1.json, keyed by the series' layout, stamped with the kernel
            # sources it was measured on; tools/update_n1.py writes it from one-GPU lines).  A one-GPU run IS the N = 1 value.
            nkey = n1_key(args, T_global, packed)
            try:
                n1 = json.load(open(args.n1_file)).get(nkey)
            except Exception:
                n1 = None
            n1 = n1 if isinstance(n1, dict) else None
            cfg = out["config"]
            cfg["n1_key"] = nkey
            cfg["n1_stored"] = None if n1 is None else {k: n1.get(k) for k in ("value", "csrc_sha", "source")}
            cfg["n1_stale"] = None if n1 is None else bool(n1.get("csrc_sha") != csrc_sha)
            if world == 1:
                cfg["speedup_vs_n1"], cfg["n1_value"] = 1.0, out["value"]
            else:
                cfg["speedup_vs_n1"] = None if n1 is None else out["value"] / n1["value"]
                cfg["n1_value"] = None if n1 is None else n1["value"]
                if producer is not None and n1 is not None and n1.get("value_incl_producer"):
                    cfg["speedup_vs_n1_incl_producer"] = cfg["value_incl_producer"] / n1["value_incl_producer"]

        # The CPU leg runs AFTER the timed region and the verification collectives, on rank 0 only; at N > 1 the peers wait for it in the
        # closing barrier (17 s for "single"; the process group's timeout is 10 min for RCCL, 30 for gloo).  north_star: the N-GPU
        # throughput "next to the reference CPU path timed on the node's own host cores in the same run" (the reference clocks its whole
        # call: lorenzcycletoolkit.py:173,180,199).  `parity` at N > 1 is rank 0's shard: its first steps against the oracle.
        if cpu_kind != "none" and with_q:
            leg = cpu_kind if world == 1 or cpu_kind == "quick" else "single"
            if args.moving:
                # the first steps of the crop rank 0's series came from (a streamed run: generated again -- seeded per global step)
                n = cpu_leg_steps_moving(leg, T_local)
                hold = min(n + 1, T_global)
                if packed and resident:
                    keep = {k: v[:hold] for k, v in crop_head["f"].items()}
                elif resident:
                    keep = {k: v[:hold].clone() for k, v in held[2].items()}
                else:
                    keep = synthetic_cube(hold, level, lat, lon, device=device, dtype=tdtype, seed=1234, t0_global=0)
                got_s = {k: v[:n] for k, v in res.scalars_dict().items()}
                got_l = {k: v[:n] for k, v in res.levels_dict().items()}
                lims = limits_of(0, hold)

                def cpu_leg():
                    out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity_moving(leg, keep, lims, lat, lon, level, time_s, got_s, got_l, n)
            else:
                n = min(cpu_leg_steps(leg), T_local)
                if resident:
                    keep = {k: (None if v is None else v[:n].clone()) for k, v in held[2].items()}
                else:
                    keep = synthetic_cube(n, level, lat, lon, device=device, dtype=tdtype, seed=1234, t0_global=0)
                note = ("the first %d time steps of the resident synthetic cube the GPU was timed on, copied to the host" if resident else
                        "the first %d time steps of the first chunk of the streamed series (generated again from the per-step seeds), copied to the host")

                def cpu_leg():
                    out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(leg, eng, keep, lat, lon, level, time_s[:n], device, note)
            if world > 1:
                inner = cpu_leg

                def cpu_leg():
                    inner()
                    out["cpu_baseline"]["ranks_waiting_in_the_closing_barrier"] = world - 1
                    out["parity"]["shard"] = f"rank 0 of {world}: global time steps {t0}..{t1 - 1}"
    return out, cpu_leg


def strong_legs(args, ctx):
    """BASELINE configs 4 and 5 inside the default N > 1 run (the driver passes --gpus N and nothing else): the fixed box at T = 2048
    (sharded, streamed through HBM in chunks) and the moving box at T = 4096, one warm-up pass and --leg-steps timed passes each, on
    the job's own process group.  Returns rank 0's summary for config.strong_scaling (None elsewhere)."""
    import copy
    import gc
    import torch
    t4, t5 = (int(x) for x in args.leg_timesteps.split(","))
    legs = {}
    for name, T, moving in (("config4", t4, False), ("config5", t5, True)):
        la = copy.copy(args)
        la.timesteps_global, la.moving, la.steps, la.warmup, la.chunk = T, moving, args.leg_steps, 1, 0
        la.write_digest = False
        tic = time.perf_counter()
        o, _ = measure(la, ctx, cpu_kind="none")
        gc.collect()
        torch.cuda.empty_cache()
        if o is None:
            continue
        c = o["config"]
        legs[name] = {"value": o["value"], "unit": "timesteps/s", "ms_per_pass": o["ms_per_step"], "timesteps_global": T, "passes": la.steps,
                      "speedup_vs_n1": c.get("speedup_vs_n1"), "n1_value": c.get("n1_value"), "n1_stale": c.get("n1_stale"), "n1_key": c.get("n1_key"),
                      "per_gpu_roofline_frac": o["roofline"]["frac"], "series_equals_n1": c.get("series_equals_n1"),
                      "baseline_config": c["baseline_config"], "workload": c["workload"], "timed_region": c["timed_region"],
                      "results_finite": c["results_finite"], "peer_blocks_ok": c.get("peer_blocks_ok"),
                      "leg_wall_s": time.perf_counter() - tic}
        for k in ("chunk", "generation_ms_per_step", "producer_ms", "value_incl_producer", "speedup_vs_n1_incl_producer", "moving_layout", "moving_check"):
            if k in c:
                legs[name][k] = c[k]
    return legs or None


def run_rank(args):
    import gc
    import torch
    import torch.distributed as dist
    ctx = RankContext(args)
    kind = "none" if args.no_cpu_baseline else args.cpu_baseline
    out, cpu_leg = measure(args, ctx, cpu_kind=kind)
    # the default N > 1 line: BASELINE configs 4 and 5 as two short strong-scaling legs, after the headline and before the CPU leg
    default_line = not (args.moving or args.timesteps_global > 0 or args.no_q or args.storage != "f64" or args.nonuniform_lon or args.tuning)
    if ctx.world > 1 and default_line and not args.no_strong_legs:
        gc.collect()
        torch.cuda.empty_cache()                # the headline's resident cube makes room for the legs' chunks
        legs = strong_legs(args, ctx)
        if out is not None:
            out["config"]["strong_scaling"] = legs
            out["config"]["strong_scaling_note"] = ("BASELINE configs 4 (fixed box) and 5 (moving box) as strong-scaling legs of this job, one warm-up and "
                                                    f"{args.leg_steps} timed passes each; speedup_vs_n1 divides by the stored one-GPU value of the same "
                                                    "configuration and layout (n1_stale: the kernel sources have changed since it was measured)")
    if ctx.rank == 0:
        if cpu_leg is not None:
            cpu_leg()
        ctx.print_line(out)
    if ctx.use_dist:
        ctx.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus > 1:
            sys.exit(launch_ranks(args))
    elif int(env_world) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} contradicts WORLD_SIZE={env_world}: launch as `python bench.py --gpus N` (this script starts "
              f"the ranks) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`", file=sys.stderr)
        sys.exit(2)
    run_rank(args)


if __name__ == "__main__":
    main()
