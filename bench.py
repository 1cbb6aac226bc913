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
all_gather of the per-time-step results per pass.  Prints ONE JSON line on rank 0.
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
    ap.add_argument("--tuning", type=str, default="", help="A/B runs: lec_tuning fields, e.g. kernel=row_sweep,tile_t=4 (default: the library's choice)")
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
# CPU baseline: the NumPy oracle (the reference's eager op sequence) on the GPU box's host cores
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_band(nt, ny, seed):
    """Seconds the oracle needs for `nt` time steps of a 37 x ny x 1440 latitude band of the benchmark grid (all 16 terms)."""
    from oracle import lec_oracle as o
    from lorenzcycletoolkit_amd.synthetic import era5_like_levels
    rng = np.random.default_rng(seed)
    nx = 1440
    level = era5_like_levels()
    lat = (np.linspace(-90.0, 90.0, 721) if ny == 721 else -45.0 + 0.25 * np.arange(ny))
    lon = np.linspace(-180.0, 179.75, nx)
    p = level[None, :, None, None]
    phi, lam = np.deg2rad(lat)[None, None, :, None], np.deg2rad(lon)[None, None, None, :]
    shp = (nt, level.size, ny, nx)
    T = 288.0 * (p / 1e5) ** 0.19 + 10.0 * np.cos(2 * phi) * (p / 1e5) + rng.standard_normal(shp)
    u = 25.0 * np.cos(phi) * (1 - p / 1.2e5) + 5.0 * rng.standard_normal(shp)
    v = 3.0 * np.sin(2 * lam) * np.cos(phi) + 3.0 * rng.standard_normal(shp)
    w = 0.05 * np.sin(3 * lam) * np.cos(phi) + 0.1 * rng.standard_normal(shp)
    ph = o.G * 7000.0 * np.log(1e5 / p) + 100.0 * rng.standard_normal(shp)
    dom = o.Domain(T, u, v, w, ph, lat, lon, level, np.arange(nt) * 3600.0)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):          # the polar rows divide by cos(90 deg) in the reference too
        o.lec_fixed(dom, lon[0], lon[-1], lat[0], lat[-1])
    return time.perf_counter() - t0


def _oracle_band_worker(a):
    return _oracle_band(*a)


def cpu_baseline(kind):
    """kind "full": (a) ONE thread, like the reference: 3 repetitions of 2 time steps at the full 37 x 721 x 1440, median;
    (b) all host cores this process may use (at most 16): one worker process per core, each 2 time steps of a 181-row band,
    scaled by 721/181 -- a best-effort figure (memory bound: the oracle materialises every 4-D temporary like the reference).
    kind "quick": 2 steps of a 31-row band, one thread (contract tests)."""
    import multiprocessing as mp
    ncpu = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = ncpu
    if kind == "quick":
        dt = _oracle_band(2, 31, 1)
        return {"value": 2 / (dt * 721 / 31), "unit": "timesteps/s", "cores": 1, "kind": "port", "host_cpu_count": ncpu,
                "sample": f"NumPy fp64 oracle, 1 thread, 2 time steps of a 37x31x1440 band in {dt:.2f} s, scaled by 721/31 (quick mode)"}
    reps = [_oracle_band(2, 721, 10 + r) for r in range(3)]
    med = float(np.median(reps))
    out = {"value": 2 / med, "unit": "timesteps/s", "cores": 1, "kind": "port", "host_cpu_count": ncpu, "usable_cores": usable,
           "seconds_per_timestep": med / 2,
           "sample": f"NumPy fp64 oracle (the reference's eager op order), 1 thread, 2 time steps at the full 37x721x1440, "
                     f"3 repetitions: {', '.join(f'{r:.1f}' for r in reps)} s, median"}
    workers = max(1, min(usable, 16))
    try:
        ctx = mp.get_context("spawn")
        t0 = time.perf_counter()
        with ctx.Pool(workers) as pool:
            pool.map(_oracle_band_worker, [(2, 181, 100 + i) for i in range(workers)])
        wall = time.perf_counter() - t0
        out["all_cores"] = {"value": workers * 2 * (181.0 / 721.0) / wall, "unit": "timesteps/s", "cores": workers,
                            "sample": f"{workers} worker processes (one per usable core, capped at 16), each 2 time steps of a "
                                      f"37x181x1440 band, in {wall:.1f} s wall incl. process start, scaled by 181/721"}
    except Exception as e:      # a host that cannot fork workers still reports the one-thread figure
        out["all_cores"] = {"value": None, "error": repr(e)}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("LEC_DIST_BACKEND", "nccl")   # nccl == RCCL on ROCm; gloo only rehearses the N > 1 path on one GPU
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible: the HIP path is the only path")
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s): RCCL needs one GPU per rank")
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)      # the rank's GPU is already current

    from lorenzcycletoolkit_amd.engine import LECEngine
    from lorenzcycletoolkit_amd.parallel import gather_result, halo_range, merge_dropmask, shard_range
    from lorenzcycletoolkit_amd.synthetic import era5_like_levels, synthetic_cube

    level = era5_like_levels()
    lat = np.linspace(-90.0, 90.0, args.ny)
    lon = np.linspace(-180.0, 180.0 - 360.0 / args.nx, args.nx)
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

    def boxes_of(a, b):
        if not args.moving:
            return [box]
        tg = np.arange(a, b)
        clat = -37.5 + 12.0 * np.sin(2 * np.pi * tg / 400.0)
        clon = -50.0 + 22.0 * np.cos(2 * np.pi * tg / 700.0)
        return [eng.box_from_limits(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(clat, clon)]

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
    if args.moving:        # thousands of boxes: build and upload their tables once (PreparedBoxes), not at every call
        chunk_boxes = {c: eng.prepare_boxes(bx, nyb_min=nyb_max) for c, bx in chunk_boxes.items()}
        all_boxes = eng.prepare_boxes(all_boxes, nyb_min=nyb_max)

    def generate(a, b):
        h0, h1 = halo_range(a, b, T_global)                  # one-step halo for dT/dt (thermodynamics.py:109-110)
        f = synthetic_cube(h1 - h0, level, lat, lon, device=device, dtype=tdtype, seed=1234, t0_global=h0)
        if args.no_q:
            f["geopt"] = None
        return h0, h1, f

    held = generate(*chunks[0]) if resident else None
    rows = None if resident else torch.empty((T_local, nl, nyb_max, 32), dtype=torch.float64, device=device)
    merge = (lambda m: merge_dropmask(m)) if (world > 1 and not args.moving) else None
    kernel_ms = []
    gen_s = [0.0]
    tuning = None
    if args.tuning:
        tuning = {k: (v if k in ("kernel", "order") else int(v)) for k, v in (kv.split("=") for kv in args.tuning.split(","))}
    stage1 = dict(with_q=with_q, tuning=tuning, per_step_boxes=bool(args.moving))

    def barrier():
        if backend == "nccl":
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()

    def sync(with_barrier=True):
        if world > 1 and with_barrier:
            barrier()
        torch.cuda.synchronize()

    def one_pass(record):
        """One pass over this rank's shard; returns (result, seconds inside the timed region)."""
        timed = 0.0
        if resident:
            h0, h1, f = held
            timing = [] if record else None
            r = eng.rowstats(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], all_boxes, time_s=time_s[h0:h1] if with_q else None,
                             t_begin=t0 - h0, t_count=T_local, timing=timing, **stage1)
            res = eng.reduce(r, all_boxes, merge_dropmask=merge, drop_any_time=not args.moving)
            if world > 1:
                gather_result(res, T_global)      # the job's only collective besides the mask (RCCL all_gather over xGMI)
            if record:
                kernel_ms.extend(timing)
            return res, None                      # resident passes are timed by the caller's wall clock around all K of them
        for (a, b) in chunks:
            g0 = time.perf_counter()
            h0, h1, f = generate(a, b)
            torch.cuda.synchronize()
            gen_s[0] += time.perf_counter() - g0
            timing = [] if record else None
            tic = time.perf_counter()
            eng.rowstats(f["tair"], f["u"], f["v"], f["omega"], f["geopt"], chunk_boxes[(a, b)], time_s=time_s[h0:h1] if with_q else None,
                         t_begin=a - h0, t_count=b - a, timing=timing, rows_out=rows[a - t0:b - t0], **stage1)
            torch.cuda.synchronize()
            timed += time.perf_counter() - tic
            if record:
                kernel_ms.extend(timing)
            del f
        sync()
        tic = time.perf_counter()
        res = eng.reduce(rows, all_boxes, merge_dropmask=merge, drop_any_time=not args.moving)
        if world > 1:
            gather_result(res, T_global)
        sync()
        timed += time.perf_counter() - tic
        return res, timed

    for _ in range(args.warmup):
        one_pass(False)
    sync()
    gen_s[0] = 0.0
    tic = time.perf_counter()
    timed_total = 0.0
    for _ in range(args.steps):
        res, timed = one_pass(True)
        if timed is not None:
            timed_total += timed
    sync()
    wall = time.perf_counter() - tic
    elapsed = wall if resident else timed_total
    if world > 1:
        el = torch.tensor([elapsed, wall], dtype=torch.float64, device=device)
        if backend == "gloo":
            el = el.cpu()
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed, wall = float(el[0]), float(el[1])

    # the BASELINE target configuration (conversion terms: T, u, v, omega) on the same resident fields, rank 0
    conv = None
    if rank == 0 and resident and with_q and not args.moving:
        h0, h1, f = held
        ev = []
        for i in range(2 + 5):
            tm = [] if i >= 2 else None
            eng.rowstats(f["tair"], f["u"], f["v"], f["omega"], None, [box], t_begin=t0 - h0, t_count=T_local, with_q=False, timing=tm)
            if tm:
                ev.extend(tm)
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        cbytes = 4 * nl * lat.size * lon.size * esz * T_local
        conv = {"achieved": cbytes / (ms * 1e-3) / 1e9, "frac": cbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": ms,
                "algorithmic_bytes_per_launch": cbytes, "timesteps_per_s": T_local / (ms * 1e-3),
                "kernel": "lec_rowsweep_kernel (T, u, v, omega; the fused conversion-terms configuration of BASELINE.json's target)"}

    if rank == 0:
        finite = bool(torch.isfinite(res.scalars).all().item())
        bytes_per_step_t = nfields * nl * lat.size * lon.size * esz        # algorithmic bytes per time step
        if args.moving:
            bytes_per_step_t = nfields * nl * 61 * 61 * esz                # only the box is read
        launch_ms = [a.elapsed_time(b) for (a, b) in kernel_ms]
        avg_launch_ms = float(np.mean(launch_ms))
        steps_per_launch = T_local if resident else float(np.mean([b - a for a, b in chunks]))
        achieved = bytes_per_step_t * steps_per_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc) and (args.ny, args.nx) == (721, 1440):
            try:
                key = ("rowstats_moving_hbm_bytes_per_timestep" if args.moving else
                       f"rowstats_{args.storage}_{'noq' if args.no_q else 'all'}_hbm_bytes_per_timestep")
                per_t = json.load(open(pmc)).get(key)
                traffic = None if per_t is None else per_t * steps_per_launch
            except Exception:
                traffic = None
        if args.moving:
            kname = "lec_boxtile_kernel (one wave per four box rows x ten levels of a time step; six values per point transposed through LDS)"
        elif args.no_q or args.storage == "f32":
            kname = "lec_rowsweep_kernel (one wave per row)" + ("" if args.no_q else " + lec_qtime_kernel")
        else:
            kname = "one lec_rowstats call = lec_rowblock_kernel + lec_qtime_kernel"
        terms = "Az Ae Kz Ke Cz Ca Ck Ce BAz BAe BKz BKe (T,u,v,omega only)" if args.no_q else "all 16 (incl. BPhi, Gz, Ge)"
        if args.moving:
            workload = (f"synthetic 0.25-degree {nl} lev x {lat.size} x {lon.size} track-extent crop, moving 15x15-degree box "
                        f"(61 x 61 points) per time step, storage {args.storage}, terms = {terms}")
        else:
            workload = (f"synthetic ERA5-res {nl} lev x {args.ny} x {args.nx}, fixed box = whole grid, storage {args.storage}, terms = {terms}")
        workload += (f"; strong scaling: global series T={T_global} sharded over {world} GPU(s), "
                     + ("shard resident in HBM" if resident else f"streamed through HBM in chunks of {chunk} steps (+ one-step T halo)")
                     if strong else f"; T={T_local} per GPU resident in HBM")
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
                "workload": workload,
                "timesteps_per_gpu": T_local, "timesteps_global": T_global, "world_size": world,
                "backend": (dist.get_backend() if world > 1 else "none"),
                "parallelism": f"time-sharded x{world}, no data-path collective; RCCL all_reduce of the NaN-level mask + all_gather of per-time-step results",
                "timed_region": ("wall clock around all passes, inputs resident in HBM" if resident else
                                 "lec_rowstats per chunk + lec_reduce + collectives (synchronised segments); synthetic generation excluded"),
                "results_finite": finite,
                **({"tuning": args.tuning} if args.tuning else {}),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": kname, "avg_launch_ms": avg_launch_ms,
                "algorithmic_bytes_per_launch": bytes_per_step_t * steps_per_launch,
            },
        }
        if not resident:
            out["config"]["chunk"] = chunk
            out["config"]["wall_ms_per_step_incl_generation"] = wall / args.steps * 1e3
            out["config"]["generation_ms_per_step"] = gen_s[0] / args.steps * 1e3
        if conv is not None:
            out["roofline"]["conversion_terms"] = conv
        if strong:
            ref = os.path.join(ROOT, "profiles", "strong_scaling_n1.json")
            try:
                n1 = json.load(open(ref)).get(f"{'moving' if args.moving else 'fixed'}_{args.storage}_{'noq' if args.no_q else 'all'}_T{T_global}")
            except Exception:
                n1 = None
            out["config"]["speedup_vs_n1"] = None if not n1 else out["value"] / n1
            out["config"]["n1_value"] = n1
        kind = "none" if args.no_cpu_baseline else args.cpu_baseline
        if world == 1 and kind != "none":
            out["cpu_baseline"] = cpu_baseline(kind)
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if world > 1:
        barrier()
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
