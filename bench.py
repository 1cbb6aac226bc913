#!/usr/bin/env python3
"""bench.py -- LEC time steps per second on synthetic 37 x 721 x 1440 fields (BASELINE.json metric).

One "step" = one pass of the whole hot path (lec_rowstats + lec_reduce, every energy, conversion,
boundary and generation term) over the batch of time steps resident in this rank's HBM.  Inputs are
generated on the device before the timed region.  N > 1: one process per GPU (torch.distributed,
backend nccl = RCCL), time steps sharded contiguously with a one-step halo for dT/dt generated
locally, one all_gather of the per-time-step results per step (weak scaling: --timesteps per GPU).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--timesteps", type=int, default=64, help="time steps resident per GPU (BASELINE config 3: T=64)")
    ap.add_argument("--storage", choices=["f64", "f32"], default="f64", help="storage dtype of the field cubes")
    ap.add_argument("--no-q", action="store_true", help="conversion-terms configuration: T,u,v,omega only (no Q, no Phi)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--moving", action="store_true", help="semi-Lagrangian configuration: one 15x15 degree box per time step "
                    "on a track-extent crop of the 0.25 degree grid (BASELINE config 5)")
    ap.add_argument("--ny", type=int, default=721)
    ap.add_argument("--nx", type=int, default=1440)
    return ap.parse_args()


def cpu_baseline():
    """The NumPy oracle (single thread, like the reference) on a bounded sample of the same workload:
    3 time steps of a 361-row latitude band of the 37 x 721 x 1440 grid, scaled by 721/361 (10-20 s of CPU work)."""
    from oracle import lec_oracle as o
    from lorenzcycletoolkit_amd.synthetic import era5_like_levels
    rng = np.random.default_rng(1234)
    nt, ny, nx = 3, 361, 1440
    level = era5_like_levels()
    lat = -45.0 + 0.25 * np.arange(ny)
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
    o.lec_fixed(dom, lon[0], lon[-1], lat[0], lat[-1])
    dt = time.perf_counter() - t0
    s_per_step_full = dt / nt * (721.0 / ny)
    return {
        "value": 1.0 / s_per_step_full, "unit": "timesteps/s", "cores": 1, "kind": "port",
        "sample": f"NumPy fp64 oracle, {nt} time steps of a 37x{ny}x{nx} latitude band in {dt:.1f} s, scaled by 721/{ny}",
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl == RCCL on ROCm; LEC_DIST_BACKEND=gloo is only for rehearsing the N > 1 path on a 1-GPU box
        dist.init_process_group(backend=os.environ.get("LEC_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from lorenzcycletoolkit_amd.engine import LECEngine
    from lorenzcycletoolkit_amd.parallel import compute_shard, gather_result, halo_range, shard_range
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

    # weak scaling: `timesteps` per GPU; global series of world * timesteps steps, contiguous shards
    T_local = args.timesteps
    T_global = T_local * world
    if T_global < 2 and not args.no_q:
        raise SystemExit("bench.py: the diabatic-heating terms differentiate T in time: need at least 2 time steps in all (or --no-q)")
    t0, t1 = shard_range(T_global, world, rank)
    h0, h1 = halo_range(t0, t1, T_global)                    # one-step halo for dT/dt (thermodynamics.py:109-110)
    fields = synthetic_cube(h1 - h0, level, lat, lon, device=device, dtype=tdtype, seed=1234, t0_global=h0)
    time_s = np.arange(T_global) * 3600.0
    eng = LECEngine(lat, lon, level, device=device)
    box = eng.box_from_limits(lon[0], lon[-1], lat[0], lat[-1])
    boxes = None
    if args.moving:
        tg = np.arange(t0, t1)
        clat = -37.5 + 12.0 * np.sin(2 * np.pi * tg / 400.0)
        clon = -50.0 + 22.0 * np.cos(2 * np.pi * tg / 700.0)
        boxes = [eng.box_from_limits(lo - 7.5, lo + 7.5, la - 7.5, la + 7.5) for la, lo in zip(clat, clon)]
    with_q = not args.no_q
    if args.no_q:
        fields = dict(fields, geopt=None)

    kernel_ms = []

    def step(record=False):
        timing = [] if record else None
        if boxes is not None:
            res = eng.compute(fields["tair"], fields["u"], fields["v"], fields["omega"], fields["geopt"], boxes,
                              time_s=time_s[h0:h1] if with_q else None, t_begin=t0 - h0, t_count=t1 - t0,
                              with_q=with_q, timing=timing)
        else:
            res = compute_shard(eng, fields, time_s, T_global, world, rank, box, with_q=with_q, timing=timing)
        if world > 1:
            gather_result(res, T_global)      # the job's only collective (RCCL all_gather over xGMI)
        if record:
            kernel_ms.append(timing)
        return res

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    tic = time.perf_counter()
    for _ in range(args.steps):
        res = step(record=True)
    sync()
    elapsed = time.perf_counter() - tic
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    if rank == 0:
        finite = bool(torch.isfinite(res.scalars).all().item())
        nfields = 4 if args.no_q else 5
        bytes_per_step_t = nfields * nl * lat.size * lon.size * esz        # algorithmic bytes per time step
        if args.moving:
            bytes_per_step_t = nfields * nl * 61 * 61 * esz                # only the box is read
        launch_ms = [a.elapsed_time(b) for pair in kernel_ms for (a, b) in pair]
        avg_launch_ms = float(np.mean(launch_ms))
        achieved = bytes_per_step_t * T_local / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc) and not args.moving and (args.ny, args.nx) == (721, 1440):
            try:
                per_t = json.load(open(pmc)).get(f"rowstats_{args.storage}_{'noq' if args.no_q else 'all'}_hbm_bytes_per_timestep")
                traffic = None if per_t is None else per_t * T_local
            except Exception:
                traffic = None
        out = {
            "metric": ("LEC timesteps/sec (all terms), moving 61x61x37 box per time step" if args.moving else
                       "LEC timesteps/sec (all energy+conversion+boundary+generation terms) at 37x721x1440"),
            "value": T_global * args.steps / elapsed,
            "unit": "timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": (f"synthetic 0.25-degree {nl} lev x {lat.size} x {lon.size} track-extent crop, T={T_local} per GPU, "
                             f"moving 15x15-degree box (61 x 61 points) per time step, storage {args.storage}, terms = "
                             if args.moving else
                             f"synthetic ERA5-res {nl} lev x {args.ny} x {args.nx}, T={T_local} per GPU resident in HBM, "
                             f"fixed box = whole grid, storage {args.storage}, terms = ")
                            + ("Az Ae Kz Ke Cz Ca Ck Ce BAz BAe BKz BKe (T,u,v,omega only)" if args.no_q else "all 16 (incl. BPhi, Gz, Ge)"),
                "timesteps_per_gpu": T_local, "timesteps_global": T_global,
                "parallelism": f"time-sharded x{world}, RCCL all_reduce of the NaN-level mask + all_gather of per-time-step results",
                "results_finite": finite,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "one lec_rowstats call = lec_rowblock_kernel + lec_qtime_kernel (lec_rowsweep_kernel for other configurations; lec_rowstats_kernel with LEC_KERNEL=0)", "avg_launch_ms": avg_launch_ms,
                "algorithmic_bytes_per_launch": bytes_per_step_t * T_local,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out, ensure_ascii=False))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
