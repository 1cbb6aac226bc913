#!/usr/bin/env python3
"""Where the streamed ingest's time goes: host staging (page cache / host memory -> pinned) against the H2D copies, per chunk.

Replays the copy side of ingest.lec_streamed on synthetic host arrays of the headline shape with time stamps on both sides:
host perf_counter around every stage() call, HIP events around every chunk's uploads.  Prints one line per chunk and the
overlap summary.  A measurement tool (GPU box), not product code.

  python tools/ingest_timeline.py --src i16 --timesteps 16 --chunk 4 [--threads 16] [--mode staged|registered|pageable]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", choices=["i16", "f32", "f64"], default="i16")
    ap.add_argument("--timesteps", type=int, default=16)
    ap.add_argument("--chunk", type=int, default=4)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--slots", type=int, default=2)
    ap.add_argument("--mode", choices=["staged", "registered", "pageable"], default="staged")
    ap.add_argument("--nl", type=int, default=37)
    ap.add_argument("--ny", type=int, default=721)
    ap.add_argument("--nx", type=int, default=1440)
    args = ap.parse_args()
    import torch
    from concurrent.futures import ThreadPoolExecutor

    dev = torch.device("cuda:0")
    npdt = {"i16": np.int16, "f32": np.float32, "f64": np.float64}[args.src]
    T, C = args.timesteps, args.chunk
    step_elems = args.nl * args.ny * args.nx
    rng = np.random.default_rng(0)
    fields = [rng.integers(-30000, 30000, (T, step_elems), dtype=np.int16).astype(npdt) for _ in range(5)]
    item = fields[0].itemsize
    carrier = {2: torch.int16, 4: torch.int32, 8: torch.int64}[item]
    cnp = {2: np.int16, 4: np.int32, 8: np.int64}[item]
    pool = ThreadPoolExecutor(args.threads)
    pinned = [[torch.empty((C, step_elems), dtype=carrier, pin_memory=True) for _ in range(5)] for _ in range(args.slots)]
    devbuf = [[torch.empty((C, step_elems), dtype=carrier, device=dev) for _ in range(5)] for _ in range(args.slots)]
    copier = torch.cuda.Stream(device=dev)
    rt = torch.cuda.cudart()
    if args.mode == "registered":
        t0 = time.perf_counter()
        for f in fields:
            rc = rt.cudaHostRegister(f.ctypes.data, f.nbytes, 0)
            assert int(rc) == 0, rc
        print(json.dumps({"register_seconds": time.perf_counter() - t0, "bytes": sum(f.nbytes for f in fields),
                          "GBs": sum(f.nbytes for f in fields) / (time.perf_counter() - t0) / 1e9}))
    piece = max(1, (8 << 20) // item)
    torch.cuda.synchronize()

    for rep in range(2):
        ev = []
        host = []
        done = [None] * args.slots
        t_begin = time.perf_counter()
        base_ev = torch.cuda.Event(enable_timing=True)
        base_ev.record(copier)
        for c in range((T + C - 1) // C):
            slot = c % args.slots
            a, b = c * C, min((c + 1) * C, T)
            if done[slot] is not None:
                done[slot].synchronize()
            h0 = time.perf_counter()
            if args.mode == "staged":
                jobs = []
                for fi, f in enumerate(fields):
                    dst = pinned[slot][fi].numpy()
                    for r in range(a, b):
                        src = f[r].view(cnp)
                        for o in range(0, step_elems, piece):
                            jobs.append((dst[r - a, o:o + piece], src[o:o + piece]))
                list(pool.map(lambda j: np.copyto(*j), jobs))
            h1 = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(copier):
                e0.record(copier)
                for fi, f in enumerate(fields):
                    if args.mode == "staged":
                        devbuf[slot][fi][: b - a].copy_(pinned[slot][fi][: b - a], non_blocking=True)
                    else:
                        devbuf[slot][fi][: b - a].copy_(torch.from_numpy(f[a:b].view(cnp)), non_blocking=True)
                e1.record(copier)
            h2 = time.perf_counter()
            done[slot] = e1
            ev.append((e0, e1))
            host.append((h0 - t_begin, h1 - t_begin, h2 - t_begin))
        torch.cuda.synchronize()
        total = time.perf_counter() - t_begin
        nbytes = 5 * T * step_elems * item
        if rep == 1:
            for c, ((e0, e1), (h0, h1, h2)) in enumerate(zip(ev, host)):
                print(f"chunk {c}: stage {h0 * 1e3:8.1f} .. {h1 * 1e3:8.1f} ms ({(h1 - h0) * 1e3:6.1f}), enqueue till {h2 * 1e3:8.1f}; "
                      f"H2D {base_ev.elapsed_time(e0):8.1f} .. {base_ev.elapsed_time(e1):8.1f} ms ({e0.elapsed_time(e1):6.1f})")
            print(json.dumps({"mode": args.mode, "src": args.src, "threads": args.threads, "slots": args.slots, "chunk": C, "timesteps": T,
                              "bytes": nbytes, "seconds": total, "GBs": nbytes / total / 1e9,
                              "stage_GBs": nbytes / max(sum(h1 - h0 for h0, h1, _ in host), 1e-9) / 1e9,
                              "h2d_GBs_while_copying": nbytes / (sum(e0.elapsed_time(e1) for e0, e1 in ev) * 1e-3) / 1e9}))


if __name__ == "__main__":
    main()
