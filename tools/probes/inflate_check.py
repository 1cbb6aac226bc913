#!/usr/bin/env python3
"""lec_inflate against zlib on random streams (GPU box): levels 0-9, strategies (default / filtered / huffman-only / rle / fixed),
data kinds (noise, shuffled int16 fields, runs, text-like, zeros), sizes from 0 bytes to a few MB.  `--cases N --seed S`."""
import argparse
import ctypes as C
import os
import sys
import time
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lorenzcycletoolkit_amd import _lib  # noqa: E402


def make_payload(rng, n):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 1:                                   # a shuffled int16 field: smooth high bytes, noisy low bytes
        m = max(1, n // 2)
        x = (np.cumsum(rng.standard_normal(m)) * 40).astype(np.int16)
        b = x.view(np.uint8).reshape(m, 2).T.copy().reshape(-1)
        return b.tobytes()[:n].ljust(n, b"\0")
    if kind == 2:
        return bytes(rng.integers(0, 4, n, dtype=np.uint8))
    if kind == 3:                                   # long runs and far repeats
        base = rng.integers(0, 256, max(1, n // 7), dtype=np.uint8).tobytes()
        return (base * 8)[:n].ljust(n, b"x")
    if kind == 4:
        return bytes(n)
    if kind == 5:                                   # text-like: skewed alphabet (long codes for rare symbols)
        p = 1.0 / np.arange(1, 257) ** 1.3
        return rng.choice(256, n, p=p / p.sum()).astype(np.uint8).tobytes()
    x = (np.sin(np.arange(max(1, n // 4)) * 0.01) * 1000).astype(np.float32)       # unshuffled floats
    return x.tobytes()[:n].ljust(n, b"\0")


def compress(rng, data):
    level = int(rng.integers(0, 10))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    mem = int(rng.integers(1, 10))
    c = zlib.compressobj(level, zlib.DEFLATED, 15, mem, strategy)
    out = c.compress(data)
    if len(data) > 1000 and rng.random() < 0.2:     # a sync flush in the middle: an empty stored block
        half = len(data) // 2
        c = zlib.compressobj(level, zlib.DEFLATED, 15, mem, strategy)
        out = c.compress(data[:half]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(data[half:])
    return out + c.flush(), (level, strategy, mem)


FLAGS = 0


def run(lib, streams, sizes, dev, flags=None):
    n = len(streams)
    desc = np.zeros((n, 4), dtype=np.int64)
    so = do = 0
    for i, (s, m) in enumerate(zip(streams, sizes)):
        desc[i] = (so, len(s), do, m)
        so += (len(s) + 15) & ~15
        do += (m + 15) & ~15
    src = np.zeros(so + 1024, dtype=np.uint8)
    for i, s in enumerate(streams):
        src[desc[i, 0]: desc[i, 0] + len(s)] = np.frombuffer(s, dtype=np.uint8)
    src_d = torch.from_numpy(src).to(dev)
    desc_d = torch.from_numpy(desc).to(dev)
    dst_d = torch.full((do + 16,), 0xAA, dtype=torch.uint8, device=dev)
    status_d = torch.full((n, 4), -1, dtype=torch.int32, device=dev)
    a = _lib.InflateArgs(src_d=src_d.data_ptr(), src_bytes=src.size, desc_d=desc_d.data_ptr(), n_streams=n, flags=FLAGS if flags is None else flags, dst_d=dst_d.data_ptr(),
                         status_d=status_d.data_ptr(), stream=C.c_void_p(torch.cuda.current_stream().cuda_stream), dst_bytes=dst_d.numel())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.lec_inflate(C.byref(a)), "lec_inflate")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return desc, dst_d.cpu().numpy(), status_d.cpu().numpy(), dt


def bench(lib, dev, n_streams, level, noise):
    """ERA5-like chunks (361 x 720 int16, shuffle + deflate): `n_streams` streams cycled from 48 distinct ones; kernel time by events."""
    rng = np.random.default_rng(5)
    distinct, raw = [], []
    for i in range(48):
        y, x = np.mgrid[0:361, 0:720]
        f = 250.0 + 30.0 * np.sin(x * 0.01 + i) * np.cos(y * 0.02) + noise * rng.standard_normal((361, 720))
        q = np.round((f - 250.0) / 0.002).clip(-32000, 32000).astype(np.int16)
        sh = q.view(np.uint8).reshape(-1, 2).T.copy().reshape(-1).tobytes()
        distinct.append(zlib.compress(sh, level)); raw.append(sh)
    streams = [distinct[i % 48] for i in range(n_streams)]
    sizes = [len(raw[i % 48]) for i in range(n_streams)]
    ratio = sum(sizes) / sum(len(s) for s in streams)
    best = None
    for rep in range(4):
        desc, out, status, dt = run(lib, streams, sizes, dev)
        best = dt if best is None else min(best, dt)
    ok = bool((status[:, 0] == 0).all()) and all(out[desc[i, 2]: desc[i, 2] + sizes[i]].tobytes() == raw[i % 48] for i in range(0, n_streams, 97))
    total = sum(sizes)
    print(f"bench: {n_streams} streams of {sizes[0]} bytes, deflate level {level}, noise {noise}: ratio {ratio:.2f}, "
          f"{best * 1e3:.2f} ms -> {total / best / 1e9:.2f} GB/s inflated, {total / ratio / best / 1e9:.2f} GB/s compressed; correct: {ok}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="lec_inflate flags (2: the 4 KiB ring)")
    ap.add_argument("--bench-one", nargs=3, metavar=("STREAMS", "LEVEL", "NOISE"), help="one configuration (for profiling)")
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-bytes", type=int, default=1 << 20)
    a = ap.parse_args()
    global FLAGS
    FLAGS = a.flags
    rng = np.random.default_rng(a.seed)
    lib = _lib.load()
    dev = "cuda:0"
    if a.bench_one:
        bench(lib, dev, int(a.bench_one[0]), int(a.bench_one[1]), float(a.bench_one[2]))
        return
    if a.bench:
        for n in (1536, 6144):
            for level, noise in ((4, 0.5), (1, 0.5), (4, 0.02)):
                bench(lib, dev, n, level, noise)
        return
    payloads, streams, how = [], [], []
    for c in range(a.cases):
        n = int(rng.choice([0, 1, 2, 100, 1000, int(rng.integers(1, 70000)), int(rng.integers(1, a.max_bytes))]))
        data = make_payload(rng, n)
        z, h = compress(rng, data)
        payloads.append(data); streams.append(z); how.append(h)
    desc, out, status, dt = run(lib, streams, [len(p) for p in payloads], dev)
    fails = 0
    for i, p in enumerate(payloads):
        got = out[desc[i, 2]: desc[i, 2] + len(p)].tobytes()
        if status[i, 0] != 0 or got != p:
            fails += 1
            first = next((k for k in range(len(p)) if k >= len(got) or got[k] != p[k]), -1)
            print(f"FAIL stream {i}: {len(p)} bytes -> {len(streams[i])}, level/strategy/mem {how[i]}: status {status[i].tolist()} "
                  f"({lib.lec_inflate_status_text(int(status[i, 0])).decode()}), first difference at {first}")
    total = sum(len(p) for p in payloads)
    print(f"inflate check: {a.cases} streams, {total / 1e6:.1f} MB inflated in {dt * 1e3:.1f} ms ({total / dt / 1e9:.2f} GB/s incl. launch), {fails} failures")
    # malformed input must end with a status, not a hang: truncated, bit-flipped and random streams
    bad, sizes = [], []
    for i in range(min(60, a.cases)):
        z = bytearray(streams[i])
        k = int(rng.integers(0, 3))
        if k == 0 and len(z) > 8:
            z = z[: len(z) // 2]
        elif k == 1 and len(z) > 8:
            for _ in range(3):
                z[int(rng.integers(2, len(z)))] ^= 1 << int(rng.integers(0, 8))
        else:
            z = bytearray(b"\x78\x9c") + bytearray(rng.integers(0, 256, 200, dtype=np.uint8).tobytes())
        bad.append(bytes(z)); sizes.append(len(payloads[i]))
    _d, _o, st, dt2 = run(lib, bad, sizes, dev)
    print(f"malformed streams: {len(bad)} ended in {dt2 * 1e3:.1f} ms with statuses {sorted(set(st[:, 0].tolist()))}")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
