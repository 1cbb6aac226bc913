#!/usr/bin/env python3
"""What a process that pinned / mapped a lot of memory pays when it ends (the `exit` phase of tools/bench_cli.py): pinned host
allocations freed one by one at interpreter shutdown, against a hard exit.  Usage: exit_cost.py <GiB pinned> <mode: del|leave|hard>"""
import os
import sys
import time

import torch

gib, mode = float(sys.argv[1]), sys.argv[2]
t0 = time.time()
n = int(gib * (1 << 30))
bufs = [torch.empty(n // 8, dtype=torch.uint8, pin_memory=True) for _ in range(8)]
dev = [torch.empty(n // 8, dtype=torch.uint8, device="cuda:0") for _ in range(8)]
for b, d in zip(bufs, dev):
    d.copy_(b, non_blocking=True)
torch.cuda.synchronize()
t1 = time.time()
print(f"pin + copy {gib} GiB: {t1 - t0:.3f} s", flush=True)
if mode == "del":
    del bufs
    import gc
    gc.collect()
    print(f"del pinned: {time.time() - t1:.3f} s", flush=True)
open(f"/tmp/exit_cost_{os.getpid()}", "w").write(repr(time.time()))
print(os.getpid(), flush=True)
if mode == "hard":
    os._exit(0)
