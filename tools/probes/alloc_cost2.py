"""One allocation pattern per process (argv: piece GiB, number of pieces): time of torch.empty for each piece, and of touching them."""
import sys
import time

import torch

piece, count = float(sys.argv[1]), int(sys.argv[2])
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
n = int(piece * (1 << 30))
ts, xs = [], []
a0 = time.perf_counter()
for i in range(count):
    a = time.perf_counter()
    xs.append(torch.empty(n, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - a)
total = time.perf_counter() - a0
a = time.perf_counter()
for x in xs:
    x.zero_()
torch.cuda.synchronize()
touch = time.perf_counter() - a
print(f"{count:3d} x {piece:5.2f} GiB = {count * piece:6.1f} GiB: alloc {total * 1e3:8.1f} ms ({total / (count * piece) * 1e3:6.2f} ms/GiB; first {ts[0] * 1e3:.1f}, last {ts[-1] * 1e3:.1f}, max {max(ts) * 1e3:.1f})  zero all {touch * 1e3:7.1f} ms", flush=True)
