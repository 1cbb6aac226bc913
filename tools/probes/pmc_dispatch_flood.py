"""How does `rocprofv3 --pmc` cope with thousands of small unsynchronised dispatches?  (profiles/r05_notes.md section 2)

Re-creates what round 4's synthetic_cube did for `bench.py --moving --timesteps 512` -- per time step and field: a draw, a scale, an
add and a copy into the cube, ~10,000 launches with no synchronisation in between -- without any lec_* kernel, and prints the host
time every 1000 launches and the time of the final synchronisation.  Run it plain and under
`rocprofv3 --pmc FETCH_SIZE -- python3 tools/probes/pmc_dispatch_flood.py N [launches between synchronisations]`.
"""
import sys
import time

import torch

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 514
sync_every = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # launches between device synchronisations (0: never)
dev = torch.device("cuda:0")
nl, ny, nx = 37, 162, 243
base = torch.zeros((nl, ny, nx), dtype=torch.float64, device=dev)
out = [torch.empty((n_steps, nl, ny, nx), dtype=torch.float64, device=dev) for _ in range(5)]
gen = torch.Generator(device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
launched = 0
for t in range(n_steps):
    gen.manual_seed(1234 + t)
    for k in range(5):
        noise = torch.randn((nl, ny, nx), generator=gen, dtype=torch.float64, device=dev)
        out[k][t] = base + 3.0 * noise            # mul, add, copy
        launched += 4
        if sync_every and launched % sync_every == 0:
            torch.cuda.synchronize()
        if launched % 1000 == 0:
            print(f"launched {launched:6d}  host {time.perf_counter() - t0:8.3f} s", flush=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{launched} launches: issue {t1 - t0:.3f} s, drain {t2 - t1:.3f} s, {1e3 * (t2 - t0) / launched:.3f} ms per launch", flush=True)
