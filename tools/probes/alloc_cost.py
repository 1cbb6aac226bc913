"""What does device memory cost to allocate on this box?  (profiles/r05_notes.md section 4: the streamed pipeline's `setup`)
torch.empty (caching allocator -> hipMalloc) against hipMalloc / hipMallocAsync called directly, by size, and the first touch."""
import ctypes as C
import time

import torch

hip = C.CDLL("libamdhip64.so")
torch.zeros(1, device="cuda")
torch.cuda.synchronize()


def t(f):
    torch.cuda.synchronize()
    a = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return time.perf_counter() - a, r


for gb in (1, 4, 16, 32):
    n = gb << 30
    dt, x = t(lambda: torch.empty(n, dtype=torch.uint8, device="cuda"))
    dz, _ = t(lambda: x.zero_())
    dz2, _ = t(lambda: x.zero_())
    del x
    df, _ = t(torch.cuda.empty_cache)
    p = C.c_void_p()
    dh, rc = t(lambda: hip.hipMalloc(C.byref(p), C.c_size_t(n)))
    dfree, _ = t(lambda: hip.hipFree(p))
    q = C.c_void_p()
    da, rc2 = t(lambda: hip.hipMallocAsync(C.byref(q), C.c_size_t(n), None))
    dafree, _ = t(lambda: hip.hipFreeAsync(q, None))
    da2, rc3 = t(lambda: hip.hipMallocAsync(C.byref(q), C.c_size_t(n), None))      # again: from the pool now?
    hip.hipFreeAsync(q, None)
    print(f"{gb:3d} GiB: torch.empty {dt * 1e3:8.1f} ms ({dt / gb * 1e3:5.1f} ms/GiB)  first zero_ {dz * 1e3:7.1f}  second {dz2 * 1e3:7.1f}  empty_cache {df * 1e3:7.1f} | "
          f"hipMalloc {dh * 1e3:8.1f} (rc {rc})  hipFree {dfree * 1e3:7.1f} | hipMallocAsync {da * 1e3:8.1f} (rc {rc2})  free {dafree * 1e3:6.1f}  again {da2 * 1e3:8.1f}", flush=True)
# many small against one large: 32 x 1 GiB
dt, xs = t(lambda: [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(32)])
print(f"32 x 1 GiB torch.empty: {dt * 1e3:.1f} ms")
del xs
torch.cuda.empty_cache()
# two threads allocating at once
import threading
out = {}


def work(k):
    a = time.perf_counter()
    p = C.c_void_p()
    hip.hipMalloc(C.byref(p), C.c_size_t(16 << 30))
    out[k] = (time.perf_counter() - a, p)


a = time.perf_counter()
th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
[x.start() for x in th]
[x.join() for x in th]
print(f"2 threads x 16 GiB hipMalloc: wall {1e3 * (time.perf_counter() - a):.1f} ms, each {[round(1e3 * v[0], 1) for v in out.values()]}")
for v in out.values():
    hip.hipFree(v[1])
