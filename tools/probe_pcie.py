import os, time, numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
n = 1 << 30
dev = torch.device("cuda:0")
d = torch.empty(n, dtype=torch.uint8, device=dev)
p = torch.empty(n, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize()
for _ in range(2):
    t = time.perf_counter(); d.copy_(p, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("pinned -> device GB/s", n / dt / 1e9)
src = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
dst = p.numpy()
for th in (1, 4, 8, 16, 32):
    pool = ThreadPoolExecutor(th)
    piece = 8 << 20
    jobs = [(dst[a:a + piece], src[a:a + piece]) for a in range(0, n, piece)]
    t = time.perf_counter(); list(pool.map(lambda j: np.copyto(*j), jobs)); dt = time.perf_counter() - t
    print("memcpy pageable->pinned threads", th, "GB/s", n / dt / 1e9)
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
# pageable -> device directly
t0 = torch.from_numpy(src)
for _ in range(2):
    t = time.perf_counter(); d.copy_(t0); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("pageable -> device GB/s", n / dt / 1e9)
# register in place
rt = torch.cuda.cudart()
try:
    rc = rt.cudaHostRegister(src.ctypes.data, n, 0)
    print("hostRegister anonymous rc", rc, "is_pinned", t0.is_pinned())
    for _ in range(2):
        t = time.perf_counter(); d.copy_(t0, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("registered -> device GB/s", n / dt / 1e9)
    rt.cudaHostUnregister(src.ctypes.data)
except Exception as e:
    print("hostRegister anonymous failed", e)
# mmap'd file
path = "/tmp/probe.bin"
src[: 1 << 28].tofile(path)
mm = np.memmap(path, dtype=np.uint8, mode="r")
try:
    rc = rt.cudaHostRegister(mm.ctypes.data, mm.size, 0)
    print("hostRegister mmap(r) rc", rc)
    tm = torch.from_numpy(np.asarray(mm))
    for _ in range(2):
        t = time.perf_counter(); d[: mm.size].copy_(tm, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("registered mmap -> device GB/s", mm.size / dt / 1e9)
    rt.cudaHostUnregister(mm.ctypes.data)
except Exception as e:
    print("hostRegister mmap failed", repr(e)[:200])
