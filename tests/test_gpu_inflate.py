"""lec_inflate / lec_chunk_scatter through the C ABI (GPU): zlib is the oracle for the first, NumPy indexing for the second.
The reference has no counterpart in its own code -- it reads deflated NetCDF-4 files through netCDF4 / HDF5, which inflate with
zlib on the host (src/utils/preprocessing.py:35-146) -- so "what zlib returns" is the contract."""
import ctypes as C
import zlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from lorenzcycletoolkit_amd import _lib

DEV = "cuda:0"


def payload(rng, n, kind):
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()                    # noise: literals, stored blocks at level 0
    if kind == 1:                                                                      # a shuffled int16 field: smooth high bytes, noisy low bytes
        m = max(1, n // 2)
        x = (np.cumsum(rng.standard_normal(m)) * 40).astype(np.int16)
        return x.view(np.uint8).reshape(m, 2).T.copy().reshape(-1).tobytes()[:n].ljust(n, b"\0")
    if kind == 2:
        return bytes(rng.integers(0, 4, n, dtype=np.uint8))                          # four symbols: short codes, many matches
    if kind == 3:                                                                      # far repeats: matches that reach behind the LDS ring
        base = rng.integers(0, 256, max(1, n // 7), dtype=np.uint8).tobytes()
        return (base * 8)[:n].ljust(n, b"x")
    if kind == 4:
        return bytes(n)                                                                # zeros: length-258 matches at distance 1
    p = 1.0 / np.arange(1, 257) ** 1.3                                                 # skewed alphabet: codes longer than the lookup width
    return rng.choice(256, n, p=p / p.sum()).astype(np.uint8).tobytes()


def deflate(rng, data):
    level = int(rng.integers(0, 10))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    c = zlib.compressobj(level, zlib.DEFLATED, 15, int(rng.integers(1, 10)), strategy)
    if len(data) > 1000 and rng.random() < 0.25:                                       # a sync flush: an empty stored block mid-stream
        half = len(data) // 2
        return c.compress(data[:half]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(data[half:]) + c.flush()
    return c.compress(data) + c.flush()


def inflate(streams, sizes, *, packed=False, plain=(), flags=0):
    """Runs lec_inflate; ``packed``: streams back to back at arbitrary byte offsets; ``plain``: indices passed as stored-as-is."""
    lib = _lib.load()
    n = len(streams)
    desc = np.zeros((n, 4), dtype=np.int64)
    so = do = 0
    for i, (s, m) in enumerate(zip(streams, sizes)):
        desc[i] = (so, -len(s) if i in plain else len(s), do, m)
        so += len(s) if packed else (len(s) + 15) & ~15
        do += (m + 15) & ~15
    src = np.zeros(so + 1024, dtype=np.uint8)
    for i, s in enumerate(streams):
        src[desc[i, 0]: desc[i, 0] + len(s)] = np.frombuffer(s, dtype=np.uint8)
    src_d, desc_d = torch.from_numpy(src).to(DEV), torch.from_numpy(desc).to(DEV)
    dst_d = torch.full((do + 16,), 0xAA, dtype=torch.uint8, device=DEV)
    status_d = torch.full((n, 4), -1, dtype=torch.int32, device=DEV)
    a = _lib.InflateArgs(src_d=src_d.data_ptr(), src_bytes=src.size, desc_d=desc_d.data_ptr(), n_streams=n, flags=flags, dst_d=dst_d.data_ptr(),
                         status_d=status_d.data_ptr(), stream=C.c_void_p(torch.cuda.current_stream().cuda_stream), dst_bytes=dst_d.numel())
    _lib.check(lib.lec_inflate(C.byref(a)), "lec_inflate")
    torch.cuda.synchronize()
    out, status = dst_d.cpu().numpy(), status_d.cpu().numpy()
    return [out[desc[i, 2]: desc[i, 2] + sizes[i]].tobytes() for i in range(n)], status, out, desc


@pytest.mark.parametrize("short_ring", [False, True])
@pytest.mark.parametrize("packed", [False, True])
def test_random_streams_inflate_to_what_zlib_returns(packed, short_ring):
    """150 streams in one launch: every deflate level and strategy (stored, fixed and dynamic blocks, Huffman-only, RLE), window
    memory levels, six kinds of data, 0 bytes to 300 KB; 16-byte-aligned and back-to-back (any byte offset) layouts; the 8 KiB and the
    4 KiB history ring (flags bit 1: a hint only -- far matches are then served from HBM).  Nothing is written outside a stream's own output."""
    rng = np.random.default_rng(20 + packed)
    data = []
    for c in range(150):
        n = int(rng.choice([0, 1, 2, 3, 100, 1000, int(rng.integers(1, 70000)), int(rng.integers(1, 300000))]))
        data.append(payload(rng, n, c % 6))
    streams = [deflate(rng, d) for d in data]
    assert all(zlib.decompress(s) == d for s, d in zip(streams, data))
    got, status, out, desc = inflate(streams, [len(d) for d in data], packed=packed, flags=2 if short_ring else 0)
    assert (status[:, 0] == 0).all(), [(i, status[i].tolist()) for i in np.flatnonzero(status[:, 0])][:5]
    for i, d in enumerate(data):
        assert got[i] == d, (i, len(d))
        assert status[i, 2] == len(d)
        gap = out[desc[i, 2] + len(d): desc[i, 2] + ((len(d) + 15) & ~15)]
        assert (gap == 0xAA).all(), i                                                 # the padding after a stream's output is untouched


def test_long_streams_window_sized_matches_and_plain_chunks():
    """3 MB streams (hundreds of deflate blocks each), matches at the far end of the 32 KB window (served from HBM, behind the LDS
    ring), maximal matches at distance 1, and a chunk HDF5 stored as it is (negative size: copied)."""
    rng = np.random.default_rng(3)
    period = rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    data = [payload(rng, 3_000_000, 1), (period * 40)[:1_000_000], bytes(2_000_000), payload(rng, 1_000_001, 5), payload(rng, 70001, 0)]
    streams = [zlib.compress(data[0], 6), zlib.compress(data[1], 9), zlib.compress(data[2], 1), zlib.compress(data[3], 4), data[4]]
    for flags in (0, 2):
        got, status, _, _ = inflate(streams, [len(d) for d in data], plain={4}, flags=flags)
        assert (status[:, 0] == 0).all(), status.tolist()
        assert all(g == d for g, d in zip(got, data))
    assert status[0, 1] > 50                                                           # many blocks were walked


def test_malformed_streams_end_with_a_status():
    """Truncated, bit-flipped and random streams, wrong sizes: a status per stream, no hang, no write beyond the stream's output."""
    lib = _lib.load()
    rng = np.random.default_rng(9)
    good = [zlib.compress(payload(rng, 20000, k % 6), 6) for k in range(40)]
    bad, sizes = [], []
    for k, z in enumerate(good):
        z = bytearray(z)
        if k % 4 == 0:
            z = z[: len(z) // 2]
        elif k % 4 == 1:
            for _ in range(4):
                z[int(rng.integers(2, len(z)))] ^= 1 << int(rng.integers(0, 8))
        elif k % 4 == 2:
            z = bytearray(b"\x78\x9c") + bytearray(rng.integers(0, 256, 300, dtype=np.uint8).tobytes())
        bad.append(bytes(z)); sizes.append(20000 if k % 4 != 3 else 19999)              # k % 4 == 3: a good stream, the wrong size
    got, status, out, desc = inflate(bad, sizes)
    assert (status[:, 0] >= 0).all() and (status[:, 0] <= 13).all()
    assert (status[0::4, 0] != 0).all() and (status[3::4, 0] != 0).all()                # truncated / wrong size: always caught
    for i in range(len(bad)):
        end = desc[i, 2] + ((sizes[i] + 15) & ~15)
        nxt = desc[i + 1, 2] if i + 1 < len(bad) else out.size - 16
        assert (out[end:nxt] == 0xAA).all()
        assert lib.lec_inflate_status_text(int(status[i, 0]))
    # a flipped DATA bit that leaves the stream's structure intact (a stored block's payload): only zlib's Adler-32 of the inflated
    # data can see it -- the host's zlib.decompress raises, so must the device; the untouched streams beside it pass
    data = payload(rng, 5000, 0)
    stored = bytearray(zlib.compress(data, 0))
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    fixed = c.compress(data) + c.flush()
    stored[2000] ^= 0x10
    with pytest.raises(zlib.error):
        zlib.decompress(bytes(stored))
    _, st2, _, _ = inflate([bytes(stored), zlib.compress(data, 0), bytes(fixed)], [5000, 5000, 5000])
    assert st2[0, 0] == 13 and b"adler32" in lib.lec_inflate_status_text(13) and st2[1, 0] == 0 and st2[2, 0] == 0
    # arguments
    st = torch.zeros(4, dtype=torch.int32, device=DEV)
    a = _lib.InflateArgs(src_d=st.data_ptr(), src_bytes=16, desc_d=st.data_ptr(), n_streams=0, dst_d=st.data_ptr(), status_d=st.data_ptr(), dst_bytes=16)
    assert lib.lec_inflate(C.byref(a)) == 1 and b"n_streams" in lib.lec_last_error()
    a.n_streams, a.dst_d = 1, 0
    assert lib.lec_inflate(C.byref(a)) == 1 and b"null" in lib.lec_last_error()


@pytest.mark.parametrize("es,shuffled", [(2, True), (2, False), (4, True), (8, True), (1, False), (4, False)])
def test_chunk_scatter_puts_chunks_in_place(es, shuffled):
    """Chunks of 2 x 3 x 5 x 7 elements tiling a 5 x 7 x 11 x 16 variable (edge chunks padded), byte-shuffled or not: a selection of
    time steps (out of order), levels and a latitude band lands where NumPy indexing puts it; unselected output stays untouched."""
    lib = _lib.load()
    rng = np.random.default_rng(es * 2 + shuffled)
    shape, chunk = (5, 7, 11, 16), (2, 3, 5, 7)
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[es]
    a = rng.integers(0, np.iinfo(dt).max, shape, dtype=dt, endpoint=True)
    origins = [(t, k, j, i) for t in range(0, 5, 2) for k in range(0, 7, 3) for j in range(0, 11, 5) for i in range(0, 16, 7)]
    n_elem = int(np.prod(chunk))
    blobs, recs, at = [], [], 0
    for o in origins:
        blk = np.zeros(chunk, dtype=dt)
        part = a[o[0]: o[0] + 2, o[1]: o[1] + 3, o[2]: o[2] + 5, o[3]: o[3] + 7]
        blk[: part.shape[0], : part.shape[1], : part.shape[2], : part.shape[3]] = part
        raw = blk.reshape(-1).view(np.uint8)
        raw = raw.reshape(n_elem, es).T.copy().reshape(-1) if shuffled else raw
        recs.append((at,) + o)
        blobs.append(raw.tobytes() + b"\0" * (-len(raw) % 16))
        at += len(blobs[-1])
    src = torch.from_numpy(np.frombuffer(b"".join(blobs), dtype=np.uint8).copy()).to(DEV)
    recs_d = torch.tensor(recs, dtype=torch.int64, device=DEV)
    steps, levels, j0, j1 = [3, 0, 4], [1, 2, 5, 6], 2, 9                              # output rows <- file steps; kept levels; latitude band
    tmap = np.full(5, -1, dtype=np.int32)
    for r, t in enumerate(steps):
        tmap[t] = r + 1                                                                  # output row 0 stays free
    kmap = np.full(7, -1, dtype=np.int32)
    kmap[levels] = np.arange(4)
    tmap_d, kmap_d = torch.from_numpy(tmap).to(DEV), torch.from_numpy(kmap).to(DEV)
    sentinel = np.iinfo(dt).max // 3
    out = torch.full((4, 4, j1 - j0 + 1, 16), sentinel, dtype={1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[es], device=DEV)
    sa = _lib.ChunkScatterArgs(src_d=src.data_ptr(), chunk_d=recs_d.data_ptr(), n_chunks=len(origins), elem_size=es, shuffled=int(shuffled),
                               ct=2, ck=3, cj=5, ci=7, t_base=0, n_tmap=5, n_kmap=7, j0=j0, tmap_d=tmap_d.data_ptr(), kmap_d=kmap_d.data_ptr(),
                               nt=4, nl=4, ny=j1 - j0 + 1, nx=16, out_d=out.data_ptr(), stream=C.c_void_p(torch.cuda.current_stream().cuda_stream),
                               src_bytes=src.numel())
    _lib.check(lib.lec_chunk_scatter(C.byref(sa)), "lec_chunk_scatter")
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(dt)
    assert (got[0] == sentinel).all()
    for r, t in enumerate(steps):
        assert np.array_equal(got[r + 1], a[t][levels][:, j0: j1 + 1]), (r, t)
    # malformed chunk records: a payload offset that is negative or runs past src_bytes is skipped on the device -- the
    # rows it would have filled keep what they held, the other chunks land as before, nothing outside `out` is touched
    bad = recs_d.clone()
    bad[0, 0] = -16
    bad[1, 0] = src.numel() - 8
    out2 = torch.full_like(out, sentinel)
    sa.chunk_d, sa.out_d = bad.data_ptr(), out2.data_ptr()
    _lib.check(lib.lec_chunk_scatter(C.byref(sa)), "lec_chunk_scatter")
    torch.cuda.synchronize()
    got2 = out2.cpu().numpy().view(dt)
    differs = got2 != got
    assert differs.any() and (got2[differs] == sentinel).all()          # only the two skipped chunks' elements are missing
    sa.src_bytes = 8
    assert lib.lec_chunk_scatter(C.byref(sa)) == 1 and b"src_bytes" in lib.lec_last_error()
    sa.src_bytes = src.numel()
    sa.elem_size = 3
    assert lib.lec_chunk_scatter(C.byref(sa)) == 1 and b"elem_size" in lib.lec_last_error()


def test_malformed_descriptors_end_with_a_status_code_not_a_write():
    """the destination side of a descriptor is checked on the device like the source side -- an output offset that is
    negative, not a multiple of 16, or whose output would end beyond dst_bytes gets status "size" (10); nothing of that stream is
    written (the guard bytes around the buffer stay), and the well-formed streams of the same launch inflate as usual."""
    lib = _lib.load()
    rng = np.random.default_rng(5)
    data = [payload(rng, 3000, k) for k in range(6)]
    streams = [zlib.compress(d, 6) for d in data]
    n = len(streams)
    src_off = np.cumsum([0] + [(len(s) + 15) & ~15 for s in streams])
    src = np.zeros(int(src_off[-1]) + 1024, dtype=np.uint8)
    for i, s in enumerate(streams):
        src[src_off[i]: src_off[i] + len(s)] = np.frombuffer(s, dtype=np.uint8)
    room = 3008
    dst_bytes = n * room
    guard = 4096
    whole = torch.full((guard + dst_bytes + guard,), 0xAA, dtype=torch.uint8, device=DEV)
    desc = np.array([(src_off[i], len(streams[i]), i * room, 3000) for i in range(n)], dtype=np.int64)
    desc[1, 2] = -16                       # before the buffer
    desc[2, 2] = 2 * room + 8              # not a multiple of 16
    desc[3, 2] = dst_bytes - 1008          # would end beyond dst_bytes
    desc[4, 2] = 1 << 40                   # far outside
    src_d, desc_d = torch.from_numpy(src).to(DEV), torch.from_numpy(desc).to(DEV)
    status_d = torch.full((n, 4), -1, dtype=torch.int32, device=DEV)
    a = _lib.InflateArgs(src_d=src_d.data_ptr(), src_bytes=src.size, desc_d=desc_d.data_ptr(), n_streams=n, flags=0,
                         dst_d=whole.data_ptr() + guard, status_d=status_d.data_ptr(), stream=C.c_void_p(torch.cuda.current_stream().cuda_stream),
                         dst_bytes=dst_bytes)
    _lib.check(lib.lec_inflate(C.byref(a)), "lec_inflate")
    torch.cuda.synchronize()
    st, out = status_d.cpu().numpy(), whole.cpu().numpy()
    assert st[:, 0].tolist() == [0, 10, 10, 10, 10, 0] and b"size" in lib.lec_inflate_status_text(10)
    assert (out[:guard] == 0xAA).all() and (out[guard + dst_bytes:] == 0xAA).all()
    body = out[guard: guard + dst_bytes]
    assert body[:3000].tobytes() == data[0] and body[5 * room: 5 * room + 3000].tobytes() == data[5]
    assert (body[room: 5 * room] == 0xAA).all()                                  # the four refused streams wrote nothing
    a.dst_bytes = 0
    assert lib.lec_inflate(C.byref(a)) == 1 and b"dst_bytes" in lib.lec_last_error()


def test_fletcher32_of_a_large_stored_chunk():
    """The wave's Fletcher-32 folds its partial sums (unfolded 64-bit sums wrap beyond ~47 MB and would report a false
    checksum error): a 64 MiB chunk stored as it is, with the checksum hdf5_lite computes on the host; and a corrupt byte is seen."""
    from lorenzcycletoolkit_amd import hdf5_lite
    lib = _lib.load()
    rng = np.random.default_rng(9)
    n = 64 << 20
    data = rng.integers(0, 256, n, dtype=np.uint8)
    data[: n // 4] = 0xFF                                                           # long runs of large words: the sums grow fastest
    ck = hdf5_lite._fletcher32(data.tobytes())[0]
    for flip in (False, True):
        src = np.zeros(n + 4 + 1024, dtype=np.uint8)
        src[:n] = data
        src[n: n + 4] = np.frombuffer(int(ck).to_bytes(4, "little"), dtype=np.uint8)
        if flip:
            src[n // 2] ^= 1
        src_d = torch.from_numpy(src).to(DEV)
        desc_d = torch.tensor([[0, -n, 0, n]], dtype=torch.int64, device=DEV)
        dst_d = torch.zeros(n + 16, dtype=torch.uint8, device=DEV)
        status_d = torch.full((1, 4), -1, dtype=torch.int32, device=DEV)
        a = _lib.InflateArgs(src_d=src_d.data_ptr(), src_bytes=src.size, desc_d=desc_d.data_ptr(), n_streams=1, flags=1, dst_d=dst_d.data_ptr(),
                             status_d=status_d.data_ptr(), stream=C.c_void_p(torch.cuda.current_stream().cuda_stream), dst_bytes=dst_d.numel())
        _lib.check(lib.lec_inflate(C.byref(a)), "lec_inflate")
        torch.cuda.synchronize()
        code = int(status_d[0, 0])
        if flip:
            assert code == 12
        else:
            assert code == 0 and torch.equal(dst_d[:n].cpu(), torch.from_numpy(data))
