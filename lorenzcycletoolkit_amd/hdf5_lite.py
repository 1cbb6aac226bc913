"""A small read-only HDF5 reader for NetCDF-4 files (this image has no HDF5 library for
the default interpreter).  Pure Python + zlib + NumPy; it understands what netCDF-C / h5netcdf / h5py write for
gridded data:

* superblock versions 0-3; object headers version 1 and 2 (with continuation blocks);
* the root group as a symbol table (v1 B-tree + local heap), as compact link messages, or as dense link storage
  (v2 B-tree name index + fractal heap);
* datasets: contiguous, compact, and chunked (layout v3 with a v1 B-tree; layout v4 single-chunk / implicit /
  fixed-array / extensible-array / v2-B-tree indexes), filters deflate + shuffle + fletcher32 (verified);
* datatypes: integers and IEEE floats of 1-8 bytes in either byte order, fixed- and variable-length strings,
  object references (for DIMENSION_LIST);
* attributes in the object header or in dense storage.

Not supported (raises Hdf5Error): sub-groups, compound / enum data, other filters (szip, lzf, zstd ...), external or
virtual storage, v2 B-trees deeper than two levels.  tools/soak_hdf5.py checks the reader against h5py on random files (build
container only).  Format reference: "HDF5 File Format Specification
Version 3.0" (The HDF Group).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class Hdf5Error(ValueError):
    pass


class ChunkTable:
    """The chunk index of a large chunked dataset as ARRAYS: origins [n, rank] and three values per chunk (file address, stored size,
    filter mask -- or, from ``chunk_streams``, offset in the mapped file, stream bytes, deflate skipped).  It answers like the dict
    {origin tuple: (a, b, c)} the small-file paths use (``[]``, ``get``, ``items``, ``len``, ``in``), but a month of hourly ERA5
    -- 110,000 chunks per variable, 550,000 per file -- is never turned into Python tuples: ``lookup`` takes an array of origins
    (round 5: the dict cost 0.3 s of a 1.8-s run, profiles/r05_notes.md section 8)."""

    def __init__(self, offs: np.ndarray, a: np.ndarray, b: np.ndarray, c: np.ndarray, chunk, shape):
        self.chunk = tuple(int(x) for x in chunk)
        self.counts = tuple(-(-int(s) // c) for s, c in zip(shape, self.chunk))
        offs = np.asarray(offs, dtype=np.int64).reshape(-1, len(self.chunk))
        a, b, c = (np.asarray(x) for x in (a, b, c))
        # a B-tree may still list chunks beyond the dataset's CURRENT extent (an unlimited dimension that was shrunk): they hold no
        # element of the dataset and are ignored, as the dict of the small-file paths never looked them up (and they do not count as
        # "written" chunks: __len__)
        g = offs // np.array(self.chunk, dtype=np.int64) if len(offs) else offs
        inside = ((g >= 0) & (g < np.array(self.counts, dtype=np.int64))).all(axis=1) if len(offs) else np.zeros(0, dtype=bool)
        if not inside.all():
            offs, a, b, c = offs[inside], a[inside], b[inside], c[inside]
        self.offs = np.ascontiguousarray(offs, dtype=np.int64)
        self.a, self.b, self.c = a, b, c
        self._index = None

    def __len__(self):
        return len(self.a)

    def _rows(self, origins: np.ndarray) -> np.ndarray:
        """Row of every origin ([m, rank], elements), -1 where the chunk was never written or lies outside the dataset."""
        if self._index is None:
            idx = np.full(int(np.prod(self.counts, dtype=np.int64)), -1, dtype=np.int64)
            if len(self.a):
                idx[np.ravel_multi_index(tuple((self.offs // np.array(self.chunk)).T), self.counts)] = np.arange(len(self.a))
            self._index = idx
        o = np.asarray(origins, dtype=np.int64).reshape(-1, len(self.chunk))
        ch = np.array(self.chunk)
        g = o // ch
        ok = ((o % ch) == 0).all(axis=1) & (g >= 0).all(axis=1) & (g < np.array(self.counts)).all(axis=1)
        rows = np.full(len(o), -1, dtype=np.int64)
        if ok.any():
            rows[ok] = self._index[np.ravel_multi_index(tuple(g[ok].T), self.counts)]
        return rows

    def lookup(self, origins):
        """(a, b, c) arrays of the chunks at ``origins`` [m, rank]; KeyError if one is not in the index."""
        rows = self._rows(origins)
        if (rows < 0).any():
            raise KeyError(tuple(int(x) for x in np.asarray(origins).reshape(-1, len(self.chunk))[int(np.argmax(rows < 0))]))
        return self.a[rows], self.b[rows], self.c[rows]

    def __getitem__(self, key):
        a, b, c = self.lookup(np.asarray(key, dtype=np.int64)[None, :])
        return (a[0].item(), b[0].item(), c[0].item())

    def get(self, key, default=None):
        r = int(self._rows(np.asarray(key, dtype=np.int64)[None, :])[0])
        return default if r < 0 else (self.a[r].item(), self.b[r].item(), self.c[r].item())

    def __contains__(self, key):
        return int(self._rows(np.asarray(key, dtype=np.int64)[None, :])[0]) >= 0

    def keys(self):
        return map(tuple, self.offs.tolist())

    __iter__ = keys

    def values(self):
        return zip(self.a.tolist(), self.b.tolist(), self.c.tolist())

    def items(self):
        return zip(self.keys(), self.values())

    def with_values(self, a, b, c) -> "ChunkTable":
        t = ChunkTable.__new__(ChunkTable)
        t.offs, t.chunk, t.counts, t._index = self.offs, self.chunk, self.counts, self._index
        t.a, t.b, t.c = np.asarray(a), np.asarray(b), np.asarray(c)
        return t


@dataclass
class _Dtype:
    kind: str                       # "num", "str", "vlen_str", "vlen", "ref", "other"
    np_dtype: Optional[np.dtype] = None
    size: int = 0
    base: Optional["_Dtype"] = None


@dataclass
class H5Variable:
    """One dataset: shape, NumPy dtype (file byte order), attributes, dimension names, and lazy reads along axis 0."""
    name: str
    shape: Tuple[int, ...]
    dtype: np.dtype
    attrs: Dict[str, object]
    dims: Tuple[str, ...] = ()
    _file: "H5File" = None
    _layout: dict = field(default_factory=dict)
    _filters: list = field(default_factory=list)
    _cache: dict = field(default_factory=dict)

    @property
    def ndim(self):
        return len(self.shape)

    def read(self) -> np.ndarray:
        """The whole array."""
        return self._file._read_dataset(self, None)

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, t) -> np.ndarray:
        """``var[t]`` for an integer t: the block at index t of axis 0 (a contiguous array)."""
        if not isinstance(t, (int, np.integer)):
            raise TypeError("H5Variable supports integer indexing along axis 0 and read()")
        t = int(t)
        if t < 0:
            t += self.shape[0]
        if not 0 <= t < self.shape[0]:
            raise IndexError(t)
        return self._file._read_dataset(self, t)


    def read_step(self, t: int, levels) -> np.ndarray:
        """``var[t][levels]`` for a variable of rank >= 2 (axis 1 = level) that inflates only the chunks those levels lie in -- the
        850-hPa diagnostics of a track need one level of three variables, not the 37 a whole time step holds."""
        levels = [int(k) for k in levels]
        t = int(t) + (self.shape[0] if t < 0 else 0)
        if not 0 <= t < self.shape[0] or any(not 0 <= k < self.shape[1] for k in levels):
            raise IndexError((t, levels))
        return self._file._read_dataset(self, t, levels)[levels]

    def chunk_streams(self):
        """What a device-side inflate needs, or None when this variable is not a fully written chunked dataset with no filter other
        than shuffle / deflate / fletcher32 (a chunked variable WITHOUT deflate -- every record variable of an uncompressed NetCDF-4
        file -- qualifies: its chunks are "stored as they are" and are copied into place):
        ``{"chunk": chunk shape, "shuffle": bool, "table": {chunk origin (elements) -> (offset in the mapped file, stored bytes,
        deflate skipped for this chunk)} -- a dict, or for large v1-B-tree indexes a ``ChunkTable`` (the same mapping kept as arrays,
        with a vectorised ``lookup``) --, "fletcher32": bool, "map": the file's memory map}``.  The stored bytes of a chunk are its zlib
        stream; with ``fletcher32`` four checksum bytes follow them (not counted in "stored bytes"; ``lec_inflate`` verifies them)."""
        if "streams" not in self._cache:
            self._cache["streams"] = self._chunk_streams()
        return self._cache["streams"]

    def _chunk_streams(self):
        lay = self._layout
        ids = [fid for fid, _cd in self._filters]
        if lay.get("class") != "chunked" or any(f not in (1, 2, 3) for f in ids):
            return None
        if sorted(ids, key=lambda f: {2: 0, 1: 1, 3: 2}[f]) != ids or len(set(ids)) != len(ids):
            return None                                    # the usual pipeline order only: shuffle, deflate, fletcher32
        if 2 in ids and self._filters[ids.index(2)][1] and self._filters[ids.index(2)][1][0] != self.dtype.itemsize:
            return None
        # what lec_inflate takes: stored sizes below 2^28 bytes, chunks below 2^31 (its bit and output positions are 32-bit); an
        # uncompressed record variable kept as ONE huge chunk per step stays on the host reader
        if int(np.prod(lay["chunk"], dtype=np.int64)) * self.dtype.itemsize >= (1 << 31):
            return None
        table = self._file._chunks(self)
        counts = [-(-s // c) for s, c in zip(self.shape, lay["chunk"])]
        if len(table) != int(np.prod(counts)):
            return None                                    # chunks that were never written read as the fill value: host path
        if not table:
            return None
        # (one NumPy pass over the table instead of a Python loop per chunk: a month of hourly ERA5 is 10^5 chunks per variable)
        if isinstance(table, ChunkTable):
            addr, size, mask = table.a, table.b, table.c
        else:
            import itertools
            rec = np.fromiter(itertools.chain.from_iterable(table.values()), dtype=np.int64, count=3 * len(table)).reshape(-1, 3)    # address, stored size, filter mask
            addr, size, mask = rec[:, 0], rec[:, 1], rec[:, 2]
        if (size >= (1 << 28)).any():
            return None
        skipped = lambda fid: (mask & (1 << ids.index(fid))) != 0
        if (2 in ids and skipped(2).any()) or (3 in ids and skipped(3).any()):
            return None
        plain = skipped(1) if 1 in ids else np.ones(len(addr), dtype=bool)
        if isinstance(table, ChunkTable):
            out = table.with_values(addr + self._file.base, size - (4 if 3 in ids else 0), plain)
        else:
            out = dict(zip(table.keys(), zip((addr + self._file.base).tolist(), (size - (4 if 3 in ids else 0)).tolist(), plain.tolist())))
        return {"chunk": tuple(lay["chunk"]), "shuffle": 2 in ids, "fletcher32": 3 in ids, "table": out, "map": self._file._m}


class H5File:
    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        import mmap
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self.variables: Dict[str, H5Variable] = {}
        self.attrs: Dict[str, object] = {}
        self._parse()

    def close(self):
        self.variables.clear()
        try:
            self._m.close()
        except BufferError:
            pass
        self._f.close()

    def _refusal(self, how: str) -> "Hdf5Error":
        """A layout of HDF5 metadata this reader does not follow: says which object of which file, and what gets the user past it --
        there is no other HDF5 reader to fall back on (the image has no netCDF4 / h5py for the default interpreter)."""
        return Hdf5Error(f"{self.path}: the links or attributes of {getattr(self, '_context', 'an object')} are {how}, which this reader does "
                         "not follow.  Rewrite the file without that layout and run again: `nccopy -k cdf5 in.nc out.nc` (classic 64-bit-data "
                         "NetCDF: no HDF5 container at all; `-k nc3` for files below 2 GiB per variable) or `nccopy -k nc4 -d0 in.nc out.nc` "
                         "(NetCDF-4 rewritten by the netCDF library with its default, compact metadata)")

    # ---- primitive reads -------------------------------------------------------------------
    def _u(self, off: int, n: int) -> int:
        return int.from_bytes(self._m[off: off + n], "little")

    def _addr(self, off: int) -> int:
        return self._u(off, self.O)

    def _len(self, off: int) -> int:
        return self._u(off, self.L)

    # ---- superblock ------------------------------------------------------------------------
    def _parse(self):
        m = self._m
        base = 0
        while m[base: base + 8] != SIGNATURE:
            base = 512 if base == 0 else base * 2
            if base + 8 > len(m):
                raise Hdf5Error(f"{self.path}: not an HDF5 file")
        ver = m[base + 8]
        if ver in (0, 1):
            self.O, self.L = m[base + 13], m[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = self._addr(p)
            p += 4 * self.O                          # base, free-space info, end of file, driver info
            root = self._addr(p + self.O)            # symbol table entry: link name offset, object header address
        elif ver in (2, 3):
            self.O, self.L = m[base + 9], m[base + 10]
            p = base + 12
            self.base = self._addr(p)
            root = self._addr(p + 3 * self.O)
        else:
            raise Hdf5Error(f"superblock version {ver} not supported")
        if self.base not in (0, base):
            raise Hdf5Error("non-zero base address not supported")
        self.base = base
        self._context = "the root group"
        msgs = self._object_messages(root)
        self.attrs = self._attributes(msgs)
        links = self._group_links(msgs)
        by_addr = {}
        self.skipped: Dict[str, str] = {}              # datasets that are not numeric arrays (name -> kind): strings, compounds, references
        for name, addr in links.items():
            try:
                dm = self._object_messages(addr)
            except Hdf5Error:
                continue
            self._context = f"the dataset '{name}'"
            var = self._dataset(name, dm)
            if var is not None:
                self.variables[name] = var
                by_addr[addr] = name
        # dimension names from DIMENSION_LIST (one variable-length list of object references per axis)
        for var in self.variables.values():
            dl = var.attrs.pop("DIMENSION_LIST", None)
            if isinstance(dl, list) and len(dl) == var.ndim:
                names = []
                for refs in dl:
                    a = int(refs[0]) if len(refs) else None
                    names.append(by_addr.get(a, ""))
                var.dims = tuple(names)
            elif var.ndim == 1:
                var.dims = (var.name,)
            for k in ("REFERENCE_LIST", "CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates", "_nc3_strict", "_NCProperties"):
                var.attrs.pop(k, None)

    # ---- object headers --------------------------------------------------------------------
    def _object_messages(self, addr: int) -> List[Tuple[int, int, int]]:
        """(type, offset of data, size) of every header message of the object at `addr`."""
        m, a = self._m, addr + self.base
        out: List[Tuple[int, int, int]] = []
        if m[a: a + 4] == b"OHDR":
            if m[a + 4] != 2:
                raise Hdf5Error("object header version")
            flags = m[a + 5]
            p = a + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            nsz = 1 << (flags & 3)
            chunk = self._u(p, nsz)
            p += nsz
            blocks = [(p, chunk)]
            while blocks:
                p, size = blocks.pop(0)
                end = p + size
                while p + 4 <= end:
                    mtype, msize, _mflags = m[p], self._u(p + 1, 2), m[p + 3]
                    p += 4 + (2 if flags & 0x04 else 0)
                    if p + msize > end:
                        break
                    if mtype == 0x10:
                        ca, cl = self._addr(p) + self.base, self._len(p + self.O)
                        if m[ca: ca + 4] != b"OCHK":
                            raise Hdf5Error("bad continuation block")
                        blocks.append((ca + 4, cl - 8))
                    elif mtype != 0:
                        out.append((mtype, p, msize))
                    p += msize
            return out
        if m[a] != 1:
            raise Hdf5Error(f"no object header at {addr}")
        nmsg, hsize = self._u(a + 2, 2), self._u(a + 8, 4)
        blocks = [(a + 16, hsize)]
        while blocks and nmsg > 0:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and nmsg > 0:
                mtype, msize = self._u(p, 2), self._u(p + 2, 2)
                p += 8
                nmsg -= 1
                if mtype == 0x10:
                    blocks.append((self._addr(p) + self.base, self._len(p + self.O)))
                elif mtype != 0:
                    out.append((mtype, p, msize))
                p += msize
        return out

    # ---- groups ----------------------------------------------------------------------------
    def _group_links(self, msgs) -> Dict[str, int]:
        links: Dict[str, int] = {}
        for mtype, p, size in msgs:
            if mtype == 0x11:                                   # symbol table: v1 B-tree + local heap
                btree, heap = self._addr(p), self._addr(p + self.O)
                h = heap + self.base
                if self._m[h: h + 4] != b"HEAP":
                    raise Hdf5Error("bad local heap")
                data = self._addr(h + 8 + 2 * self.L) + self.base
                self._walk_group_btree(btree, data, links)
            elif mtype == 0x06:
                name, addr = self._link(p)
                if addr is not None:
                    links[name] = addr
            elif mtype == 0x02:                                 # link info: dense storage in a fractal heap
                flags = self._m[p + 1]
                q = p + 2 + (8 if flags & 1 else 0)
                fheap, btree = self._addr(q), self._addr(q + self.O)
                if fheap != (1 << (8 * self.O)) - 1:
                    for off in self._dense_objects(fheap, btree, 5):
                        name, addr = self._link(off)
                        if addr is not None:
                            links[name] = addr
        return links

    def _walk_group_btree(self, addr, heap_data, links):
        a = addr + self.base
        m = self._m
        if m[a: a + 4] == b"SNOD":
            n = self._u(a + 6, 2)
            p = a + 8
            for _ in range(n):
                noff, oaddr = self._addr(p), self._addr(p + self.O)
                s = heap_data + noff
                e = m.find(b"\0", s)
                links[m[s:e].decode("utf-8")] = oaddr
                p += 2 * self.O + 24
            return
        if m[a: a + 4] != b"TREE":
            raise Hdf5Error("bad group B-tree node")
        n = self._u(a + 6, 2)
        p = a + 8 + 2 * self.O
        for i in range(n):
            child = self._addr(p + self.L)
            self._walk_group_btree(child, heap_data, links)
            p += self.L + self.O

    def _link(self, off):
        """(name, object header address or None) of the link message at file offset `off`."""
        b = bytes(self._m[off: off + 1100])
        if len(b) < 4 or b[0] != 1:
            return "", None
        flags = b[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = b[p]; p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nl = 1 << (flags & 3)
        n = int.from_bytes(b[p: p + nl], "little"); p += nl
        name = b[p: p + n].decode("utf-8"); p += n
        if ltype != 0:
            return name, None
        return name, int.from_bytes(b[p: p + self.O], "little")

    # ---- dense storage: a v2 B-tree of names whose records carry fractal-heap IDs ----------------------------
    def _dense_objects(self, fheap: int, btree: int, rec_type: int) -> List[int]:
        """File offsets of the link (rec_type 5) or attribute (rec_type 8) messages of a dense-storage object."""
        locate = self._fractal_heap(fheap)
        out = []
        for rec in self._btree2_records(btree, rec_type):
            hid = rec[4:] if rec_type == 5 else rec[:8]        # type 5: hash (4) + heap ID (7); type 8: heap ID (8) + flags + order + hash
            pos = locate(hid)
            if pos is not None:
                out.append(pos)
        return out

    def _btree2_records(self, addr: int, rec_type: int):
        m, a = self._m, addr + self.base
        if m[a: a + 4] != b"BTHD":
            raise Hdf5Error("bad v2 B-tree header")
        if m[a + 5] != rec_type:
            raise Hdf5Error(f"v2 B-tree of type {m[a + 5]}, expected {rec_type}")
        node_size, rec_size, depth = self._u(a + 6, 4), self._u(a + 10, 2), self._u(a + 12, 2)
        root, nroot = self._addr(a + 16), self._u(a + 16 + self.O, 2)
        if root == (1 << (8 * self.O)) - 1 or nroot == 0:
            return
        max_leaf = (node_size - 10) // rec_size
        nrec_bytes = (max_leaf.bit_length() + 7) // 8

        def node(naddr, nrec, level):
            b = naddr + self.base
            sig = b"BTLF" if level == 0 else b"BTIN"
            if m[b: b + 4] != sig:
                raise Hdf5Error("bad v2 B-tree node")
            p = b + 6
            recs = [bytes(m[p + i * rec_size: p + (i + 1) * rec_size]) for i in range(nrec)]
            if level == 0:
                yield from recs
                return
            if level > 1:
                raise Hdf5Error("v2 B-trees deeper than 2 levels are not supported")
            p += nrec * rec_size
            for i in range(nrec + 1):
                child, cn = self._addr(p), self._u(p + self.O, nrec_bytes)
                p += self.O + nrec_bytes
                yield from node(child, cn, level - 1)
                if i < nrec:
                    yield recs[i]
        yield from node(root, nroot, depth)

    def _fractal_heap(self, addr: int):
        """Returns locate(heap_id_bytes) -> file offset of a managed object (None for other ID types)."""
        m, a = self._m, addr + self.base
        if m[a: a + 4] != b"FRHP":
            raise Hdf5Error("bad fractal heap")
        p = a + 5
        filt_len = self._u(p + 2, 2)
        p += 5 + 4                                     # heap ID length, filter length, flags, max size of managed objects
        p += self.L + self.O                           # next huge id, huge b-tree
        p += self.L + self.O                           # free space, free-space manager
        p += 4 * self.L                                # managed space, allocated, iterator offset, number of managed objects
        p += 4 * self.L                                # huge size/count, tiny size/count
        width = self._u(p, 2); p += 2
        start = self._len(p); p += self.L
        max_direct = self._len(p); p += self.L
        max_bits = self._u(p, 2); p += 2
        p += 2                                         # starting rows of the root indirect block
        root = self._addr(p); p += self.O
        cur_rows = self._u(p, 2)
        if filt_len:
            raise self._refusal("stored in a FILTERED fractal heap (compressed metadata)")
        off_bytes = (max_bits + 7) // 8
        max_direct_rows = (max_direct // start).bit_length() - 1 + 2

        def locate(hid: bytes):
            if (hid[0] >> 4) & 3 != 0:                 # tiny / huge objects: not used for links and attributes of this size
                return None
            off = int.from_bytes(hid[1: 1 + off_bytes], "little")
            if cur_rows == 0:
                return root + self.base + off
            b = root + self.base
            if m[b: b + 4] != b"FHIB":
                raise Hdf5Error("bad fractal heap indirect block")
            q = b + 5 + self.O + off_bytes
            row_off = 0
            for r in range(cur_rows):
                size = start if r < 2 else start << (r - 1)
                if off < row_off + width * size:
                    if r >= max_direct_rows:
                        raise self._refusal("stored in a fractal heap with NESTED indirect blocks (a very large number of attributes or links)")
                    c = (off - row_off) // size
                    child = self._addr(q + (r * width + c) * self.O)
                    return child + self.base + (off - row_off - c * size)
                row_off += width * size
            return None
        return locate

    # ---- datatype / dataspace ---------------------------------------------------------------
    def _datatype(self, p) -> _Dtype:
        m = self._m
        cls, ver = m[p] & 0x0F, m[p] >> 4
        bits = self._u(p + 1, 3)
        size = self._u(p + 4, 4)
        order = ">" if bits & 1 else "<"
        if cls == 0:
            return _Dtype("num", np.dtype(f"{order}{'i' if bits & 8 else 'u'}{size}"), size)
        if cls == 1:
            return _Dtype("num", np.dtype(f"{order}f{size}"), size)
        if cls == 3:
            return _Dtype("str", np.dtype(f"S{size}"), size)
        if cls == 7:
            return _Dtype("ref", np.dtype(f"<u{size}"), size)
        if cls == 9:
            base = self._datatype(p + 8)
            return _Dtype("vlen_str" if (bits & 0x0F) == 1 else "vlen", None, size, base)
        return _Dtype("other", None, size)

    def _dataspace(self, p) -> Tuple[int, ...]:
        m = self._m
        ver, rank = m[p], m[p + 1]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if m[p + 3] == 2:          # null dataspace
                return (0,)
            q = p + 4
        else:
            raise Hdf5Error("dataspace version")
        return tuple(self._len(q + i * self.L) for i in range(rank))

    # ---- attributes -------------------------------------------------------------------------
    def _attributes(self, msgs) -> Dict[str, object]:
        out: Dict[str, object] = {}
        for mtype, p, size in msgs:
            if mtype == 0x0C:
                self._attribute(p, out)
            elif mtype == 0x15:                                # attribute info: dense storage
                flags = self._m[p + 1]
                q = p + 2 + (2 if flags & 1 else 0)
                fheap, btree = self._addr(q), self._addr(q + self.O)
                if fheap != (1 << (8 * self.O)) - 1:
                    for off in self._dense_objects(fheap, btree, 8):
                        self._attribute(off, out)
        return out

    def _attribute(self, p, out):
        m = self._m
        ver = m[p]
        if ver not in (1, 2, 3):
            return
        ns, ds, ss = self._u(p + 2, 2), self._u(p + 4, 2), self._u(p + 6, 2)
        q = p + 8 + (1 if ver == 3 else 0)
        pad = (lambda x: (x + 7) & ~7) if ver == 1 else (lambda x: x)
        name = bytes(m[q: q + ns]).split(b"\0")[0].decode("utf-8")
        q += pad(ns)
        dt = self._datatype(q); q += pad(ds)
        shape = self._dataspace(q); q += pad(ss)
        n = int(np.prod(shape)) if shape else 1
        out[name] = self._decode(dt, q, n, shape)

    def _decode(self, dt: _Dtype, q: int, n: int, shape):
        m = self._m
        if dt.kind == "num":
            a = np.frombuffer(m[q: q + n * dt.size], dtype=dt.np_dtype).astype(dt.np_dtype.newbyteorder("="))
            return a[0].item() if not shape else a.reshape(shape)
        if dt.kind == "str":
            vals = [bytes(m[q + i * dt.size: q + (i + 1) * dt.size]).split(b"\0")[0].decode("utf-8", "replace") for i in range(n)]
            return vals[0] if not shape or n == 1 else vals
        if dt.kind in ("vlen_str", "vlen"):
            vals = []
            for i in range(n):
                e = q + i * (4 + self.O + 4)
                cnt, coll, idx = self._u(e, 4), self._addr(e + 4), self._u(e + 4 + self.O, 4)
                raw = self._global_heap_object(coll, idx)
                if dt.kind == "vlen_str":
                    vals.append(raw[:cnt].decode("utf-8", "replace"))
                else:
                    base = dt.base.np_dtype if dt.base is not None and dt.base.np_dtype is not None else np.dtype("u1")
                    vals.append(np.frombuffer(raw[: cnt * base.itemsize], dtype=base))
            if dt.kind == "vlen_str":
                return vals[0] if not shape or n == 1 else vals
            return vals
        if dt.kind == "ref":
            return np.frombuffer(m[q: q + n * dt.size], dtype=dt.np_dtype)
        return None

    def _global_heap_object(self, coll: int, idx: int) -> bytes:
        m, a = self._m, coll + self.base
        if m[a: a + 4] != b"GCOL":
            raise Hdf5Error("bad global heap collection")
        size = self._len(a + 8)
        p, end = a + 8 + self.L, a + size
        while p + 8 + self.L <= end:
            i, osz = self._u(p, 2), self._len(p + 8)
            if i == idx:
                return bytes(m[p + 8 + self.L: p + 8 + self.L + osz])
            if i == 0:
                break
            p += 8 + self.L + ((osz + 7) & ~7)
        raise Hdf5Error("global heap object not found")

    # ---- datasets ---------------------------------------------------------------------------
    def _dataset(self, name, msgs) -> Optional[H5Variable]:
        dt = shape = layout = None
        filters = []
        for mtype, p, size in msgs:
            if mtype == 0x03:
                dt = self._datatype(p)
            elif mtype == 0x01:
                shape = self._dataspace(p)
            elif mtype == 0x08:
                layout = self._layout(p)
            elif mtype == 0x0B:
                filters = self._filter_pipeline(p)
        if dt is None or shape is None or layout is None or dt.kind != "num":
            if dt is not None and shape is not None:           # e.g. the CDS's per-time string coordinate `expver`: nothing numeric to hand out
                self.skipped[name] = dt.kind
            return None
        var = H5Variable(name, tuple(shape), dt.np_dtype, self._attributes(msgs), (), self, layout, filters)
        for mtype, p, size in msgs:                            # fill value of chunks that were never written
            if mtype == 0x05:
                ver = self._m[p]
                q = None
                if ver in (1, 2) and (ver == 1 or self._m[p + 3]):
                    q = p + 4
                elif ver == 3 and self._m[p + 1] & 0x20:
                    q = p + 2
                if q is not None and self._u(q, 4) == dt.size:
                    var._cache["fill"] = np.frombuffer(self._m[q + 4: q + 4 + dt.size], dtype=dt.np_dtype)[0]
        return var

    def _layout(self, p) -> dict:
        m = self._m
        ver, cls = m[p], m[p + 1]
        if ver == 3:
            if cls == 0:
                n = self._u(p + 2, 2)
                return {"class": "compact", "offset": p + 4, "size": n}
            if cls == 1:
                return {"class": "contiguous", "addr": self._addr(p + 2), "size": self._len(p + 2 + self.O)}
            if cls == 2:
                nd = m[p + 2]
                btree = self._addr(p + 3)
                dims = [self._u(p + 3 + self.O + 4 * i, 4) for i in range(nd)]
                return {"class": "chunked", "index": "btree1", "addr": btree, "chunk": tuple(dims[:-1])}
        if ver == 4:
            if cls == 1:
                return {"class": "contiguous", "addr": self._addr(p + 2), "size": self._len(p + 2 + self.O)}
            if cls == 0:
                n = self._u(p + 2, 2)
                return {"class": "compact", "offset": p + 4, "size": n}
            if cls == 2:
                flags, nd, enc = m[p + 2], m[p + 3], m[p + 4]
                dims = [self._u(p + 5 + enc * i, enc) for i in range(nd)]
                q = p + 5 + enc * nd
                itype = m[q]; q += 1
                lay = {"class": "chunked", "chunk": tuple(dims[:-1]), "flags": flags}
                if itype == 1:
                    if flags & 2:
                        lay.update(index="single", fsize=self._len(q), fmask=self._u(q + self.L, 4), addr=self._addr(q + self.L + 4))
                    else:
                        lay.update(index="single", fsize=None, fmask=0, addr=self._addr(q))
                elif itype == 2:
                    lay.update(index="implicit", addr=self._addr(q))
                elif itype == 3:
                    lay.update(index="farray", addr=self._addr(q + 1))
                elif itype == 4:
                    lay.update(index="earray", addr=self._addr(q + 5))
                elif itype == 5:
                    lay.update(index="btree2", addr=self._addr(q + 6))
                else:
                    raise Hdf5Error(f"chunk index type {itype} not supported")
                return lay
        raise Hdf5Error(f"data layout version {ver} class {cls} not supported")

    def _filter_pipeline(self, p) -> list:
        m = self._m
        ver, n = m[p], m[p + 1]
        q = p + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = self._u(q, 2); q += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = self._u(q, 2); q += 2
            q += 2                                     # flags
            ncd = self._u(q, 2); q += 2
            q += ((nlen + 7) & ~7) if ver == 1 else nlen
            cd = [self._u(q + 4 * i, 4) for i in range(ncd)]
            q += 4 * ncd
            if ver == 1 and ncd % 2:
                q += 4
            out.append((fid, cd))
        return out

    def _chunks(self, var: H5Variable) -> Dict[Tuple[int, ...], Tuple[int, int, int]]:
        """chunk offset (element coordinates) -> (file address, stored size, filter mask)"""
        if "table" in var._cache:
            return var._cache["table"]
        lay = var._layout
        table: Dict[Tuple[int, ...], Tuple[int, int, int]] = {}
        rank = len(var.shape)
        chunk = lay["chunk"]
        nbytes = int(np.prod(chunk)) * var.dtype.itemsize
        if lay["index"] == "btree1" and self.O == 8:
            # a node's entries are fixed-size records (key: stored size, filter mask, rank + 1 offsets; then the child address):
            # one NumPy view per node instead of a Python loop per chunk (a 4096-step ERA5 variable has ~10^6 chunks)
            rec = np.dtype([("size", "<u4"), ("mask", "<u4"), ("offs", "<u8", (rank + 1,)), ("child", "<u8")])
            buf = np.frombuffer(self._m, dtype=np.uint8)

            leaves = []

            def walk(addr):
                a = addr + self.base
                m = self._m
                if m[a: a + 4] != b"TREE" or m[a + 4] != 1:
                    raise Hdf5Error("bad chunk B-tree node")
                level, n = m[a + 5], self._u(a + 6, 2)
                if n == 0:
                    return
                p = a + 8 + 2 * self.O
                e = buf[p: p + n * rec.itemsize].view(rec)
                if level == 0:
                    leaves.append(e)               # (views of the mapped file: nothing is copied until the concatenation below)
                else:
                    for child in e["child"].tolist():
                        walk(child)
            if lay["addr"] != (1 << (8 * self.O)) - 1:
                walk(lay["addr"])
            if leaves:
                e = np.concatenate(leaves)
                table = ChunkTable(e["offs"][:, :rank].astype(np.int64), e["child"].astype(np.int64), e["size"].astype(np.int64),
                                   e["mask"].astype(np.int64), chunk, var.shape)
        elif lay["index"] == "btree1":
            def walk(addr):
                a = addr + self.base
                m = self._m
                if m[a: a + 4] != b"TREE" or m[a + 4] != 1:
                    raise Hdf5Error("bad chunk B-tree node")
                level, n = m[a + 5], self._u(a + 6, 2)
                p = a + 8 + 2 * self.O
                ksz = 8 + 8 * (rank + 1)
                for i in range(n):
                    size, mask = self._u(p, 4), self._u(p + 4, 4)
                    offs = tuple(self._u(p + 8 + 8 * d, 8) for d in range(rank))
                    child = self._addr(p + ksz)
                    if level == 0:
                        table[offs] = (child, size, mask)
                    else:
                        walk(child)
                    p += ksz + self.O
            if lay["addr"] != (1 << (8 * self.O)) - 1:
                walk(lay["addr"])
        elif lay["index"] == "single":
            if lay["addr"] != (1 << (8 * self.O)) - 1:
                table[(0,) * rank] = (lay["addr"], lay["fsize"] if lay["fsize"] is not None else nbytes, lay["fmask"])
        elif lay["index"] == "btree2":                     # several unlimited dimensions: records carry the scaled chunk offsets
            a = lay["addr"] + self.base
            rtype = self._m[a + 5]
            if rtype not in (10, 11):
                raise Hdf5Error("unexpected v2 B-tree type for a chunk index")
            rec_size = self._u(a + 10, 2)
            csz = rec_size - self.O - 8 * rank - 4 if rtype == 11 else 0
            for rec in self._btree2_records(lay["addr"], rtype):
                caddr = int.from_bytes(rec[: self.O], "little")
                q = self.O
                size, mask = nbytes, 0
                if rtype == 11:
                    size = int.from_bytes(rec[q: q + csz], "little"); q += csz
                    mask = int.from_bytes(rec[q: q + 4], "little"); q += 4
                offs = tuple(int.from_bytes(rec[q + 8 * d: q + 8 * d + 8], "little") * chunk[d] for d in range(rank))
                table[offs] = (caddr, size, mask)
        elif lay["index"] == "earray":
            counts = [-(-s // c) for s, c in zip(var.shape, chunk)]
            coords = list(np.ndindex(*counts))
            for i, (caddr, csize, cmask) in enumerate(self._earray_elements(lay["addr"], len(coords), nbytes)):
                if caddr != (1 << (8 * self.O)) - 1:
                    table[tuple(ci * ch for ci, ch in zip(coords[i], chunk))] = (caddr, csize, cmask)
        else:
            counts = [-(-s // c) for s, c in zip(var.shape, chunk)]
            coords = list(np.ndindex(*counts))
            if lay["index"] == "implicit":
                for i, c in enumerate(coords if lay["addr"] != (1 << (8 * self.O)) - 1 else ()):
                    table[tuple(ci * ch for ci, ch in zip(c, chunk))] = (lay["addr"] + i * nbytes, nbytes, 0)
            else:                                      # fixed array
                a = lay["addr"] + self.base
                m = self._m
                if m[a: a + 4] != b"FAHD":
                    raise Hdf5Error("bad fixed array header")
                client, esize, pbits = m[a + 5], m[a + 6], m[a + 7]
                nent = self._len(a + 8)
                db = self._addr(a + 8 + self.L) + self.base
                if m[db: db + 4] != b"FADB":
                    raise Hdf5Error("bad fixed array data block")
                p = db + 6 + self.O
                undef = (1 << (8 * self.O)) - 1
                page_n = 1 << pbits
                if nent > page_n:
                    # paged data block: a bitmap of the initialised pages (most significant bit first) and the block's checksum,
                    # then the pages, each followed by its own checksum; pages that were never initialised hold no chunks
                    npages = -(-nent // page_n)
                    bitmap = bytes(m[p: p + (npages + 7) // 8])
                    p += (npages + 7) // 8 + 4
                    starts = []
                    for pg in range(npages):
                        n_here = min(page_n, nent - pg * page_n)
                        live = bool(bitmap[pg // 8] & (0x80 >> (pg % 8)))
                        starts.extend([(p + e * esize) if live else None for e in range(n_here)])
                        p += n_here * esize + 4
                else:
                    starts = [p + e * esize for e in range(nent)]
                for i, c in enumerate(coords[:nent]):
                    q = starts[i]
                    if q is None or self._addr(q) == undef:            # a chunk that was never written reads as the fill value
                        continue
                    off = tuple(ci * ch for ci, ch in zip(c, chunk))
                    if client == 0:
                        table[off] = (self._addr(q), nbytes, 0)
                    else:
                        csz = esize - self.O - 4
                        table[off] = (self._addr(q), self._u(q + self.O, csz), self._u(q + self.O + csz, 4))
        var._cache["table"] = table
        return table

    def _earray_elements(self, addr: int, want: int, chunk_bytes: int):
        """The first `want` chunk records (address, stored size, filter mask) of an extensible-array chunk index
        (one unlimited dimension, written with libver >= v110)."""
        m, a = self._m, addr + self.base
        if m[a: a + 4] != b"EAHD":
            raise Hdf5Error("bad extensible array header")
        client, esize, max_bits, idx_elmts, dblk_min, sblk_min_ptrs, page_bits = (m[a + 5 + i] for i in range(7))
        iblk = self._addr(a + 12 + 6 * self.L) + self.base
        undef = (1 << (8 * self.O)) - 1
        off_size = (max_bits + 7) // 8
        page_n = 1 << page_bits

        def element(p):
            if client == 0:
                return self._addr(p), chunk_bytes, 0
            csz = esize - self.O - 4
            return self._addr(p), self._u(p + self.O, csz), self._u(p + self.O + csz, 4)

        out = []
        if m[iblk: iblk + 4] != b"EAIB":
            raise Hdf5Error("bad extensible array index block")
        p = iblk + 6 + self.O
        for _ in range(idx_elmts):
            out.append(element(p)); p += esize
        nsblks_total = 1 + (max_bits - (dblk_min.bit_length() - 1))
        iblk_nsblks = 2 * (sblk_min_ptrs.bit_length() - 1)
        ndblk_addrs = 2 * (sblk_min_ptrs - 1)
        dblk_addrs = [self._addr(p + i * self.O) for i in range(ndblk_addrs)]
        p += ndblk_addrs * self.O
        sblk_addrs = [self._addr(p + i * self.O) for i in range(max(nsblks_total - iblk_nsblks, 0))]

        def data_block(daddr, nel):
            if daddr == undef:
                out.extend([(undef, 0, 0)] * nel)
                return
            b = daddr + self.base
            if m[b: b + 4] != b"EADB":
                raise Hdf5Error("bad extensible array data block")
            q = b + 6 + self.O + off_size
            if nel <= page_n:
                for _ in range(nel):
                    out.append(element(q)); q += esize
            else:                                      # paged: the pages follow the block's own checksum, each with a checksum
                q += 4
                for _ in range(nel // page_n):
                    for _ in range(page_n):
                        out.append(element(q)); q += esize
                    q += 4

        di = 0
        for u in range(nsblks_total):
            if len(out) >= want:
                break
            ndblks, nel = 1 << (u // 2), (1 << ((u + 1) // 2)) * dblk_min
            if u < iblk_nsblks:
                for _ in range(ndblks):
                    if len(out) >= want:
                        break
                    data_block(dblk_addrs[di], nel); di += 1
            else:
                saddr = sblk_addrs[u - iblk_nsblks]
                if saddr == undef:
                    out.extend([(undef, 0, 0)] * (ndblks * nel))
                    continue
                b = saddr + self.base
                if m[b: b + 4] != b"EASB":
                    raise Hdf5Error("bad extensible array super block")
                q = b + 6 + self.O + off_size
                if nel > page_n:                       # page-initialised bitmaps, one per data block
                    q += ndblks * (((nel // page_n) + 7) // 8)
                for i in range(ndblks):
                    if len(out) >= want:
                        break
                    data_block(self._addr(q + i * self.O), nel)
        return out[:want]

    def _read_chunk(self, var: "H5Variable", addr: int, size: int, mask: int) -> np.ndarray:
        raw = bytes(self._m[addr + self.base: addr + self.base + size])
        for i, (fid, cd) in reversed(list(enumerate(var._filters))):
            if mask & (1 << i):
                continue
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:                             # shuffle
                es = cd[0] if cd else var.dtype.itemsize
                n = len(raw) // es
                out = np.empty((n, es), dtype=np.uint8)
                np.copyto(out, np.frombuffer(raw, dtype=np.uint8)[: n * es].reshape(es, n).T)     # array assignment releases the GIL (.tobytes() does not:
                raw = out.reshape(-1).data                                                          # the inflate threads queued for it)
            elif fid == 3:                             # fletcher32: the payload's checksum follows it (4 bytes, little-endian)
                raw, stored = raw[:-4], int.from_bytes(raw[-4:], "little")
                if stored not in _fletcher32(raw):
                    raise Hdf5Error(f"fletcher32 checksum mismatch in a chunk of '{var.name}' at file offset {addr + self.base}: the file is corrupt")
            else:
                raise Hdf5Error(f"HDF5 filter {fid} not supported (deflate, shuffle and fletcher32 are)")
        return np.frombuffer(raw, dtype=var.dtype)

    def _read_dataset(self, var: H5Variable, t: Optional[int], axis1=None) -> np.ndarray:
        """``axis1``: with a time step ``t`` -- only the chunks that hold these indices of axis 1 are read (the rest of the returned
        block is fill)."""
        lay = var._layout
        shape = var.shape
        if lay["class"] in ("contiguous", "compact"):
            n = int(np.prod(shape))
            if lay["class"] == "compact":
                a = np.frombuffer(self._m[lay["offset"]: lay["offset"] + n * var.dtype.itemsize], dtype=var.dtype)
            elif lay["addr"] == (1 << (8 * self.O)) - 1:
                a = np.zeros(n, dtype=var.dtype)
            else:
                a = np.frombuffer(self._m, dtype=var.dtype, count=n, offset=lay["addr"] + self.base)
            a = a.reshape(shape)
            return a if t is None else a[t]
        chunk = lay["chunk"]
        table = self._chunks(var)
        if t is None:
            out = np.zeros(shape, dtype=var.dtype)
            lo0, hi0 = 0, shape[0]
        else:
            out = np.zeros((1,) + shape[1:], dtype=var.dtype)
            lo0, hi0 = t, t + 1
        fill = var._cache.get("fill")        # the dataset's HDF5 fill value (netCDF-C sets it to _FillValue); none defined: zeros, like the library
        if fill is not None:
            out[...] = fill
        if "by_t" not in var._cache:                           # chunks grouped by their first-axis offset: a time-step read looks at its own only
            by_t = {}
            for offs, loc in table.items():
                by_t.setdefault(offs[0], []).append((offs, loc))
            var._cache["by_t"] = by_t
        by_t = var._cache["by_t"]
        first = (lo0 // chunk[0]) * chunk[0]
        need = [item for o0 in range(first, hi0, chunk[0]) for item in by_t.get(o0, ())]
        if axis1 is not None and t is not None and len(shape) >= 2:
            wanted = {k // chunk[1] for k in axis1}
            need = [item for item in need if item[0][1] // chunk[1] in wanted]
        if t is None:                                          # a whole-variable read is used once: nothing to keep (the cache would hold
            missing, keep = need, False                        # the variable a second time)
        else:
            missing, keep = [(offs, loc) for offs, loc in need if ("chunk", offs) not in var._cache], True
        if keep and len(var._cache) + len(missing) > 256:               # bounded chunk cache (time-step reads revisit chunks that span steps)
            for k in [k for k in var._cache if isinstance(k, tuple) and k and k[0] == "chunk"]:
                del var._cache[k]
            missing = need
        inflate = lambda item: (item[0], self._read_chunk(var, *item[1]).reshape(chunk))
        if len(missing) > 1 and var._filters:
            results = list(_inflate_pool().map(inflate, missing))          # zlib releases the GIL
        else:
            results = [inflate(item) for item in missing]
        fresh = dict(results)
        if keep:
            for offs, data in results:
                var._cache[("chunk", offs)] = data
        for offs, _loc in need:
            data = fresh[offs] if offs in fresh else var._cache[("chunk", offs)]
            src, dst = [], []
            for d in range(len(shape)):
                a0 = max(offs[d], lo0 if d == 0 else 0)
                a1 = min(offs[d] + chunk[d], hi0 if d == 0 else shape[d])
                src.append(slice(a0 - offs[d], a1 - offs[d]))
                dst.append(slice(a0 - (lo0 if d == 0 else 0), a1 - (lo0 if d == 0 else 0)))
            out[tuple(dst)] = data[tuple(src)]
        return out if t is None else out[0]


def _fletcher32(data: bytes):
    """HDF5's H5_checksum_fletcher32 of `data` (16-bit big-endian words, ones'-complement sums) and its byte-swapped twin,
    which the library also accepts on read (H5Zfletcher32.c: files written by 1.6.2 on little-endian hosts)."""
    n = len(data) // 2
    w = np.frombuffer(data, dtype=">u2", count=n).astype(np.uint64)
    if len(data) % 2:
        w = np.append(w, np.uint64(data[-1] << 8))
    fold = lambda x: 0 if x == 0 else (x - 1) % 65535 + 1
    s1 = s2 = 0
    for a in range(0, w.size, 1 << 20):               # blocks keep the weighted sums far inside 64 bits
        blk = w[a: a + (1 << 20)]
        m = blk.size
        s2 += s1 * m + int((blk * np.arange(m, 0, -1, dtype=np.uint64)).sum())
        s1 += int(blk.sum())
    c = (fold(s2) << 16) | fold(s1)
    swapped = ((c & 0x00FF00FF) << 8) | ((c >> 8) & 0x00FF00FF)
    return (c, swapped)


_POOL = None


def _inflate_pool():
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1)), thread_name_prefix="h5-inflate")
    return _POOL


def is_hdf5(path: str) -> bool:
    with open(path, "rb") as f:
        return f.read(8) == SIGNATURE
