#!/usr/bin/env python3
"""Randomised soak of lorenzcycletoolkit_amd/hdf5_lite.py against h5py (BUILD CONTAINER ONLY: h5py exists only for
/opt/conda/bin/python3.9 there; neither on the GPU box nor for the default interpreter -- hence a tool, not a test).

    python tools/soak_hdf5.py --cases 300 --seed 1

Two processes: this file re-runs itself under the conda interpreter with ``--write`` to create the files with h5py (random library
version bounds, creation-order tracking, 3-14 root objects and 0-12 attributes per object so that compact AND dense link / attribute
storage occur, dataset types i1..i8 / u1..u4 / f4 / f8 in either byte order, contiguous / compact / chunked layouts with chunk
shapes that do not divide the extents, unlimited dimensions, deflate levels, shuffle, fletcher32, fill values with never-written
chunks, dimension scales) plus an .npz of what h5py reads back; the default interpreter then reads every file with hdf5_lite and
compares shapes, dtypes, values (whole arrays and per-index reads along axis 0), attributes and dimension names."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA = "/opt/conda/bin/python3.9"


# ---------------------------------------------------------------------------------------------------------------------
# writer (conda interpreter, h5py)
# ---------------------------------------------------------------------------------------------------------------------
def random_attr(rng):
    k = int(rng.integers(0, 8))
    if k == 0:
        return np.float64(rng.standard_normal())
    if k == 1:
        return np.float32(rng.standard_normal())
    if k == 2:
        return np.int16(rng.integers(-30000, 30000))
    if k == 3:
        return np.int32(rng.integers(-10 ** 9, 10 ** 9))
    if k == 4:
        return rng.standard_normal(int(rng.integers(1, 5)))
    if k == 5:
        return "text %d" % int(rng.integers(0, 1000))                       # variable-length string
    if k == 6:
        return np.bytes_("fixed %d" % int(rng.integers(0, 1000)))           # fixed-length string
    return np.array(rng.integers(0, 100, int(rng.integers(1, 4))), dtype=np.int64)


def write_files(outdir, cases, seed):
    import h5py
    rng = np.random.default_rng(seed)
    manifest = []
    for c in range(cases):
        path = os.path.join(outdir, "case%d.h5" % c)
        libver = [("earliest", "latest"), ("latest", "latest"), ("earliest", "v108"), ("v108", "v108"), ("v110", "v110"),
                  ("earliest", "v110")][int(rng.integers(0, 6))]
        track = bool(rng.random() < 0.5)
        nt, nl = int(rng.integers(1, 7)), int(rng.integers(1, 6))
        ny, nx = int(rng.integers(2, 24)), int(rng.integers(2, 40))
        if rng.random() < 0.15:           # a long axis of tiny chunks: deep B-trees, paged fixed arrays, extensible-array super blocks
            nt, nl, ny, nx = int(rng.integers(200, 5000)), int(rng.integers(1, 3)), int(rng.integers(2, 4)), int(rng.integers(2, 5))
        expect = {}
        info = {"path": path, "libver": libver, "track": track, "vars": {}}
        with h5py.File(path, "w", libver=libver, track_order=track) as f:
            for k in range(int(rng.integers(0, 10))):
                f.attrs["g%d" % k] = random_attr(rng)
            dims = {"time": nt, "level": nl, "latitude": ny, "longitude": nx}
            for name, n in dims.items():
                unlimited = name == "time" and rng.random() < 0.5
                dt = str(rng.choice(["<f4", "<f8", "<i4", ">f4", "<i8"]))
                d = f.create_dataset(name, (n,), dtype=dt, maxshape=(None,) if unlimited else None,
                                     chunks=(int(rng.integers(1, 9)),) if unlimited else None, track_order=track)
                vals = np.sort(rng.standard_normal(n) * 50).astype(dt) if dt[1] == "f" else np.arange(n).astype(dt) * int(rng.integers(1, 7))
                d[:] = vals
                d.make_scale(name)
                if rng.random() < 0.7:
                    d.attrs["units"] = str(rng.choice(["hPa", "degrees_north", "hours since 1900-01-01 00:00:00.0", "Pa"]))
                expect[name] = vals
                info["vars"][name] = {"dims": [name]}
            nvar = int(rng.integers(1, 10))
            for v in range(nvar):
                name = "var%d" % v
                rank = int(rng.choice([4, 4, 4, 3, 2]))
                dnames = ["time", "level", "latitude", "longitude"][4 - rank:] if rng.random() < 0.8 else ["time", "latitude", "longitude", "level"][:rank]
                shape = tuple(dims[n] for n in dnames)
                dt = str(rng.choice(["<i2", "<i2", "<f4", "<f8", "<i4", ">i2", ">f4", ">f8", "<i1", "<u1", "<u2", "<i8", "<u4"]))
                layout = str(rng.choice(["contiguous", "chunked", "chunked", "chunked", "compact"]))
                kw = {}
                if layout == "chunked":
                    kw["chunks"] = tuple(int(rng.integers(1, min(s, 8) + 1)) if s > 100 else int(rng.integers(1, s + 1)) for s in shape)
                    if dnames[0] == "time" and f["time"].maxshape[0] is None:
                        kw["maxshape"] = (None,) + shape[1:]
                    if rng.random() < 0.7:
                        kw["compression"] = "gzip"; kw["compression_opts"] = int(rng.integers(1, 10))
                    if rng.random() < 0.6:
                        kw["shuffle"] = True
                    if rng.random() < 0.25:
                        kw["fletcher32"] = True
                if rng.random() < 0.5:
                    kw["fillvalue"] = np.dtype(dt).type(-99 if np.dtype(dt).kind != "u" else 250)
                if np.dtype(dt).kind == "f":
                    a = (rng.standard_normal(shape) * 100).astype(dt)
                else:
                    ii = np.iinfo(np.dtype(dt))
                    a = rng.integers(max(ii.min, -2 ** 40), min(ii.max, 2 ** 40), shape, endpoint=True).astype(dt)
                if layout == "compact":
                    if a.nbytes > 60000:
                        layout = "contiguous"
                    else:
                        # h5py has no keyword for the compact layout: go through the low-level API
                        space = h5py.h5s.create_simple(shape)
                        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
                        dcpl.set_layout(h5py.h5d.COMPACT)
                        tid = h5py.h5t.py_create(np.dtype(dt))
                        did = h5py.h5d.create(f.id, name.encode(), tid, space, dcpl)
                        d = h5py.Dataset(did)
                        d[...] = a
                if layout != "compact":
                    d = f.create_dataset(name, shape, dtype=dt, track_order=track, **kw)
                    partial = layout == "chunked" and rng.random() < 0.3
                    if partial:                              # leave some chunks unwritten: they read as the fill value
                        sl = tuple(slice(0, max(1, int(rng.integers(1, s + 1)))) for s in shape)
                        d[sl] = a[sl]
                    else:
                        d[...] = a
                for axis, n in enumerate(dnames):
                    d.dims[axis].attach_scale(f[n])
                for k in range(int(rng.integers(0, 13))):
                    d.attrs["a%d" % k] = random_attr(rng)
                if rng.random() < 0.5:
                    d.attrs["scale_factor"] = np.float64(rng.random() + 0.1)
                    d.attrs["add_offset"] = np.float64(rng.standard_normal())
                    d.attrs["_FillValue"] = np.array([-32767]).astype(dt)
                info["vars"][name] = {"dims": dnames, "dtype": dt, "layout": layout, "kw": {k: str(x) for k, x in kw.items()}}
        with h5py.File(path, "r") as f:                      # what h5py reads back is the expectation
            arrays, attrs = {}, {"/": {}}
            for k, x in f.attrs.items():
                attrs["/"][k] = x
            for name in f:
                arrays[name] = f[name][...]
                attrs[name] = {k: x for k, x in f[name].attrs.items() if k not in ("DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME")}
        flat = {}
        for name, a in arrays.items():
            flat["data/" + name] = a
        for obj, d in attrs.items():
            for k, x in d.items():
                if isinstance(x, (str, bytes)):
                    x = np.array(x if isinstance(x, str) else x.decode())
                flat["attr/%s/%s" % (obj, k)] = np.asarray(x)
        np.savez(path + ".npz", **flat)
        manifest.append(info)
    with open(os.path.join(outdir, "manifest.json"), "w") as fh:
        json.dump(manifest, fh)


# ---------------------------------------------------------------------------------------------------------------------
# reader (default interpreter, hdf5_lite)
# ---------------------------------------------------------------------------------------------------------------------
def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind in "fc":
        return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())
    return bool((a == b).all())


def emulate_device_path(v, info):
    """lec_inflate + lec_chunk_scatter on the CPU: zlib for each stream of ``chunk_streams()``, the byte planes of the shuffle filter
    undone, the chunk cut to the variable's extent."""
    import zlib
    chunk, es = info["chunk"], v.dtype.itemsize
    n_elem = int(np.prod(chunk))
    out = np.zeros(v.shape, dtype=v.dtype)
    mm = np.frombuffer(info["map"], dtype=np.uint8)
    for org, (addr, size, plain) in info["table"].items():
        raw = bytes(mm[addr: addr + size])
        data = raw if plain else zlib.decompress(raw)
        if len(data) != n_elem * es:
            return np.zeros(0)
        b = np.frombuffer(data, dtype=np.uint8)
        if info["shuffle"] and es > 1:
            b = b.reshape(es, n_elem).T.copy().reshape(-1)
        blk = b.view(v.dtype).reshape(chunk)
        sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(org, chunk, v.shape))
        out[sl] = blk[tuple(slice(0, x.stop - x.start) for x in sl)]
    return out


def check_files(outdir):
    sys.path.insert(0, ROOT)
    from lorenzcycletoolkit_amd import hdf5_lite as h5
    with open(os.path.join(outdir, "manifest.json")) as fh:
        manifest = json.load(fh)
    fails = []
    for c, info in enumerate(manifest):
        what = "case %d (libver %s track %s)" % (c, "/".join(info["libver"]), info["track"])
        exp = np.load(info["path"] + ".npz", allow_pickle=False)
        try:
            f = h5.H5File(info["path"])
        except Exception as e:
            fails.append("%s: open raised %r; vars %s" % (what, e, info["vars"]))
            continue
        try:
            for key in exp.files:
                kind, rest = key.split("/", 1)
                if kind == "data":
                    name = rest
                    vi = info["vars"].get(name, {})
                    if name not in f.variables:
                        fails.append("%s: variable %s missing (%s)" % (what, name, vi))
                        continue
                    v = f.variables[name]
                    want = exp[key]
                    try:
                        got = v.read()
                    except Exception as e:
                        fails.append("%s: %s.read() raised %r (%s)" % (what, name, e, vi))
                        continue
                    if tuple(v.shape) != want.shape or got.shape != want.shape:
                        fails.append("%s: %s shape %s / %s, h5py %s (%s)" % (what, name, v.shape, got.shape, want.shape, vi))
                    elif np.dtype(got.dtype).newbyteorder("=") != want.dtype.newbyteorder("="):
                        fails.append("%s: %s dtype %s, h5py %s (%s)" % (what, name, got.dtype, want.dtype, vi))
                    elif not same(got, want):
                        fails.append("%s: %s values differ in %d of %d (%s)" % (what, name, int((np.asarray(got) != want).sum()), want.size, vi))
                    else:
                        for t in range(want.shape[0]):
                            if not same(v[t], want[t]):
                                fails.append("%s: %s[%d] differs (%s)" % (what, name, t, vi))
                                break
                        if want.ndim >= 2 and want.shape[0]:
                            t, ks = want.shape[0] // 2, sorted({0, want.shape[1] - 1, want.shape[1] // 2})
                            if not same(v.read_step(t, ks), want[t][ks]):
                                fails.append("%s: %s.read_step(%d, %s) differs (%s)" % (what, name, t, ks, vi))
                        # what the device path would do with chunk_streams(): every stream inflated (or copied), un-shuffled, put in place
                        streams = v.chunk_streams()
                        if streams is not None:
                            emu = emulate_device_path(v, streams)
                            if emu is not None and not same(emu, want):
                                fails.append("%s: %s rebuilt from chunk_streams() differs (%s)" % (what, name, vi))
                    if tuple(v.dims) != tuple(vi.get("dims", v.dims)):
                        fails.append("%s: %s dims %s, written %s" % (what, name, v.dims, vi.get("dims")))
                else:
                    obj, aname = rest.split("/", 1) if not rest.startswith("//") else ("/", rest[2:])
                    attrs = f.attrs if obj == "/" else (f.variables[obj].attrs if obj in f.variables else None)
                    if attrs is None:
                        continue
                    if aname in ("_Netcdf4Dimid", "_NCProperties"):
                        continue
                    if aname not in attrs:
                        fails.append("%s: attribute %s of %s missing" % (what, aname, obj))
                        continue
                    g, w = attrs[aname], exp[key]
                    if w.dtype.kind in "US":
                        ok = (g.decode() if isinstance(g, bytes) else str(g)) == str(w)
                    else:
                        ok = same(np.asarray(g).reshape(-1), w.reshape(-1))
                    if not ok:
                        fails.append("%s: attribute %s of %s is %r, h5py %r" % (what, aname, obj, g, w))
        except Exception as e:
            import traceback
            fails.append("%s: checking raised %r at %s" % (what, e, " | ".join(x.strip() for x in traceback.format_exc().splitlines()[-5:-1])))
        f.close()
    return fails, len(manifest)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--write", help="(internal) write the files into this directory with h5py and exit")
    ap.add_argument("--keep", help="directory to keep the files in (default: a temporary one)")
    a = ap.parse_args()
    if a.write:
        write_files(a.write, a.cases, a.seed)
        return
    if not os.path.exists(CONDA):
        sys.exit("needs %s with h5py (the build container)" % CONDA)
    t0 = time.time()
    ctx = tempfile.TemporaryDirectory() if not a.keep else None
    out = a.keep or ctx.name
    os.makedirs(out, exist_ok=True)
    subprocess.run([CONDA, os.path.abspath(__file__), "--write", out, "--cases", str(a.cases), "--seed", str(a.seed)], check=True)
    fails, n = check_files(out)
    for ln in fails[:60]:
        print("FAIL", ln[:1500])
    print("hdf5 soak: %d files, seed %d: %d failures, %.0f s" % (n, a.seed, len(fails), time.time() - t0))
    if ctx:
        ctx.cleanup()
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
