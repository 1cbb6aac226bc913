#!/opt/conda/bin/python3.9
"""(soak helper; needs h5py + scipy: the image's conda interpreter)  Rewrites classic NetCDF files as NetCDF-4 / HDF5 with the SAME stored
values and attributes, in a random layout per file: chunk shapes that need not divide the extents, shuffle, deflate 1-9, fletcher32,
some variables chunked without deflate or contiguous (so one file mixes the ingest's stagers), random library version bounds.

    /opt/conda/bin/python3.9 tools/classic_to_nc4.py <seed> <in1.nc> <out1.nc> [<in2.nc> <out2.nc> ...]"""
import sys

import h5py
import numpy as np
from scipy.io import netcdf_file


def convert(src, dst, rng):
    nc = netcdf_file(src, mmap=False)
    libver = [("earliest", "latest"), ("latest", "latest"), ("earliest", "v108"), ("v110", "v110"), ("earliest", "v110")][int(rng.integers(0, 5))]
    note = []
    with h5py.File(dst, "w", libver=libver, track_order=bool(rng.random() < 0.5)) as h:
        dims = list(nc.dimensions)
        for d in dims:
            v = nc.variables[d]
            ds = h.create_dataset(d, data=np.array(v.data).astype(v.data.dtype.newbyteorder("=")))
            ds.make_scale(d)
            for k, a in v._attributes.items():
                ds.attrs[k] = a.decode() if isinstance(a, bytes) else a
        for name, v in nc.variables.items():
            if name in dims:
                continue
            a = np.array(v.data).astype(v.data.dtype.newbyteorder("="))
            kind = float(rng.random())
            kw = {}
            if kind < 0.12:
                how = "contiguous"
            else:
                chunk = tuple(int(rng.integers(1, s + 1)) if rng.random() < 0.7 else int(s) for s in a.shape)
                if rng.random() < 0.6:
                    chunk = (1,) + chunk[1:]                                   # the usual: one time step per chunk
                kw = dict(chunks=chunk)
                if kind >= 0.25:
                    kw.update(compression="gzip", compression_opts=int(rng.integers(1, 10)))
                if rng.random() < 0.7:
                    kw["shuffle"] = True
                if rng.random() < 0.3:
                    kw["fletcher32"] = True
                how = "chunks %s %s" % (chunk, "+".join(k for k in ("shuffle", "compression", "fletcher32") if k in kw))
            ds = h.create_dataset(name, data=a, **kw)
            for k, att in v._attributes.items():
                ds.attrs[k] = att.decode() if isinstance(att, bytes) else att
            for i, d in enumerate(v.dimensions):
                ds.dims[i].attach_scale(h[d])
            note.append("%s: %s" % (name, how))
    nc.close()
    return "libver %s; %s" % ("/".join(libver), "; ".join(note))


if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[1]))
    pairs = sys.argv[2:]
    for i in range(0, len(pairs), 2):
        print(pairs[i + 1], "|", convert(pairs[i], pairs[i + 1], rng), flush=True)
