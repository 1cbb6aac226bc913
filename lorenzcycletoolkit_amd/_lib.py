"""ctypes binding of liblec_hip.so (the C ABI declared in include/lec_hip.h).

The product path has no CPU fallback: if the HIP library is missing, loading raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LEC_LIB: alternative build of the same ABI (kernel experiments only)
LIB_PATH = os.environ.get("LEC_LIB") or os.path.join(_HERE, "liblec_hip.so")

LEC_ABI_VERSION = 10
LEC_NSTAT = 32
LEC_NLEVRAW = 40
LEC_NSCALAR = 16
LEC_NLEVTAB = 21
LEC_NLEVFUN = 28
LEC_F64, LEC_F32, LEC_I16, LEC_I32, LEC_I8 = 0, 1, 2, 3, 4

# enum lec_kernel / enum lec_order (include/lec_hip.h)
KERNEL_AUTO, KERNEL_TWO_SWEEP, KERNEL_ROW_SWEEP, KERNEL_ROW_BLOCK, KERNEL_BOX_TILE, KERNEL_BOX_PLANE = 0, 1, 2, 3, 4, 5
ORDER_AUTO, ORDER_MEMORY, ORDER_XCD_LAT, ORDER_XCD_TILED = 0, 1, 2, 7



def source_digest() -> str:
    """sha256 (first 16 hex digits) over the kernel sources the library is built from (csrc/*.hip|.h|.inc + include/lec_hip.h,
    names and contents, sorted): stored with a profile so that a later run can tell whether a stored counter measurement was
    taken on the kernels it runs (the GPU box has the sources but no git history)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h"))
                   + glob.glob(os.path.join(_HERE, "csrc", "*.inc")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "lec_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


EXPORTS = ["lec_version", "lec_last_error", "lec_max_row", "lec_rowstats", "lec_reduce", "lec_dropmask", "lec_ingest", "lec_track_diag",
           "lec_check_boxes", "lec_check_maps", "lec_host_register", "lec_host_unregister", "lec_copy_rows_async",
           "lec_inflate", "lec_inflate_status_text", "lec_chunk_scatter", "lec_format_csv_rows", "lec_dtdt"]


class Tuning(C.Structure):
    """struct lec_tuning (include/lec_hip.h): kernel selection of lec_rowstats; all zero = library defaults."""
    _fields_ = [
        ("kernel", C.c_int32), ("block_shape", C.c_int32), ("order", C.c_int32),
        ("tile_t", C.c_int32), ("tile_j", C.c_int32), ("f32_vec", C.c_int32), ("reserved", C.c_int32 * 2),
    ]


class RowstatsArgs(C.Structure):
    """struct lec_rowstats_args (include/lec_hip.h)."""
    _fields_ = [
        ("tair_d", C.c_void_p), ("u_d", C.c_void_p), ("v_d", C.c_void_p), ("omega_d", C.c_void_p),
        ("geopt_d", C.c_void_p), ("dTdt_d", C.c_void_p),
        ("dtype", C.c_int32), ("with_q", C.c_int32),
        ("nt", C.c_int32), ("nl", C.c_int32), ("ny", C.c_int32), ("nx", C.c_int32),
        ("t_begin", C.c_int32), ("t_count", C.c_int32),
        ("n_box", C.c_int32), ("nxb_max", C.c_int32), ("nyb_max", C.c_int32), ("lon_uniform", C.c_int32),
        ("box_per_step", C.c_int32), ("reserved0", C.c_int32),
        ("box_d", C.c_void_p), ("boxtab_d", C.c_void_p), ("wlon_d", C.c_void_p), ("glon_d", C.c_void_p),
        ("lattab_d", C.c_void_p), ("levtab_d", C.c_void_p), ("tcoef_d", C.c_void_p),
        ("rows_d", C.c_void_p), ("stream", C.c_void_p),
        ("tuning", Tuning),
        ("tm_d", C.c_void_p), ("tp_d", C.c_void_p),
    ]


class ReduceArgs(C.Structure):
    """struct lec_reduce_args (include/lec_hip.h)."""
    _fields_ = [
        ("rows_d", C.c_void_p),
        ("t_count", C.c_int32), ("nl", C.c_int32), ("n_box", C.c_int32), ("nyb_max", C.c_int32),
        ("box_d", C.c_void_p), ("boxtab2_d", C.c_void_p), ("lattab2_d", C.c_void_p), ("levtab2_d", C.c_void_p),
        ("phi_scale", C.c_double),
        ("drop_any_time", C.c_int32), ("stage", C.c_int32), ("dropmask_d", C.c_void_p),
        ("am_d", C.c_void_p), ("levraw_d", C.c_void_p), ("scalars_d", C.c_void_p), ("levels_d", C.c_void_p),
        ("nanflag_d", C.c_void_p), ("stream", C.c_void_p),
        ("scalars_stride", C.c_int64), ("levels_stride", C.c_int64),
    ]


class IngestArgs(C.Structure):
    """struct lec_ingest_args (include/lec_hip.h)."""
    _fields_ = [
        ("src_d", C.c_void_p), ("src_dtype", C.c_int32), ("swap_bytes", C.c_int32),
        ("nt", C.c_int32), ("nl_in", C.c_int32), ("ny_in", C.c_int32), ("nx_in", C.c_int32),
        ("nl", C.c_int32), ("ny", C.c_int32), ("nx", C.c_int32),
        ("kmap_d", C.c_void_p), ("jmap_d", C.c_void_p), ("imap_d", C.c_void_p),
        ("has_packing", C.c_int32), ("has_fill", C.c_int32),
        ("scale_factor", C.c_double), ("add_offset", C.c_double), ("fill_value", C.c_double), ("unit_scale", C.c_double),
        ("out_dtype", C.c_int32), ("decode_dtype", C.c_int32),
        ("out_d", C.c_void_p), ("stream", C.c_void_p),
        ("step_d", C.c_void_p), ("step_base", C.c_int32), ("nt_src", C.c_int32), ("jmap_len", C.c_int32), ("imap_len", C.c_int32),
    ]


class LecLibraryError(RuntimeError):
    pass


class DiagArgs(C.Structure):
    """struct lec_diag_args (include/lec_hip.h)."""
    _fields_ = [("u_d", C.c_void_p), ("v_d", C.c_void_p), ("hgt_d", C.c_void_p),
                ("nt", C.c_int32), ("ny", C.c_int32), ("nx", C.c_int32), ("reserved0", C.c_int32),
                ("box_d", C.c_void_p), ("xcoef_d", C.c_void_p), ("ycoef_d", C.c_void_p), ("curv_d", C.c_void_p),
                ("val_d", C.c_void_p), ("pos_d", C.c_void_p), ("stream", C.c_void_p)]


class DtdtArgs(C.Structure):
    """struct lec_dtdt_args (include/lec_hip.h)."""
    _fields_ = [("tm_d", C.c_void_p), ("t_d", C.c_void_p), ("tp_d", C.c_void_p), ("dtype", C.c_int32), ("n_steps", C.c_int32),
                ("step_elems", C.c_int64), ("tcoef_d", C.c_void_p), ("out_d", C.c_void_p), ("stream", C.c_void_p)]


class InflateArgs(C.Structure):
    """struct lec_inflate_args (include/lec_hip.h)."""
    _fields_ = [("src_d", C.c_void_p), ("src_bytes", C.c_int64), ("desc_d", C.c_void_p), ("n_streams", C.c_int32), ("flags", C.c_int32),
                ("dst_d", C.c_void_p), ("status_d", C.c_void_p), ("stream", C.c_void_p), ("dst_bytes", C.c_int64)]


class ChunkScatterArgs(C.Structure):
    """struct lec_chunk_scatter_args (include/lec_hip.h)."""
    _fields_ = [("src_d", C.c_void_p), ("chunk_d", C.c_void_p),
                ("n_chunks", C.c_int32), ("elem_size", C.c_int32), ("shuffled", C.c_int32), ("reserved0", C.c_int32),
                ("ct", C.c_int32), ("ck", C.c_int32), ("cj", C.c_int32), ("ci", C.c_int32),
                ("t_base", C.c_int32), ("n_tmap", C.c_int32), ("n_kmap", C.c_int32), ("j0", C.c_int32),
                ("tmap_d", C.c_void_p), ("kmap_d", C.c_void_p),
                ("nt", C.c_int32), ("nl", C.c_int32), ("ny", C.c_int32), ("nx", C.c_int32),
                ("out_d", C.c_void_p), ("stream", C.c_void_p), ("src_bytes", C.c_int64)]


STAGE_BOTH, STAGE_LEVELS, STAGE_VERTICAL = 0, 1, 2      # lec_reduce_args.stage

_lib = None


def load():
    """Loads liblec_hip.so (once).  Raises LecLibraryError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LecLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C lorenzcycletoolkit_amd/csrc` (there is no CPU fallback)")
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  The process must end up with ONE HIP runtime: if
    # this library were loaded first it would pull in /opt/rocm's copy, torch would then load its bundled one, and kernels
    # launched through the first runtime would see no device.  Importing torch first makes the loader resolve our dependency
    # to the runtime torch already brought.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.lec_version.restype = C.c_int
    lib.lec_last_error.restype = C.c_char_p
    lib.lec_max_row.restype = C.c_int
    lib.lec_max_row.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.lec_rowstats.restype = C.c_int
    lib.lec_rowstats.argtypes = [C.POINTER(RowstatsArgs)]
    lib.lec_reduce.restype = C.c_int
    lib.lec_reduce.argtypes = [C.POINTER(ReduceArgs)]
    lib.lec_dropmask.restype = C.c_int
    lib.lec_dropmask.argtypes = [C.POINTER(ReduceArgs)]
    lib.lec_ingest.restype = C.c_int
    lib.lec_ingest.argtypes = [C.POINTER(IngestArgs)]
    lib.lec_track_diag.restype = C.c_int
    lib.lec_track_diag.argtypes = [C.POINTER(DiagArgs)]
    lib.lec_check_boxes.restype = C.c_int
    lib.lec_check_boxes.argtypes = [C.POINTER(RowstatsArgs), C.c_void_p]
    lib.lec_check_maps.restype = C.c_int
    lib.lec_check_maps.argtypes = [C.POINTER(IngestArgs), C.c_void_p]
    lib.lec_host_register.restype = C.c_int
    lib.lec_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.lec_host_unregister.restype = C.c_int
    lib.lec_host_unregister.argtypes = [C.c_void_p]
    lib.lec_copy_rows_async.restype = C.c_int
    lib.lec_copy_rows_async.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    lib.lec_inflate.restype = C.c_int
    lib.lec_inflate.argtypes = [C.POINTER(InflateArgs)]
    lib.lec_inflate_status_text.restype = C.c_char_p
    lib.lec_inflate_status_text.argtypes = [C.c_int]
    lib.lec_chunk_scatter.restype = C.c_int
    lib.lec_chunk_scatter.argtypes = [C.POINTER(ChunkScatterArgs)]
    lib.lec_dtdt.restype = C.c_int
    lib.lec_dtdt.argtypes = [C.POINTER(DtdtArgs)]
    lib.lec_format_csv_rows.restype = C.c_longlong
    lib.lec_format_csv_rows.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_char_p, C.c_int, C.c_void_p, C.c_longlong]
    if lib.lec_version() != LEC_ABI_VERSION:
        raise LecLibraryError(f"liblec_hip.so ABI {lib.lec_version()} != expected {LEC_ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    """Maps a C-ABI return code onto the reference's error behaviour: bad arguments raise ValueError
    (the reference raises ValueError for invalid boxes, lec_fixed_framework.py:121-154)."""
    if rc != 0:
        msg = load().lec_last_error().decode("utf-8", "replace")
        if rc == 1:
            raise ValueError(f"{what}: {msg}")
        raise LecLibraryError(f"{what} failed (code {rc}): {msg}")
