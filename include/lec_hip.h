/*
 * lec_hip.h -- C ABI of the MI355X (gfx950) Lorenz-Energy-Cycle engine.
 *
 * The reference (daniloceano/LorenzCycleToolkit, pure Python) has no FFI; its hot path is the Python
 * call surface between the frameworks and the numerics.  Each entry point below names the reference
 * interface it replaces (paths relative to the reference repository root):
 *
 *   lec_rowstats  replaces the 4-D passes of  BoxData.__init__            src/utils/box_data.py:78-295
 *                 (CalcZonalAverage            src/utils/calc_averages.py:25-43,
 *                  AdiabaticHEating            src/utils/thermodynamics.py:76-124)
 *                 and every 4-D eddy product / edge selection inside
 *                 EnergyContents.calc_*        src/analysis/energy_contents.py:99-165
 *                 ConversionTerms.calc_*       src/analysis/conversion_terms.py:103-245
 *                 BoundaryTerms.calc_*         src/analysis/boundary_terms.py:125-418
 *                 GenerationDissipationTerms   src/analysis/generation_and_dissipation_terms.py:122-152
 *   lec_ingest    replaces, for data that is already in device memory as raw file bytes, the decode of
 *                 xr.open_dataset (CF scale_factor / add_offset / _FillValue; get_data,
 *                 src/utils/preprocessing.py:35-146), the longitude wrap + sorts + >= 10 hPa filter of
 *                 process_data (preprocessing.py:275-365), the crop of slice_domain
 *                 (src/utils/select_area.py:254-338) and the unit conversion of BoxData._extract_data
 *                 (box_data.py:297-310): one gather pass instead of four host copies.
 *   lec_reduce    replaces the (level x lat) math of the same calc_* methods: CalcAreaAverage
 *                 (calc_averages.py:46-78), StaticStability (thermodynamics.py:26-73), the
 *                 differentiate("rlats"/level) calls, _handle_nans (energy_contents.py:190-208),
 *                 and the integrate(level) epilogues.
 *   lec_track_diag  replaces MetPy's wind_speed / vorticity on the 850-hPa slice and get_position /
 *                 find_extremum_coordinates of the moving framework
 *                 (src/frameworks/lec_moving_framework.py:269-417,650-663; src/utils/tools.py:95-128).
 *
 * Conventions
 *   - All pointers named *_d are DEVICE pointers owned by the caller; the library allocates nothing.
 *   - Field cubes are C-contiguous [nt][nl][ny][nx] (time, level, lat S->N, lon W->E), fp64 or fp32.
 *   - Calls are asynchronous on `stream` (a hipStream_t passed as void*); they do not synchronise.
 *   - Return value 0 = success; non-zero = error, text via lec_last_error() (thread-local).
 *   - No exceptions cross the ABI.  Thread-safe across distinct streams.
 */
#ifndef LEC_HIP_H
#define LEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI 8 (round 4): lec_reduce_args.stage (the two halves of stage 2 run apart; no 65535-step limit), lec_inflate_args.dst_bytes and
 * lec_chunk_scatter_args.src_bytes (the destination / payload ranges of every descriptor are bounds-checked on the device). */
#define LEC_ABI_VERSION 10
/* ABI 10 (round 6): lec_ingest_args.nt_src (was reserved0) / jmap_len / imap_len: a per-step gather table (step_d) is bounds-checked --
 * lec_ingest refuses one without them, its kernel writes NaN rows for an entry that points outside the source or the maps instead of
 * reading there, and lec_check_maps scans the table and names the first bad entry.
 * ABI 9 (round 5): lec_format_csv_rows (host): a per-level table as the text pandas writes for it, one call per table;
 * lec_rowstats_args.tm_d / tp_d: box-packed series of the moving framework (the struct grew by two pointers at its end); lec_dtdt;
 * lec_ingest_args.step_d / step_base: per-step gathers (the struct grew at its end). */

/* number of fp64 values per (time, level, lat) row record written by lec_rowstats */
#define LEC_NSTAT 32
/* number of per-(time, level) intermediate values written by lec_reduce into `levraw_d` */
#define LEC_NLEVRAW 40
/* per-time outputs of lec_reduce */
#define LEC_NSCALAR 16 /* Az Ae Kz Ke Cz Ca Ck Ce BAz BAe BKz BKe BPhiZ BPhiE Gz Ge */
#define LEC_NLEVFUN 28 /* functions of level that _handle_nans repairs (10 terms + 3 x 6 boundary pieces) */
#define LEC_NLEVTAB 21 /* Az Ae Kz Ke Ge Gz Cz Cz_1 Cz_2 Ca Ca_1 Ca_2 Ce Ce_1 Ce_2 Ck Ck_1..Ck_5 */

enum lec_dtype { LEC_F64 = 0, LEC_F32 = 1, LEC_I16 = 2, LEC_I32 = 3, LEC_I8 = 4 /* the integers: lec_ingest source only */ };

/* Stage-1 kernel families (lec_tuning.kernel).  Every family writes the same row records; AUTO is what is
 * measured and shipped, the others exist for cross-checks and A/B measurements.  The library reads NO
 * environment variables: everything that steers it is in the argument structs.
 * The workgroup orders assume the MI355X's 8 XCDs (workgroups are dealt to them round-robin, so blockIdx % 8 labels the XCD); on a
 * part with another count the results are the same and only the locality differs. */
enum lec_kernel {
    LEC_KERNEL_AUTO = 0,       /* box tiles for per-time-step boxes, row blocks for all terms on one fixed fp64 box, else one wave per row */
    LEC_KERNEL_TWO_SWEEP = 1,  /* lec_rowstats.hip: the reference's own order (deviation from the zonal mean, then products); rows <= lec_max_row() */
    LEC_KERNEL_ROW_SWEEP = 2,  /* lec_rowsweep.hip: one wave per row, one sweep */
    LEC_KERNEL_ROW_BLOCK = 3,  /* lec_rowblock.hip: blocks of neighbouring rows exchange T through LDS (all terms, one fixed box, dT/dt from the cube) */
    LEC_KERNEL_BOX_TILE = 4,   /* lec_boxtile.hip: one wave per four box rows walks the levels; six values per point transposed through LDS */
    LEC_KERNEL_BOX_PLANE = 5   /* lec_boxplane.hip (ABI 10): the same rows and sums with the planes brought into LDS by DMA -- a box-packed fp64 series with a
                                  dT/dt cube on even longitudes, slabs of at most 64 columns (what LEC_KERNEL_AUTO picks there; bit-identical records) */
};

/* workgroup -> row order of the row kernels (speed only) */
enum lec_order {
    LEC_ORDER_AUTO = 0,
    LEC_ORDER_MEMORY = 1,      /* rows in memory order */
    LEC_ORDER_XCD_LAT = 2,     /* every XCD owns a latitude chunk, walked latitude-fastest per (time, level) */
    LEC_ORDER_XCD_TILED = 7    /* tiles of tile_t time steps x tile_j latitudes at one level, levels next (all terms on one fixed box) */
};

/* Kernel selection of lec_rowstats; all zero = library defaults. */
typedef struct lec_tuning {
    int32_t kernel;        /* enum lec_kernel */
    int32_t block_shape;   /* LEC_KERNEL_ROW_BLOCK: 100 bt + 10 bk + bj waves (time x level x latitude, each 1 or 2); 0 = 212 */
    int32_t order;         /* enum lec_order */
    int32_t tile_t, tile_j; /* tile extents of LEC_ORDER_XCD_TILED / of the row-block kernel (in blocks); box tiles: tile_t = time steps per
                               workgroup group, tile_j = levels per wave (default: from the launch size; at most 21, more is LEC_ERR_ARG);
                               0 = default; must be >= 0 */
    int32_t f32_vec;       /* fp32 storage, one wave per row: 0 = float4 trips when the cubes are 16-byte aligned, 2 = float2 trips */
    int32_t reserved[2];   /* must be 0 */
} lec_tuning;

enum lec_status {
    LEC_OK = 0,
    LEC_ERR_ARG = 1,      /* bad argument (null pointer, shape, alignment) */
    LEC_ERR_UNSUPPORTED = 2, /* row longer than the kernels support */
    LEC_ERR_LAUNCH = 3    /* HIP reported a launch error */
};

/* row-record layout (index into the LEC_NSTAT values of one row) */
enum lec_stat {
    LEC_S_MT = 0, LEC_S_MU, LEC_S_MV, LEC_S_MW, LEC_S_MP, LEC_S_MQ,      /* zonal means [T] [u] [v] [w] [Phi] [Q] */
    LEC_S_TT = 6, LEC_S_UU, LEC_S_VV, LEC_S_VT, LEC_S_WT, LEC_S_UV,      /* [T'T'] [u'u'] [v'v'] [v'T'] [w'T'] [u'v'] */
    LEC_S_WU = 12, LEC_S_WV, LEC_S_WP, LEC_S_QT,                         /* [w'u'] [w'v'] [w'Phi'] [Q'T'] */
    LEC_S_VTT = 16, LEC_S_WTT, LEC_S_KV, LEC_S_KW, LEC_S_EV, LEC_S_EW,   /* [vT'T'] [wT'T'] [Kv] [Kw] [Ev] [Ew] */
    LEC_S_TW = 22, LEC_S_TE, LEC_S_UW, LEC_S_UE, LEC_S_VW, LEC_S_VE,     /* T,u,v at the west / east box column */
    LEC_S_SPARE = 28    /* 28..31: scratch of lec_rowstats (cross-time covariance pieces), not an interface */
};

/*
 * Stage 1: one pass over the field cubes -> LEC_NSTAT row statistics per (time, level, box-lat) row.
 *
 * Boxes: `box_d` holds n_box quadruples {iw, ie, js, jn} (inclusive grid indices).  box_per_step == 0: one
 * fixed (Eulerian) box for every time step, n_box == 1; box_per_step == 1: one box per processed time step
 * (semi-Lagrangian), n_box == t_count -- also when t_count == 1, so that a one-step shard or chunk of a
 * moving series runs the same kernels (and gives the same bits) as the whole series.  Every table indexed
 * by box has room for nxb_max / nyb_max entries per box.
 *
 * dT/dt for the diabatic-heating residual: if dTdt_d != NULL it is a cube like the fields (moving
 * framework, lorenzcycletoolkit.py:184-186); otherwise dT/dt = tc[t][0] T[t-1] + tc[t][1] T[t] +
 * tc[t][2] T[t+1] with tc = tcoef_d (np.gradient coefficients over the cube's time axis;
 * thermodynamics.py:109-110), neighbours taken from the same cube (so the cube must include the
 * halo time steps a shard needs; coefficients of non-existent neighbours must be 0).
 *
 * Box-packed series (ABI 9; box_per_step == 1, served by the box-tile kernel).  The reference's moving framework slices the box of
 * every time step out of the data (src/utils/box_data.py:297-310) before anything is computed; a producer that gathers anyway -- the
 * device ingest from file bytes, a host that uploads a track -- may hand over just those slices: cubes [nt][nl][ny][nx] whose step t
 * holds box t alone, its south-west corner at [t][k][0][0] (ny, nx >= the tallest / widest box; what lies outside a step's box is
 * never read).  `box_d` then describes the boxes AS THE CUBES HOLD THEM, {0, nxb - 1, 0, nyb - 1}, while every table indexed by box
 * (boxtab_d, wlon_d, glon_d, lattab_d) is built from the box's true grid coordinates as always.  The cube's time neighbours of step t
 * are other boxes, so T(t - 1) and T(t + 1) ON THE BOX OF STEP t come in two more cubes of the same layout, `tm_d` / `tp_d` (the step
 * itself where a neighbour does not exist: its coefficient is 0 -- but 0 x NaN is NaN), and dT/dt = tc[t][0] tm[t] + tc[t][1] T[t] +
 * tc[t][2] tp[t].  Same arithmetic on the same values: the row records are bit-identical to those of the unpacked cubes.  Rows of
 * 488 bytes scattered through a 243-column crop replay at 4.8 TB/s on MI355X, a step's box stored as one dense block at 5.7
 * (tools/probes/probe_boxread.hip) -- and a producer writes a tenth of the bytes.
 */
typedef struct lec_rowstats_args {
    /* fields */
    const void* tair_d;
    const void* u_d;
    const void* v_d;
    const void* omega_d;
    const void* geopt_d; /* may be NULL: Phi statistics are written as 0 */
    const void* dTdt_d;  /* may be NULL, see above */
    int32_t dtype;       /* enum lec_dtype, common to all cubes */
    int32_t with_q;      /* 0: skip the diabatic-heating statistics (MQ, QT written as 0) */
    int32_t nt, nl, ny, nx;     /* cube dimensions */
    int32_t t_begin, t_count;   /* process time steps [t_begin, t_begin + t_count) */
    /* boxes */
    int32_t n_box, nxb_max, nyb_max;
    int32_t lon_uniform;        /* 1: longitudes of every box are uniformly spaced (fast path) */
    int32_t box_per_step;       /* 0: one fixed box (n_box == 1); 1: one box per processed time step (n_box == t_count) */
    int32_t reserved0;          /* must be 0 */
    const int32_t* box_d;       /* [n_box][4] iw ie js jn */
    const double* boxtab_d;     /* [n_box][4]  1/xlength [rad^-1], h_rad, 1/h_deg, spare (h_* used if lon_uniform) */
    const double* wlon_d;       /* [n_box][nxb_max]     trapezoid weights in radians   (used if !lon_uniform) */
    const double* glon_d;       /* [n_box][nxb_max][3]  d/dlon[deg] coefficients a,b,c (used if !lon_uniform) */
    const double* lattab_d;     /* [n_box][nyb_max][4]  d/dlat[deg]/dy coefficients a,b,c ; 1/dx_j   (with_q) */
    const double* levtab_d;     /* [nl][3]  S = al T[k-1] + be T[k] + ga T[k+1] static-stability coefficients (with_q) */
    const double* tcoef_d;      /* [nt][3]  (with_q && dTdt_d == NULL) */
    /* output */
    double* rows_d;             /* [t_count][nl][nyb_max][LEC_NSTAT] */
    void* stream;
    lec_tuning tuning;          /* all zero = defaults */
    /* ABI 9: a BOX-PACKED series (see above); both NULL otherwise */
    const void* tm_d;           /* T of the previous time step on the box of step t, laid out like tair_d */
    const void* tp_d;           /* T of the next time step on the box of step t */
} lec_rowstats_args;

/*
 * Stage 2: (level x lat) math on the row records -> per-time scalars and per-level tables.
 * Limits: nl <= 160 levels (LEC_ERR_UNSUPPORTED beyond); nl * t_count < 2^26 per call (the launch is nl * t_count workgroups of 64
 * threads and HIP takes fewer than 2^32 threads per launch; LEC_ERR_UNSUPPORTED beyond: 1.8 million steps of 37 levels).
 *
 * The call has two halves that may also be run apart (`stage`, ABI 8), for series whose row records are not held whole:
 *   LEC_STAGE_LEVELS    rows_d -> levraw_d : the level x latitude work of the call's time steps; needs rows_d, box / lat / lev tables,
 *                       am_d, levraw_d -- 6.8 MB of row records per 37 x 721 time step become 12 KB, so a streamed series runs
 *                       this half chunk by chunk into ONE levraw buffer of the whole series and recycles the row records;
 *   LEC_STAGE_VERTICAL  levraw_d -> dropmask / scalars / levels / nanflag over t_count steps of levraw records (rows_d, box_d, am_d and
 *                       lattab2_d are not read and may be NULL; boxtab2_d is: the per-box constants c1, c2).
 * LEC_STAGE_BOTH (0) is the whole call.  Either way every number is the same: the halves are the same kernels.
 */
#define LEC_MAX_LEVELS 160
#define LEC_STAGE_BOTH 0
#define LEC_STAGE_LEVELS 1
#define LEC_STAGE_VERTICAL 2
typedef struct lec_reduce_args {
    const double* rows_d;       /* [t_count][nl][nyb_max][LEC_NSTAT] from lec_rowstats; 16-byte aligned */
    int32_t t_count, nl;
    int32_t n_box, nyb_max;
    const int32_t* box_d;       /* [n_box][4] */
    const double* boxtab2_d;    /* [n_box][4]  c1 = -1/(Re xlen ylen), c2 = -1/(Re ylen), spare, spare */
    const double* lattab2_d;    /* [n_box][nyb_max][8]  cos*wphi/ylen, wphi, cos, tan, d/dphi[rad] a,b,c, spare; 16-byte aligned */
    const double* levtab2_d;    /* [nl][4]  p [Pa], d/dp a,b,c */
    double phi_scale;           /* multiplies the geopotential statistics (g when the file holds height) */
    int32_t drop_any_time;      /* _handle_nans' dropna(dim=level) on a [time, level] array (energy_contents.py:203-207), fixed framework:
                                   a level that is still NaN after the interpolation at ANY time step is dropped from the pressure
                                   integrals of EVERY time step.
                                   0: per time step (moving framework: one BoxData per step);
                                   1: any-time over the time steps of this call (mask computed by the call);
                                   2: any-time with the mask in dropmask_d taken as given: the caller ran lec_dropmask on every
                                      shard / chunk of the series and merged the masks (element-wise max) */
    int32_t stage;              /* LEC_STAGE_BOTH (0), LEC_STAGE_LEVELS, LEC_STAGE_VERTICAL (see above) */
    int32_t* dropmask_d;        /* [LEC_NLEVFUN][nl] (needed when drop_any_time): workspace zeroed by the call (mode 1), input (mode 2) */
    double* am_d;               /* workspace [t_count][nl][8]  area means (always required; left untouched when nyb_max <= 64) */
    double* levraw_d;           /* workspace [t_count][nl][LEC_NLEVRAW] */
    double* scalars_d;          /* out [t_count][LEC_NSCALAR] */
    double* levels_d;           /* out [t_count][LEC_NLEVTAB][nl] */
    int32_t* nanflag_d;         /* out [t_count] number of NaN level values repaired/dropped (0 = clean) */
    void* stream;
    int64_t scalars_stride;     /* doubles between the scalars of consecutive time steps; 0 = dense (LEC_NSCALAR) */
    int64_t levels_stride;      /* doubles between the level tables of consecutive time steps; 0 = dense (LEC_NLEVTAB * nl).  With
                                   scalars_d = buf, levels_d = buf + LEC_NSCALAR and both strides = LEC_NSCALAR + LEC_NLEVTAB * nl the call
                                   writes one packed record per time step: the send buffer of the time-sharded gather, no repacking */
} lec_reduce_args;

/*
 * Ingest: raw source cube [nt][nl_in][ny_in][nx_in] (file order, file byte order, possibly CF-packed) ->
 * field cube [nt][nl][ny][nx] in the order lec_rowstats wants (level ascending in Pa, lat S->N, lon W->E),
 * cropped to the analysis domain, in SI units.
 *
 *   value = decode(src[t][kmap[k]][jmap[j]][imap[i]])            source element, byte-swapped if asked
 *   fill:                  value == fill_value (compared before scaling) -> NaN
 *   decode_dtype LEC_F64:  v = (double)value; packed: v = v * scale_factor; v = v + add_offset      (two roundings, fp64)
 *   decode_dtype LEC_F32:  v = (float)value;  packed: v = (float)((double)v * scale_factor); v = (float)((double)v + add_offset)
 *                          -- the reference's pinned xarray 2024.2.0 decodes int16 data to float32 when the variable has a
 *                          fill value or no add_offset, and keeps float32 data float32 (float64 attributes: each operation
 *                          is computed in fp64 and rounded to float32); the caller picks the dtype by those rules
 *   out = v * unit_scale   in decode_dtype, stored as out_dtype
 */
typedef struct lec_ingest_args {
    const void* src_d;          /* device copy of the raw variable bytes for nt time steps */
    int32_t src_dtype;          /* LEC_I8, LEC_I16, LEC_I32, LEC_F32 or LEC_F64 (int32 values are exact in fp64 only: decode_dtype LEC_F64) */
    int32_t swap_bytes;         /* 1: source is in the opposite byte order (classic NetCDF is big-endian) */
    int32_t nt, nl_in, ny_in, nx_in;
    int32_t nl, ny, nx;         /* output extents */
    const int32_t* kmap_d;      /* [nl] output level -> source level */
    const int32_t* jmap_d;      /* [ny] output latitude -> source latitude */
    const int32_t* imap_d;      /* [nx] output longitude -> source longitude */
    int32_t has_packing, has_fill;
    double scale_factor, add_offset, fill_value, unit_scale;
    int32_t out_dtype;          /* LEC_F64 or LEC_F32: storage of the output cube (>= decode_dtype; widening is exact) */
    int32_t decode_dtype;       /* LEC_F64 or LEC_F32: the precision the reference's decode gives this variable (see below) */
    void* out_d;                /* [nt][nl][ny][nx] */
    void* stream;
    /* ABI 9: per-step gathers (a box-packed series of the moving framework: every output step holds another box, possibly of another
     * source step).  NULL: output step t reads source step t with the maps as they are.  Else step_d[t] = {source step, latitude
     * offset, longitude offset}: output step t reads source step step_d[t][0] - step_base of src_d through jmap_d[j + step_d[t][1]] and
     * imap_d[i + step_d[t][2]] -- the maps must be long enough for the largest offset + ny / nx (lengthen them by repeating their last
     * entry); nt counts OUTPUT steps, the source holds nt_src steps.
     * ABI 10: the table lives in device memory, where argument validation cannot see it (the reference bounds-checks its track against
     * the data on the host, lec_moving_framework.py:112-154), so the call carries what bounds it: nt_src, jmap_len, imap_len.  An entry
     * with a source step outside [step_base, step_base + nt_src) or an offset outside [0, jmap_len - ny] / [0, imap_len - nx] never
     * reads: lec_ingest writes NaN into that output step's rows, and lec_check_maps names the first such entry. */
    const int32_t* step_d;      /* [nt][3] or NULL */
    int32_t step_base;          /* the source step that src_d starts with */
    int32_t nt_src;             /* time steps src_d holds (step_d given: >= 1; else ignored -- the source then holds nt steps) */
    int32_t jmap_len, imap_len; /* entries of jmap_d / imap_d (step_d given: >= ny / nx; else 0 = ny / nx) */
} lec_ingest_args;

/*
 * 850-hPa track diagnostics of the moving framework, one box per time step:
 *   lec_moving_framework.py:650-663  wind_speed(u, v), vorticity(u, v) on the whole 850-hPa slice
 *   lec_moving_framework.py:269-417  get_position: the extrema inside the box (inclusive label slices)
 *   tools.py:95-128                  find_extremum_coordinates
 * zeta(j, i) = sum_k xcoef[j][i][k] v[j][i0 + k] - sum_k ycoef[j][k] u[j0 + k][i] + curv[j] u[j][i],  i0 = clamp(i - 1, 0, nx - 3),
 * j0 = clamp(j - 1, 0, ny - 3): three-point derivatives, second order also at the ends of the slice (the stencil of
 * metpy.calc.first_derivative).  The FORMULATION is the caller's: the coefficient tables carry the metric (1/m).  The Python host
 * builds two (diagnostics.vorticity_tables): "metpy_no_crs" -- what MetPy 1.6.2 evaluates for DataArrays without a CRS, as the
 * reference passes them: plain dv/dx - du/dy, distances = great-circle arcs between neighbouring grid points on pyproj's default
 * sphere (a = 6,370,997 m), curv = 0 -- and "spherical": dx = Re cos(phi) dlambda, dy = Re dphi, curv = tan(phi) / Re.
 * NaN (below-ground points) is skipped, values and positions alike; among equal values the first in row-major order of the box
 * wins (numpy's argmin / argmax).  Parity of the vorticity against MetPy itself is unpinned.
 */
typedef struct lec_diag_args {
    const double* u_d;          /* [nt][ny][nx] eastward wind at 850 hPa (m/s) */
    const double* v_d;          /* [nt][ny][nx] northward wind */
    const double* hgt_d;        /* [nt][ny][nx] geopotential height (gpm) */
    int32_t nt, ny, nx, reserved0;
    const int32_t* box_d;       /* [nt][6]  iw, ie, js, jn: inclusive index ranges of the box; jc, ic: the grid point nearest its centre */
    const double* xcoef_d;      /* [ny][nx][3]  d/dx coefficients of point (j, i) along its row (1/m) */
    const double* ycoef_d;      /* [ny][3]      d/dy coefficients of row j along a column (1/m) */
    const double* curv_d;       /* [ny]         coefficient of u (1/m): tan(phi) / Re on the sphere, 0 for the plain Cartesian form */
    double* val_d;              /* [nt][5]  zeta minimum, zeta maximum, height minimum, wind-speed maximum, zeta at (jc, ic);
                                            NaN when the box holds no finite value */
    int32_t* pos_d;             /* [nt][8]  (j, i) grid indices of the four extrema, in that order; -1 when there is none */
    void* stream;
} lec_diag_args;

int lec_version(void);
const char* lec_last_error(void);

/* longest box row (in grid points) lec_rowstats accepts for the given dtype / alignment / kernel family
 * (enum lec_kernel; only LEC_KERNEL_TWO_SWEEP, which holds a row in registers, has a practical limit) */
int lec_max_row(int dtype, int aligned, int kernel);

int lec_rowstats(const lec_rowstats_args* args);
int lec_ingest(const lec_ingest_args* args);
int lec_reduce(const lec_reduce_args* args);

/* The any-time NaN-level mask of the time steps in `args` alone -> dropmask_d (zeroed first; non-zero = drop).
 * Uses am_d / levraw_d as workspace (stage LEC_STAGE_VERTICAL: reads levraw_d as given); scalars_d, levels_d, nanflag_d are not
 * touched and may be NULL.
 * For series processed in shards or chunks: merge the masks, then call lec_reduce with drop_any_time = 2. */
int lec_dropmask(const lec_reduce_args* args);

int lec_track_diag(const lec_diag_args* args);

/*
 * What the library cannot see at launch: indices that live in DEVICE memory.  lec_rowstats validates every scalar argument, but a
 * quadruple of box_d outside [0, nx) x [0, ny), with ie < iw + 1 / jn < js + 1, or wider / higher than nxb_max / nyb_max would be
 * an out-of-bounds read; likewise an entry of the three ingest maps outside the source extents.  The Python host checks its boxes
 * before it uploads them (tables.box_indices; the reference validates its box the same way, lec_fixed_framework.py:98-154); a
 * plain-C caller calls these first.  One small kernel on args->stream scans the table(s), the call WAITS for it (the only
 * synchronous entry points) and returns LEC_OK, or LEC_ERR_ARG with the first offending entry in lec_last_error().
 * status_d: caller-owned device scratch of 4 int32.
 */
int lec_check_boxes(const lec_rowstats_args* args, int32_t* status_d);
int lec_check_maps(const lec_ingest_args* args, int32_t* status_d);

/*
 * Host-memory plumbing of the device ingest: move file bytes to the GPU WITHOUT a staging copy.  The reference reads its file
 * through xarray into NumPy memory (src/utils/preprocessing.py:35-146); here a span of the memory-mapped file (or of any host
 * array) is registered with the HIP runtime -- its pages are pinned and mapped for the copy engines --, the rows a chunk needs are
 * copied from it asynchronously, and the span is unregistered once the copies have completed (the caller synchronises on them
 * first and keeps the bookkeeping: registered spans must not overlap).
 *   lec_host_register    ptr / bytes: page-aligned span of host memory (read-only file mappings are fine)
 *   lec_host_unregister  the pointer a span was registered with
 *   lec_copy_rows_async  `rows` rows of `width_bytes`, source rows `src_pitch` bytes apart (host), destination rows `dst_pitch` apart
 *                        (device), on `stream`; rows == 1 or both pitches == width: one linear copy.  Truly asynchronous only from
 *                        registered (or pinned) memory.
 * Return LEC_OK, LEC_ERR_ARG, or LEC_ERR_LAUNCH with the runtime's message when HIP refuses (the Python host then falls back to its
 * pinned staging buffers).
 */
int lec_host_register(const void* ptr, size_t bytes);
int lec_host_unregister(const void* ptr);
int lec_copy_rows_async(void* dst_d, size_t dst_pitch, const void* src_h, size_t src_pitch, size_t width_bytes, size_t rows, void* stream);

/*
 * Deflated NetCDF-4 input on the device.  The reference reads such files through netCDF4 / HDF5, which inflate every chunk on one
 * host thread (src/utils/preprocessing.py:35-146 -> xr.open_dataset); here the compressed chunks cross the link as they lie in
 * the file and the GPU inflates them, one wave per chunk:
 *   lec_inflate        n_streams independent zlib streams (RFC 1950 / 1951: stored, fixed and dynamic blocks) -> their
 *                      inflated bytes.  desc_d[s] = {byte offset of the stream in src_d (any), its size, byte offset of
 *                      its output in dst_d (a multiple of 16), the output's exact size}; a NEGATIVE size marks |size| bytes that are
 *                      not compressed (HDF5 stores a chunk as it is when deflate does not pay) and are copied.  src_bytes is the size of the src_d
 *                      allocation (the kernel reads whole dwords: it may touch up to 512 bytes after a stream's end, never beyond
 *                      src_bytes).  status_d[s] = {code, deflate block, output position, input bit position}; code 0 = ok, anything
 *                      else names what was wrong with the stream (lec_inflate_status_text); zlib's Adler-32 trailer is verified against the
 *                      inflated data, HDF5's Fletcher-32 of the compressed bytes when flags says the chunks carry one.
 *                      Asynchronous like every other entry point: the caller reads status_d after synchronising.
 *   lec_chunk_scatter  the payloads of n_chunks HDF5 chunks (each ct x ck x cj x ci elements of elem_size bytes, all of one
 *                      variable; shuffled = 1: the HDF5 shuffle filter's byte planes, undone here) -> a contiguous array
 *                      [nt][nl][ny][nx] of raw elements, which lec_ingest then decodes.  chunk_d[c] = {byte offset of the payload
 *                      in src_d, the chunk's origin in FILE coordinates t, k, j, i}; file step t lands in output step
 *                      tmap_d[t - t_base], file level k in output level kmap_d[k] (either < 0: not wanted), file row j in output
 *                      row j - j0, column i in column i; whatever falls outside the output (edge chunks are padded) is skipped.
 */
typedef struct lec_inflate_args {
    const void* src_d;
    int64_t src_bytes;
    const int64_t* desc_d;      /* [n_streams][4] */
    int32_t n_streams;
    int32_t flags;              /* bit 0: every stream is followed by 4 bytes of HDF5 Fletcher-32 over its bytes (filter 3): verified first;
                                   bit 1: a hint -- matches rarely reach more than 3 KB back (byte-shuffled rows of < 3000 elements): a 4 KiB
                                   instead of an 8 KiB history ring in LDS, 18 instead of 12 streams per CU; any stream still inflates */
    void* dst_d;
    int32_t* status_d;          /* [n_streams][4] */
    void* stream;
    int64_t dst_bytes;          /* size of the dst_d allocation (ABI 8): a descriptor whose output offset is negative, not a multiple of 16, or
                                   whose output does not end inside dst_d gets the status "size" and nothing of it is written */
} lec_inflate_args;

typedef struct lec_chunk_scatter_args {
    const void* src_d;
    const int64_t* chunk_d;     /* [n_chunks][5] */
    int32_t n_chunks, elem_size, shuffled, reserved0;
    int32_t ct, ck, cj, ci;
    int32_t t_base, n_tmap, n_kmap, j0;
    const int32_t* tmap_d;      /* [n_tmap] */
    const int32_t* kmap_d;      /* [n_kmap] */
    int32_t nt, nl, ny, nx;
    void* out_d;
    void* stream;
    int64_t src_bytes;          /* size of the src_d allocation (ABI 8): a chunk whose payload does not lie inside it is skipped */
} lec_chunk_scatter_args;

int lec_inflate(const lec_inflate_args* args);
const char* lec_inflate_status_text(int code);
int lec_chunk_scatter(const lec_chunk_scatter_args* args);

/*
 * dT/dt of a box-packed series as a cube of its own: out[s][e] = tc[s][0] tm[s][e] + tc[s][1] t[s][e] + tc[s][2] tp[s][e] in fp64, evaluated
 * exactly as lec_rowstats evaluates it per point from tm_d / tp_d (the two outer products rounded, then one fused multiply-add), so a
 * series handed over as T, u, v, omega, Phi + this cube (dTdt_d) gives the records of the same series handed over with tm_d / tp_d,
 * bit for bit -- with one streamed operand fewer per point (fp64 storage: 48 instead of 56 bytes; the producer reads its three T
 * slices once).  Replaces, for a moving box, the np.gradient over time of run_lec_analysis (lorenzcycletoolkit.py:184-186) restricted
 * to each step's box.  `dtype` is the cubes' (LEC_F64 / LEC_F32); the output is fp64 (as a dTdt_d of lec_rowstats it serves fp64 storage).
 */
typedef struct lec_dtdt_args {
    const void* tm_d;          /* [n_steps][step_elems] T of the previous time step on the step's box */
    const void* t_d;           /* T */
    const void* tp_d;          /* T of the next time step */
    int32_t dtype, n_steps;
    int64_t step_elems;        /* elements per time step (nl * ny * nx of the packed slabs) */
    const double* tcoef_d;     /* [n_steps][3] np.gradient coefficients of the steps these cubes hold */
    double* out_d;             /* [n_steps][step_elems] fp64 */
    void* stream;
} lec_dtdt_args;
int lec_dtdt(const lec_dtdt_args* args);

/*
 * The text of a per-level table (HOST memory in, host memory out; no device work).  Replaces the per-cell formatting inside
 * `_save_vertical_levels` (src/analysis/conversion_terms.py:287-308 and its copies in energy_contents.py,
 * generation_and_dissipation_terms.py: DataFrame.to_csv(mode="a", header=None)) for the 21 tables of a series: `rows` lines, each
 * `labels + r * label_len` (label_len bytes: the time stamp as the reference prints it), then for each of the `cols` values of the row
 * (`row_stride` doubles from one row to the next) a ',' and the value as Python's repr(float) -- shortest round-trip digits,
 * positional for 1e-4 <= |x| < 1e16 with at least ".0", else d[.ddd]e[+-]XX --, NaN as an empty field (pandas' na_rep), and '\n'.
 * Byte for byte what pandas writes for float64 columns.  `out` must hold rows * (label_len + 27 * cols + 1) bytes.
 * Returns the number of bytes written, or -1 (text via lec_last_error()).
 */
long long lec_format_csv_rows(const double* values, long long rows, long long cols, long long row_stride,
                              const char* labels, int label_len, char* out, long long cap);

#ifdef __cplusplus
}
#endif
#endif /* LEC_HIP_H */
