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
 *   lec_reduce    replaces the (level x lat) math of the same calc_* methods: CalcAreaAverage
 *                 (calc_averages.py:46-78), StaticStability (thermodynamics.py:26-73), the
 *                 differentiate("rlats"/level) calls, _handle_nans (energy_contents.py:190-208),
 *                 and the integrate(level) epilogues.
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

#define LEC_ABI_VERSION 2

/* number of fp64 values per (time, level, lat) row record written by lec_rowstats */
#define LEC_NSTAT 32
/* number of per-(time, level) intermediate values written by lec_reduce into `levraw_d` */
#define LEC_NLEVRAW 40
/* per-time outputs of lec_reduce */
#define LEC_NSCALAR 16 /* Az Ae Kz Ke Cz Ca Ck Ce BAz BAe BKz BKe BPhiZ BPhiE Gz Ge */
#define LEC_NLEVFUN 28 /* functions of level that _handle_nans repairs (10 terms + 3 x 6 boundary pieces) */
#define LEC_NLEVTAB 21 /* Az Ae Kz Ke Ge Gz Cz Cz_1 Cz_2 Ca Ca_1 Ca_2 Ce Ce_1 Ce_2 Ck Ck_1..Ck_5 */

enum lec_dtype { LEC_F64 = 0, LEC_F32 = 1 };

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
    LEC_S_SPARE = 28
};

/*
 * Stage 1: one pass over the field cubes -> LEC_NSTAT row statistics per (time, level, box-lat) row.
 *
 * Boxes: `box_d` holds n_box quadruples {iw, ie, js, jn} (inclusive grid indices).  n_box == 1: one
 * fixed (Eulerian) box for every time step; n_box == t_count: one box per processed time step
 * (semi-Lagrangian).  Every table indexed by box has room for nxb_max / nyb_max entries per box.
 *
 * dT/dt for the diabatic-heating residual: if dTdt_d != NULL it is a cube like the fields (moving
 * framework, lorenzcycletoolkit.py:184-186); otherwise dT/dt = tc[t][0] T[t-1] + tc[t][1] T[t] +
 * tc[t][2] T[t+1] with tc = tcoef_d (np.gradient coefficients over the cube's time axis;
 * thermodynamics.py:109-110), neighbours taken from the same cube (so the cube must include the
 * halo time steps a shard needs; coefficients of non-existent neighbours must be 0).
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
} lec_rowstats_args;

/*
 * Stage 2: (level x lat) math on the row records -> per-time scalars and per-level tables.
 */
typedef struct lec_reduce_args {
    const double* rows_d;       /* [t_count][nl][nyb_max][LEC_NSTAT] from lec_rowstats */
    int32_t t_count, nl;
    int32_t n_box, nyb_max;
    const int32_t* box_d;       /* [n_box][4] */
    const double* boxtab2_d;    /* [n_box][4]  c1 = -1/(Re xlen ylen), c2 = -1/(Re ylen), spare, spare */
    const double* lattab2_d;    /* [n_box][nyb_max][8]  cos*wphi/ylen, wphi, cos, tan, d/dphi[rad] a,b,c, spare */
    const double* levtab2_d;    /* [nl][4]  p [Pa], d/dp a,b,c */
    double phi_scale;           /* multiplies the geopotential statistics (g when the file holds height) */
    int32_t drop_any_time;      /* 1 (fixed framework): a level that is still NaN after the _handle_nans interpolation at ANY
                                   processed time step is dropped from the pressure integrals of EVERY time step, as xarray's
                                   dropna(dim=level) does on a [time, level] array (energy_contents.py:203-207); 0: per time step */
    int32_t reserved0;
    int32_t* dropmask_d;        /* workspace [LEC_NLEVFUN][nl] (needed when drop_any_time), zeroed by the call */
    double* am_d;               /* workspace [t_count][nl][8]  area means */
    double* levraw_d;           /* workspace [t_count][nl][LEC_NLEVRAW] */
    double* scalars_d;          /* out [t_count][LEC_NSCALAR] */
    double* levels_d;           /* out [t_count][LEC_NLEVTAB][nl] */
    int32_t* nanflag_d;         /* out [t_count] number of NaN level values repaired/dropped (0 = clean) */
    void* stream;
} lec_reduce_args;

int lec_version(void);
const char* lec_last_error(void);

/* longest box row (in grid points) lec_rowstats accepts for the given dtype / alignment */
int lec_max_row(int dtype, int aligned);

int lec_rowstats(const lec_rowstats_args* args);
int lec_reduce(const lec_reduce_args* args);

#ifdef __cplusplus
}
#endif
#endif /* LEC_HIP_H */
